// Slot-space layout of the Co-SLAM decoders (ColorSDFNet_v2,
// slam/model_components/decoder_coslam.py) for the fused renderer.
//
// Activations live in the MFMA "D layout": lane (j = l&15 sample, q = l>>4)
// holds slots 16*t + 4*q + r (tile t, register r).  K-step s = 4*t + r of the
// next layer consumes register (t, r) as its B operand, so the chain of layers
// never leaves registers.  "Slot" order is chosen so that each lane owns whole
// pieces of the encodings:
//   sdf layer-1 input (80 slots): slots 0..31 hash features — lane group q owns
//     levels q, q+4, q+8, q+12:  slot 16t+4q+r <-> level q+4*(2t+(r>>1)),
//     component r&1;  slots 32..79 OneBlob in natural order (dim*16 + bin), so
//     lane group q owns bins 4q..4q+3 of each dimension.
//   colour layer-1 input (64 slots): 0..47 OneBlob, 48 = sdf (zero weight),
//     49..63 geometry features 0..14 (= sdf-net outputs 1..15).
//   colour layer-2 output: slots 0..2 = rgb logits (rest zero weights).
#pragma once
#include <stdint.h>

namespace xrd {
namespace cs {

// flat decoder vector (state_dict order)
constexpr int kC0 = 0;             // color_net.model.0.weight [32,63]
constexpr int kC1 = kC0 + 32 * 63; // color_net.model.2.weight [3,32]
constexpr int kS0 = kC1 + 3 * 32;  // sdf_net.model.0.weight   [32,80]
constexpr int kS1 = kS0 + 32 * 80; // sdf_net.model.2.weight   [16,32]
constexpr int kFlatLen = kS1 + 16 * 32;

// layers in slot space: out slots x in slots
constexpr int kOut[4] = {32, 16, 32, 16};
constexpr int kIn[4] = {80, 32, 64, 32};
// slot-space gradient buffer [out][in] per layer
constexpr int kD0 = 0, kD1 = kD0 + 32 * 80, kD2 = kD1 + 16 * 32,
              kD3 = kD2 + 32 * 64, kDwLen = kD3 + 16 * 32;
// forward fragments (A = W, out rows): layer l, M tile, K-step s
constexpr int kF0 = 0, kF1 = kF0 + 2 * 20, kF2 = kF1 + 1 * 8,
              kF3 = kF2 + 2 * 16, kFEnd = kF3 + 1 * 8;
// backward fragments (A = W^T, in rows): layer l, M' tile, K-step s
constexpr int kB0 = kFEnd, kB1 = kB0 + 5 * 8, kB2 = kB1 + 2 * 4,
              kB3 = kB2 + 4 * 8, kBEnd = kB3 + 2 * 4;
constexpr int kPackLen = kBEnd * 64;

// flat index of slot-space weight (layer, out slot o, in slot i); -1 = zero
inline int flat_of(int layer, int o, int i) {
  switch (layer) {
    case 0: {
      int col = i;
      if (i < 32) {
        const int t = i >> 4, q = (i >> 2) & 3, r = i & 3;
        col = 2 * (q + 4 * (2 * t + (r >> 1))) + (r & 1);
      }
      return kS0 + o * 80 + col;
    }
    case 1: return kS1 + o * 32 + i;
    case 2:
      if (i < 48) return kC0 + o * 63 + i;
      if (i == 48) return -1;
      return kC0 + o * 63 + (i - 1);
    default: return o < 3 ? kC1 + o * 32 + i : -1;
  }
}

}  // namespace cs
}  // namespace xrd
