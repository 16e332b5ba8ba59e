// Parameter layouts of the Vox-Fusion decoder
// (slam/model_components/decoder_voxfusion.py:85-149 with the model's
// defaults, sparse_voxel.py:59-62: in_dim 16, width 128, depth 2, no
// positional encoding):
//   h1 = relu(W0 x + b0)            x: trilinear voxel feature [16]
//   h2 = relu(W1 h1 + b1)
//   out = WS h2 + bS                out[0] = sdf, out[1:129] = sdf feature f
//   hc = relu(WC [f, x] + bC)
//   rgb = sigmoid(WO hc + bO)
//
// "flat"   = the reference's state_dict order, concatenated.
// "packed" = MFMA-fragment order (see nice_layout.h): a fragment = 64 floats,
//            the A operand of one v_mfma_f32_16x16x4_f32.  Forward fragment
//            (jt, s) holds W[16jt + m][kin(s, q)] (rows = output features),
//            backward fragment (kt, s) holds W[kmap(s, q)][16kt + m] (rows =
//            input features); lane l = (m = l & 15, q = l >> 4).  Each pass of
//            each layer is one contiguous range that the kernels stage in LDS.
#pragma once
#include <stdint.h>

#include "nice_layout.h"  // kmap

namespace xrd {

struct VoxFlat {
  static constexpr int W0 = 0;                  // [128][16]
  static constexpr int B0 = W0 + 128 * 16;
  static constexpr int W1 = B0 + 128;           // [128][128]
  static constexpr int B1 = W1 + 128 * 128;
  static constexpr int WS = B1 + 128;           // [129][128]
  static constexpr int BS = WS + 129 * 128;     // [129]
  static constexpr int WC = BS + 129;           // [128][144]: 128 f, 16 x
  static constexpr int BC = WC + 128 * 144;
  static constexpr int WO = BC + 128;           // [3][128]
  static constexpr int BO = WO + 3 * 128;
  static constexpr int LEN = BO + 3;            // 54276
};

struct VoxPack {
  // ---- forward passes --------------------------------------------------------
  static constexpr int F0 = 0;                       // stage: layer 0
  static constexpr int W0 = F0;                      // frag (jt, s): 8 x 4
  static constexpr int B0 = W0 + 8 * 4 * 64;         // [128]
  static constexpr int F0_LEN = 8 * 4 * 64 + 128;
  static constexpr int F1 = F0 + F0_LEN;             // stage: layer 1
  static constexpr int W1 = F1;                      // frag (jt, s): 8 x 32
  static constexpr int B1 = W1 + 8 * 32 * 64;
  static constexpr int F1_LEN = 8 * 32 * 64 + 128;
  static constexpr int FS = F1 + F1_LEN;             // stage: sdf_out
  static constexpr int WS = FS;                      // rows 1..128: 8 x 32
  static constexpr int BS = WS + 8 * 32 * 64;        // bias of rows 1..128
  static constexpr int WS0 = BS + 128;               // row 0 (sdf) [128]
  static constexpr int BS0 = WS0 + 128;              // its bias (+3 pad)
  static constexpr int FS_LEN = 8 * 32 * 64 + 128 + 128 + 4;
  static constexpr int FC = FS + FS_LEN;             // stage: colour head
  static constexpr int WC = FC;                      // frag (jt, s): 8 x 36
  static constexpr int BC = WC + 8 * 36 * 64;
  static constexpr int WO = BC + 128;                // [3][128]
  static constexpr int BO = WO + 384;                // [3] (+1 pad)
  static constexpr int FC_LEN = 8 * 36 * 64 + 128 + 384 + 4;
  // ---- backward passes -------------------------------------------------------
  static constexpr int RC = FC + FC_LEN;             // stage: colour head
  static constexpr int WOB = RC;                     // copy of WO [3][128]
  static constexpr int WCT = WOB + 384;              // frag (kt, s): 9 x 32
  static constexpr int RC_LEN = 384 + 9 * 32 * 64;
  static constexpr int RS = RC + RC_LEN;             // stage: sdf_out
  static constexpr int WS0B = RS;                    // copy of row 0 [128]
  static constexpr int WST = WS0B + 128;             // frag (kt, s): 8 x 32
  static constexpr int RS_LEN = 128 + 8 * 32 * 64;
  static constexpr int R1 = RS + RS_LEN;             // stage: layer 1
  static constexpr int W1T = R1;                     // frag (kt, s): 8 x 32
  static constexpr int R1_LEN = 8 * 32 * 64;
  static constexpr int R0 = R1 + R1_LEN;             // stage: layer 0
  static constexpr int W0T = R0;                     // frag (kt = 0, s): 32
  static constexpr int R0_LEN = 32 * 64;
  static constexpr int LEN = R0 + R0_LEN;
  static constexpr int STAGE_MAX = FC_LEN;           // largest staged range
};

// input feature consumed at K-step s by lane group q of the colour layer:
// s < 32: f[kmap(s, q)], s >= 32: x[4q + s - 32]
__host__ __device__ constexpr int vox_color_in(int s, int q) {
  return s < 32 ? kmap(s, q) : 128 + 4 * q + (s - 32);
}

}  // namespace xrd
