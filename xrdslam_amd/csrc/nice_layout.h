// Parameter layouts of the NICE-SLAM decoders.
//
// "flat"   = the reference's state_dict order, concatenated
//            (slam/model_components/decoder_nice.py:145-186, 273-288):
//              fc_c.{0..4}.weight [32,CD], fc_c.{i}.bias [32]   (MLP only)
//              embedder._B [3,93]                               (MLP only)
//              pts_linears.{0..4}.weight/bias, output_linear.weight/bias
// "packed" = MFMA-fragment order consumed by the kernels.  A fragment is 64
//            floats, one per lane, holding the A operand of one
//            v_mfma_f32_16x16x4_f32 (lane l <-> row m=l&15, k-slot q=l>>4).
//
// Canonical feature layout of activations ("D layout"): lane (q=l>>4,i=l&15)
// holds, for point i of a 16-point tile, features 16*jt + 4*q + r (jt tile,
// r register) — exactly what the MFMA leaves in the accumulator when rows are
// output features.  K-step s of the next layer consumes feature
// kmap(s,q) = 16*(s>>2) + 4*q + (s&3), i.e. register (jt=s>>2, r=s&3): the
// chain of layers never leaves registers.
#pragma once
#include <stdint.h>

namespace xrd {

constexpr int kEmbK = 93;  // Gaussian Fourier features (decoder_nice.py:150)
constexpr int kEmbS = 24;  // K-steps of the padded (96) embedding

__host__ __device__ constexpr int kmap(int s, int q) {
  return 16 * (s >> 2) + 4 * q + (s & 3);
}
// embedding feature held by lane group q at K-step s (forward)
__host__ __device__ constexpr int emap(int s, int q) { return 4 * s + q; }
// embedding feature of row m of transposed tile kt (backward: D row m=4q+r of
// tile kt must be feature 4*(4kt+r)+q so that lane group q owns emap(4kt+r,q))
__host__ __device__ constexpr int emapT(int kt, int m) {
  return 4 * (4 * kt + (m & 3)) + (m >> 2);
}

template <int CD, int OD>
struct MlpFlat {
  static constexpr int fcw(int i) { return i * (32 * CD + 32); }
  static constexpr int fcb(int i) { return i * (32 * CD + 32) + 32 * CD; }
  static constexpr int EB = 5 * (32 * CD + 32);
  static constexpr int P0W = EB + 3 * kEmbK;
  static constexpr int P0B = P0W + 32 * kEmbK;
  static constexpr int P1W = P0B + 32;
  static constexpr int P1B = P1W + 1024;
  static constexpr int P2W = P1B + 32;
  static constexpr int P2B = P2W + 1024;
  static constexpr int P3W = P2B + 32;  // [32, 93+32]
  static constexpr int P3B = P3W + 32 * (kEmbK + 32);
  static constexpr int P4W = P3B + 32;
  static constexpr int P4B = P4W + 1024;
  static constexpr int OW = P4B + 32;
  static constexpr int OB = OW + OD * 32;
  static constexpr int LEN = OB + OD;
  // hidden-part weight of layer i (1..4): base, row stride, column offset
  static constexpr int pw(int i) {
    return i == 1 ? P1W : i == 2 ? P2W : i == 3 ? P3W : P4W;
  }
  static constexpr int pstride(int i) { return i == 3 ? kEmbK + 32 : 32; }
  static constexpr int pcol(int i) { return i == 3 ? kEmbK : 0; }
  static constexpr int pb(int i) {
    return i == 0 ? P0B : i == 1 ? P1B : i == 2 ? P2B : i == 3 ? P3B : P4B;
  }
};

template <int CD, int OD>
struct MlpPack {
  static constexpr int KC = CD / 4;    // K-steps of fc_c
  static constexpr int KTC = CD / 16;  // 16-feature tiles of c
  // forward-only fragments first, then what both passes read, then the
  // transposed (backward) fragments: the forward needs [0, WHT), the backward
  // [EMB, LEN) — each ONE contiguous range to stage into LDS
  static constexpr int W0 = 0;                     // frag (jt,s): 2*24
  static constexpr int W3E = W0 + 2 * kEmbS * 64;  // frag (jt,s): 2*24
  static constexpr int WH = W3E + 2 * kEmbS * 64;  // i=1..4: frag (jt,s) 2*8
  static constexpr int WC = WH + 4 * 1024;         // i=0..4: frag (jt,s) 2*KC
  static constexpr int B = WC + 5 * 2 * KC * 64;   // [5][32]
  static constexpr int BC = B + 160;               // [5][32]
  static constexpr int EMB = BC + 160;  // [96][4]: B[0][k],B[1][k],B[2][k],0
  static constexpr int WOUT = EMB + 96 * 4;        // [4][32] (rows >= OD zero)
  static constexpr int BOUT = WOUT + 128;          // [4]
  static constexpr int WHT = BOUT + 4;             // i=1..4: frag (kt,s) 2*8
  static constexpr int WCT = WHT + 4 * 1024;       // i=0..4: frag (kt,s) KTC*8
  static constexpr int W0T = WCT + 5 * KTC * 512;  // frag (kt,s) 6*8
  static constexpr int W3ET = W0T + 6 * 512;
  static constexpr int LEN = W3ET + 6 * 512;
  static constexpr int wh(int i) { return WH + (i - 1) * 1024; }
  static constexpr int wc(int i) { return WC + i * 2 * KC * 64; }
  static constexpr int wht(int i) { return WHT + (i - 1) * 1024; }
  static constexpr int wct(int i) { return WCT + i * KTC * 512; }
};

// MLP_no_xyz (coarse): layer 3 consumes cat[c, h] (64 inputs)
struct NoXyzFlat {
  static constexpr int pw(int i) {
    return i < 4 ? i * 1056 : 3 * 1056 + (2048 + 32);
  }
  static constexpr int pstride(int i) { return i == 3 ? 64 : 32; }
  static constexpr int pb(int i) { return pw(i) + 32 * pstride(i); }
  static constexpr int OW = 4 * 1056 + 2080;
  static constexpr int OB = OW + 32;
  static constexpr int LEN = OB + 1;
};
struct NoXyzPack {
  static constexpr int ks(int i) { return i == 3 ? 16 : 8; }  // K-steps
  static constexpr int w(int i) {                             // frag (jt,s)
    return (i <= 3 ? i * 1024 : 5 * 1024);
  }
  static constexpr int B = 6 * 1024;  // [5][32]
  static constexpr int WOUT = B + 160;
  static constexpr int BOUT = WOUT + 32;
  static constexpr int WT = BOUT + 4;  // transposed: i!=3: 2*8; i==3: 4*8
  static constexpr int wt(int i) {
    return WT + (i <= 3 ? i * 1024 : 5 * 1024);
  }
  static constexpr int LEN = WT + 6 * 1024;
};

}  // namespace xrd
