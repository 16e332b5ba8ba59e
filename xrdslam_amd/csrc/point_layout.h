// Parameter layouts of Point-SLAM's colour path
// (slam/model_components/decoder_pointslam.py:276-291,293-542, the model's
// defaults: c_dim 32, hidden 128, 5 blocks, skip after the third, relative
// position encoded per neighbour, no view direction, no exposure code):
//   per neighbour k:  y_k = L2 softplus(L1 [e_rel(c_k - p), f_col_k])    F_theta
//                     e_rel = [sin, cos](2 pi x B_rel), B_rel [3,10] learnable
//   c = sum_k w_k y_k           (inverse squared-distance weights, L1-normalised)
//   h_0 = softplus(P0 e(p)) + FC0 c,  h_i = softplus(P_i in_i) + FC_i c,
//   in_3 = [e(p), h_2], e = [sin, cos](2 pi p B), B [3,20] fixed
//   rgb = sigmoid(OUT h_4)
// softplus: beta 100, threshold 20.
//
// "flat"   = the order engine/point.py concatenates the tensors in (and the
//            order of the flat gradient the kernels return).
// "packed" = MFMA fragments (nice_layout.h conventions: a fragment = 64
//            floats, lane l = (m = l & 15, q = l >> 4); activations in D
//            layout, K-step s of a hidden input consumes feature kmap(s, q),
//            of an embedding input feature 4s + q).
#pragma once
#include <stdint.h>

#include "nice_layout.h"  // kmap, emapT

namespace xrd {

struct PcFlat {
  // = the order of MLP_color.parameters(): back-to-back parameters let Adam
  // step the decoder with one launch on the flat gradient
  static constexpr int BREL = 0;                   // [3][10]
  static constexpr int W1 = BREL + 30;             // [128][52]
  static constexpr int B1 = W1 + 128 * 52;
  static constexpr int W2 = B1 + 128;              // [32][128]
  static constexpr int B2 = W2 + 32 * 128;
  static constexpr int FC = B2 + 32;               // 5 x ([128][32] + [128])
  static constexpr int fcw(int i) { return FC + i * (128 * 32 + 128); }
  static constexpr int fcb(int i) { return fcw(i) + 128 * 32; }
  static constexpr int P0W = FC + 5 * (128 * 32 + 128);   // [128][40]
  static constexpr int P0B = P0W + 128 * 40;
  static constexpr int P1W = P0B + 128;            // [128][128]
  static constexpr int P1B = P1W + 128 * 128;
  static constexpr int P2W = P1B + 128;
  static constexpr int P2B = P2W + 128 * 128;
  static constexpr int P3W = P2B + 128;            // [128][168] = [e 40 | h 128]
  static constexpr int P3B = P3W + 128 * 168;
  static constexpr int P4W = P3B + 128;
  static constexpr int P4B = P4W + 128 * 128;
  static constexpr int OW = P4B + 128;             // [3][128]
  static constexpr int OB = OW + 3 * 128;
  static constexpr int BEMB = OB + 3;              // [3][20] (not a parameter)
  static constexpr int LEN = BEMB + 60;
  static constexpr int N_GRAD = BEMB;              // gradient = [0, BEMB)
  static constexpr int pw(int i) {
    return i == 0 ? P0W : i == 1 ? P1W : i == 2 ? P2W : i == 3 ? P3W : P4W;
  }
  static constexpr int pb(int i) {
    return i == 0 ? P0B : i == 1 ? P1B : i == 2 ? P2B : i == 3 ? P3B : P4B;
  }
  static constexpr int pin(int i) { return i == 0 ? 40 : i == 3 ? 168 : 128; }
};

__host__ __device__ constexpr int pc_ksteps(int i) {
  return i == 0 ? 10 : i == 3 ? 42 : 32;
}
__host__ __device__ constexpr int pc_tlen(int i) {
  return 8 * pc_ksteps(i) * 64 + 128 + 8 * 8 * 64 + 128;
}
__host__ __device__ constexpr int pc_rlen(int i) {
  return (i >= 1 ? 8 * 32 * 64 : 0) + ((i == 0 || i == 3) ? 3 * 32 * 64 : 0) +
         2 * 32 * 64;
}

struct PcPack {
  // ---- F_theta forward (one stage) -------------------------------------------
  static constexpr int FT = 0;
  static constexpr int W1 = FT;                    // frag (jt 8, s 13)
  static constexpr int B1 = W1 + 8 * 13 * 64;
  static constexpr int W2 = B1 + 128;              // frag (jt 2, s 32)
  static constexpr int B2 = W2 + 2 * 32 * 64;
  static constexpr int BREL = B2 + 32;             // [10][4]: B, 0
  static constexpr int FT_LEN = 8 * 13 * 64 + 128 + 2 * 32 * 64 + 32 + 40;
  // ---- trunk forward, one stage per layer: W frag (jt 8, s S_i), bias,
  //      FC frag (jt 8, s 8), FC bias ------------------------------------------------
  static constexpr int ksteps(int i) { return pc_ksteps(i); }
  static constexpr int tlen(int i) { return pc_tlen(i); }
  static constexpr int T0 = FT + FT_LEN;
  static constexpr int T1 = T0 + pc_tlen(0);
  static constexpr int T2 = T1 + pc_tlen(1);
  static constexpr int T3 = T2 + pc_tlen(2);
  static constexpr int T4 = T3 + pc_tlen(3);
  static constexpr int tw(int i) {
    return i == 0 ? T0 : i == 1 ? T1 : i == 2 ? T2 : i == 3 ? T3 : T4;
  }
  static constexpr int tb(int i) { return tw(i) + 8 * ksteps(i) * 64; }
  static constexpr int tfc(int i) { return tb(i) + 128; }
  static constexpr int tfb(int i) { return tfc(i) + 8 * 8 * 64; }
  // output layer + the embedding matrix (read from global: small)
  static constexpr int OW = T4 + pc_tlen(4);         // [3][128]
  static constexpr int OB = OW + 384;              // [3] (+1)
  static constexpr int BEMB = OB + 4;              // [20][4]: B, 0
  static constexpr int FWD_LEN = BEMB + 80;
  static constexpr int STAGE_MAX = pc_tlen(3);       // largest staged range
  // ---- backward: one stage per trunk layer, then F_theta ------------------------
  //  layer i: WT frag (kt 8, s 32): d/d h_{i-1} (i >= 1)
  //           ET frag (kt 3, s 32): d/d e (i = 0, 3)
  //           FCT frag (kt 2, s 32): d/d c
  static constexpr int rlen(int i) { return pc_rlen(i); }
  static constexpr int R4 = FWD_LEN;
  static constexpr int R3 = R4 + pc_rlen(4);
  static constexpr int R2 = R3 + pc_rlen(3);
  static constexpr int R1 = R2 + pc_rlen(2);
  static constexpr int R0 = R1 + pc_rlen(1);
  static constexpr int rw(int i) {
    return i == 0 ? R0 : i == 1 ? R1 : i == 2 ? R2 : i == 3 ? R3 : R4;
  }
  static constexpr int ret(int i) { return rw(i) + (i >= 1 ? 8 * 32 * 64 : 0); }
  static constexpr int rfc(int i) {
    return ret(i) + ((i == 0 || i == 3) ? 3 * 32 * 64 : 0);
  }
  //  F_theta backward: W2T frag (kt 8, s 8), W1T_f frag (kt 2, s 32),
  //                    W1T_e frag (kt 2, s 32)
  static constexpr int RF = R0 + pc_rlen(0);
  static constexpr int W2T = RF;
  static constexpr int W1TF = W2T + 8 * 8 * 64;
  static constexpr int W1TE = W1TF + 2 * 32 * 64;
  static constexpr int RF_LEN = 8 * 8 * 64 + 4 * 32 * 64;
  static constexpr int LEN = RF + RF_LEN;
};

// input column of layer-1 of F_theta consumed at K-step s by lane group q
__host__ __device__ constexpr int pc_ft_in(int s, int q) {
  return s < 5 ? 4 * s + q : 20 + kmap(s - 5, q);
}
// input column of trunk layer i at K-step s
__host__ __device__ constexpr int pc_trunk_in(int i, int s, int q) {
  return i == 0 ? 4 * s + q
                : (i == 3 ? (s < 10 ? 4 * s + q : 40 + kmap(s - 10, q))
                          : kmap(s, q));
}

}  // namespace xrd
