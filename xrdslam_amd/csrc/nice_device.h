// Device functions of the NICE-SLAM render kernels (gfx950), shared by
// nice_render.hip (forward / backward launches, point queries, Point-SLAM
// geometry path) and nice_map.hip (the fused mapping iteration): trilinear
// lookup and its backward, the MFMA decoder chains, depth-guided sampling,
// compositing backward, the colour decoder's weight-gradient exchange.
// Reference behaviour restated (never copied): slam/models/conv_onet.py:339-524,
// slam/model_components/decoder_nice.py:195-234,297-320,386-414,
// slam/model_components/utils.py:189-244.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"
#include "nice_layout.h"

namespace xrd {
namespace {

// scheduling fence: keeps hipcc from hoisting a whole phase's weight-fragment
// loads (and their VGPRs) across phases
#define XRD_SB() __builtin_amdgcn_sched_barrier(0)

// ---------------------------------------------------------------------------
// device: trilinear lookup (torch grid_sample bilinear/border/align_corners)
// ---------------------------------------------------------------------------
struct Tri {
  int off[8];     // float offset of the 8 corners (cell * 32)
  float w[8];     // corner weights, torch order tnw,tne,tsw,tse,bnw,bne,bsw,bse
  float wa[3][2]; // per-axis weights (x,y,z)(lo,hi)
  float mult[3];  // d(grid coord)/d(normalised coord), 0 when clipped
  double inv[3];  // d(normalised)/d(world) = 2/(b1-b0)
};

__device__ __forceinline__ void axis_prepare(double p, double b0, double b1,
                                             int N, int& i0, int& i1,
                                             float& w0, float& w1,
                                             float& mult, double& inv) {
  const double ext = b1 - b0;
  inv = 2.0 / ext;
  float xn = (float)(((p - b0) / ext) * 2.0 - 1.0);
  float ix = ((xn + 1.f) / 2.f) * (float)(N - 1);
  float gm = (float)(N - 1) / 2.f;
  const float mx = (float)(N - 1);
  if (!(ix > 0.f)) {
    ix = 0.f;
    gm = 0.f;
  } else if (ix >= mx) {
    ix = mx;
    gm = 0.f;
  }
  const float f = floorf(ix);
  i0 = (int)f;
  i1 = i0 + 1;
  w1 = ix - f;
  w0 = (f + 1.f) - ix;
  if (i1 > N - 1) {  // torch skips the out-of-range corner (its weight is 0)
    i1 = N - 1;
    w1 = 0.f;
  }
  mult = gm;
}

__device__ __forceinline__ void tri_prepare(const double (&p)[3],
                                            const double* bd, double scale,
                                            const int* dim, Tri& t) {
  const int Z = dim[0], Y = dim[1], X = dim[2];
  int x0, x1, y0, y1, z0, z1;
  axis_prepare(p[0], bd[0] * scale, bd[1] * scale, X, x0, x1, t.wa[0][0],
               t.wa[0][1], t.mult[0], t.inv[0]);
  axis_prepare(p[1], bd[2] * scale, bd[3] * scale, Y, y0, y1, t.wa[1][0],
               t.wa[1][1], t.mult[1], t.inv[1]);
  axis_prepare(p[2], bd[4] * scale, bd[5] * scale, Z, z0, z1, t.wa[2][0],
               t.wa[2][1], t.mult[2], t.inv[2]);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int xi = (c & 1) ? x1 : x0, yi = (c & 2) ? y1 : y0,
              zi = (c & 4) ? z1 : z0;
    t.off[c] = ((zi * Y + yi) * X + xi) * 32;
    t.w[c] = (t.wa[0][c & 1] * t.wa[1][(c >> 1) & 1]) * t.wa[2][(c >> 2) & 1];
  }
}

// tri_prepare in two halves.  The float64 normalisation of a point (a
// subtraction, a DIVISION — ~40 instructions —, a multiply-add and the
// rounding to float, per axis) depends on the bound only, and the middle,
// fine and colour grids share it: tri_norm once per sample, tri_prepare_n per
// lookup (float arithmetic + the eight corner offsets).  The one-launch
// mapping iteration prepared the same point six times (three gathers, three
// scatters): 18 float64 divisions a sample, more instructions than a
// decoder's VALU work — and on gfx950 the f32 MFMAs run on the VALU's lanes
// (DESIGN 4.1f), so every VALU instruction is time the MFMAs do not get.
// Same operations in the same order as axis_prepare: identical cells/weights.
__device__ __forceinline__ void tri_norm(const double (&p)[3], const double* bd,
                                         float (&xn)[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double b0 = bd[2 * a] * 1.0, b1 = bd[2 * a + 1] * 1.0;
    const double ext = b1 - b0;
    xn[a] = (float)(((p[a] - b0) / ext) * 2.0 - 1.0);
  }
}
__device__ __forceinline__ void axis_cell(float xn, int N, int& i0, int& i1,
                                          float& w0, float& w1, float& mult) {
  float ix = ((xn + 1.f) / 2.f) * (float)(N - 1);
  float gm = (float)(N - 1) / 2.f;
  const float mx = (float)(N - 1);
  if (!(ix > 0.f)) {
    ix = 0.f;
    gm = 0.f;
  } else if (ix >= mx) {
    ix = mx;
    gm = 0.f;
  }
  const float f = floorf(ix);
  i0 = (int)f;
  i1 = i0 + 1;
  w1 = ix - f;
  w0 = (f + 1.f) - ix;
  if (i1 > N - 1) {  // torch skips the out-of-range corner (its weight is 0)
    i1 = N - 1;
    w1 = 0.f;
  }
  mult = gm;
}
// NEED_INV: t.inv (d normalised / d world, a float64 division per axis) is
// only read by tri_backward_dp
template <bool NEED_INV>
__device__ __forceinline__ void tri_prepare_n(const float (&xn)[3],
                                              const double* bd,
                                              const int* dim, Tri& t) {
  const int Z = dim[0], Y = dim[1], X = dim[2];
  int x0, x1, y0, y1, z0, z1;
  axis_cell(xn[0], X, x0, x1, t.wa[0][0], t.wa[0][1], t.mult[0]);
  axis_cell(xn[1], Y, y0, y1, t.wa[1][0], t.wa[1][1], t.mult[1]);
  axis_cell(xn[2], Z, z0, z1, t.wa[2][0], t.wa[2][1], t.mult[2]);
  if (NEED_INV) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
      t.inv[a] = 2.0 / (bd[2 * a + 1] * 1.0 - bd[2 * a] * 1.0);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int xi = (c & 1) ? x1 : x0, yi = (c & 2) ? y1 : y0,
              zi = (c & 4) ? z1 : z0;
    t.off[c] = ((zi * Y + yi) * X + xi) * 32;
    t.w[c] = (t.wa[0][c & 1] * t.wa[1][(c >> 1) & 1]) * t.wa[2][(c >> 2) & 1];
  }
}

// gather the lane's 8 channels (16*kt + 4*q + r) of a 32-channel cell
__device__ __forceinline__ void tri_gather(const float* __restrict__ grid,
                                           const Tri& t, int q,
                                           f32x4 (&c)[2]) {
  c[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  c[1] = c[0];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(grid + t.off[k] + 4 * q);
    const f32x4 b =
        *reinterpret_cast<const f32x4*>(grid + t.off[k] + 16 + 4 * q);
    c[0] += a * t.w[k];
    c[1] += b * t.w[k];
  }
}

// backward of one lookup, coordinate part: accumulates d(loss)/d(world p)
// (summed over this lane's 8 channels only; the caller reduces over the 4
// lane groups).  The scatter into the grid gradient is grid_scatter().
__device__ __forceinline__ void tri_backward_dp(const float* __restrict__ grid,
                                                const Tri& t, int q,
                                                const f32x4 (&gc)[2],
                                                double (&gp)[3]) {
  float gi[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(grid + t.off[k] + 4 * q);
    const f32x4 b =
        *reinterpret_cast<const f32x4*>(grid + t.off[k] + 16 + 4 * q);
    float dot = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) dot += a[r] * gc[0][r] + b[r] * gc[1][r];
    const float sx = (k & 1) ? 1.f : -1.f, sy = (k & 2) ? 1.f : -1.f,
                sz = (k & 4) ? 1.f : -1.f;
    // torch skips corners that were out of range; those have weight 0 on
    // their own axis and the coordinate gradient multiplier is 0 there.
    gi[0] += sx * dot * t.wa[1][(k >> 1) & 1] * t.wa[2][(k >> 2) & 1];
    gi[1] += sy * dot * t.wa[0][k & 1] * t.wa[2][(k >> 2) & 1];
    gi[2] += sz * dot * t.wa[0][k & 1] * t.wa[1][(k >> 1) & 1];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) gp[a] += (double)(t.mult[a] * gi[a]) * t.inv[a];
}

// Scatter-add of a tile's feature gradients into the channel-last grid.
// Measured on MI355X (tools/ubench/atomics.hip): f32 atomics issued as
// "32 consecutive channels of one cell per half-wave" run 4-6x faster than the
// accumulator's native (4 lanes x strided dwords per point) pattern, and
// consecutive samples of a ray share cells, so: transpose the tile through
// LDS, then each half-wave walks 8 consecutive points, merges runs that hit
// the same cell in registers and issues one fully coalesced 128-B atomic per
// (run, corner).
//
// Round 6 (the scatters were 3 x 9 us of the 175 us mapping block, and on
// gfx950 that is instruction issue: a SIMD's VALU instructions and MFMAs add
// up): the run structure is worked out ONCE, in parallel — the
// 16 point lanes compare their corner offsets with the next point's, eight
// ballots give "the run of corner k ends at point i" as wave-uniform bit
// masks — instead of by every lane at every (point, corner) step (an offset
// read, a compare against a tracked current cell, two conditional moves).
// The walk is a multiply-add a step; a flush block is skipped by a scalar
// branch when neither half-wave ends a run there; offsets and weights of a
// point come in as four 16-byte LDS reads instead of sixteen 4-byte ones.
// Runs, flush order and sums are those of rounds 2-5.
struct ScatterLds {
  float* gt;   // [16][33] transposed gradients (point-major)
  int* off;    // [16][8]
  float* w;    // [16][8]
};
constexpr int kScatterFloats = 16 * 33 + 16 * 8 + 16 * 8;

__device__ __forceinline__ void grid_scatter(float* __restrict__ ggrid,
                                             const uint8_t* __restrict__ cmask,
                                             const Tri& t, int lane,
                                             const f32x4 (&gc)[2],
                                             const ScatterLds& S) {
  if (ggrid == nullptr) return;
  const int q = lane >> 4, i = lane & 15;
  wave_lds_sync();
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) S.gt[i * 33 + 16 * kt + 4 * q + r] = gc[kt][r];
  // the frustum selection (cmask: one byte a cell, 0 = the cell's gradient is
  // dropped) is looked up HERE, once per (point, corner) and all of them in
  // flight together, and folded into the offset's sign: ~off (negative) for
  // a masked cell.  Rounds 2-5 read the byte inside the run loop — a
  // dependent global load in front of every flush (the frame loop always
  // passes a selection; the timing tools did not, which hid it).
  if (q == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int o = t.off[k];
      const bool keep = cmask == nullptr || cmask[o >> 5] != 0;
      S.off[i * 8 + k] = keep ? o : ~o;
      S.w[i * 8 + k] = t.w[k];
    }
  }
  // ends[k]: bit i = the run of corner k ends at point i (the next point of
  // the half-wave sits in another cell, or i is the half's last point);
  // lanes 0..15 are the point lanes (q == 0), so the ballot's low 16 bits
  // are the points
  uint32_t ends[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int nxt = __shfl_down(t.off[k], 1, 16);
    const bool e = (i & 7) == 7 || nxt != t.off[k];
    ends[k] = (uint32_t)__ballot(q == 0 && e) & 0xffffu;
  }
  wave_lds_sync();
  const int half = lane >> 5, ch = lane & 31;
  // this half's byte of each mask: bit j = point 8 half + j
  uint32_t mine[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) mine[k] = (ends[k] >> (8 * half)) & 0xffu;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const float* gt = S.gt + half * 8 * 33 + ch;
  const int* po = S.off + half * 64;
  const float* pw = S.w + half * 64;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = gt[j * 33];
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(pw + j * 8);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(pw + j * 8 + 4);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 o0 = *reinterpret_cast<const i32x4*>(po + j * 8);
    const i32x4 o1 = *reinterpret_cast<const i32x4*>(po + j * 8 + 4);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float wk = k < 4 ? w0[k & 3] : w1[k & 3];
      const int o = k < 4 ? o0[k & 3] : o1[k & 3];
      acc[k] = fmaf(wk, v, acc[k]);
      // (wave-uniform: does either half end a run of corner k here?)
      if ((ends[k] >> j) & 0x101u) {
        const bool end = (mine[k] >> j) & 1u;
        if (end && o >= 0 && acc[k] != 0.f) atomicAdd(ggrid + o + ch, acc[k]);
        acc[k] = end ? 0.f : acc[k];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// device: MLP decoder forward (decoder_nice.py:207-234)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float embed_arg(const float (&p)[3], const f32x4 b) {
  float a = p[0] * b[0];
  a = fmaf(p[1], b[1], a);
  a = fmaf(p[2], b[2], a);
  return a;
}

// EMIT_H: the layer outputs h_0..h_4 are written to ``hsc`` as five
// feature-major matrices [32 features][16 points] (512 floats each): the
// operand layout of the deferred weight-gradient contraction (nice_map.hip)
template <int NT, int CD, int OD, bool SAVE_MASK, bool SAVE_H,
          bool EMIT_H = false>
__device__ __forceinline__ void mlp_fwd(const float* __restrict__ pk, int lane,
                                        const float (&p)[NT][3],
                                        const f32x4 (&c)[NT][CD / 16],
                                        float (&out)[NT][OD],
                                        uint64_t (&mask)[NT],
                                        f32x4 (*hs)[2],
                                        float* __restrict__ hsc = nullptr) {
  static_assert(!EMIT_H || NT == 1, "h output: one tile per wave");
  // SAVE_H: the layer outputs h_0..h_4 are returned in hs[5][2] (registers:
  // the layer loop is then unrolled so that the indices are static)
  using P = MlpPack<CD, OD>;
  static_assert(!SAVE_H || NT == 1, "h output: one tile per wave");
  const int q = lane >> 4;
  f32x4 acc[NT][2], acc3[NT][2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const f32x4 b0 =
        *reinterpret_cast<const f32x4*>(pk + P::B + 0 * 32 + 16 * jt + 4 * q);
    const f32x4 b3 =
        *reinterpret_cast<const f32x4*>(pk + P::B + 3 * 32 + 16 * jt + 4 * q);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t][jt] = b0;
      acc3[t][jt] = b3;
    }
  }
  // Fourier embedding feeds layer 0 and the skip part of layer 3
#pragma unroll 4
  for (int s = 0; s < kEmbS; ++s) {
    const f32x4 bk =
        *reinterpret_cast<const f32x4*>(pk + P::EMB + emap(s, q) * 4);
    float a0[2], a3[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      a0[jt] = pk[P::W0 + (jt * kEmbS + s) * 64 + lane];
      a3[jt] = pk[P::W3E + (jt * kEmbS + s) * 64 + lane];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float e = sin_cw(embed_arg(p[t], bk));
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        acc[t][jt] = XRD_MFMA4(a0[jt], e, acc[t][jt]);
        acc3[t][jt] = XRD_MFMA4(a3[jt], e, acc3[t][jt]);
      }
    }
  }
  f32x4 h[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) mask[t] = 0;
  constexpr int kLayerUnroll = SAVE_H ? 5 : 1;
#pragma unroll kLayerUnroll
  for (int i = 0; i < 5; ++i) {
    // cc = fc_c[i](c)
    f32x4 cc[NT][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const f32x4 bc = *reinterpret_cast<const f32x4*>(pk + P::BC + i * 32 +
                                                       16 * jt + 4 * q);
#pragma unroll
      for (int t = 0; t < NT; ++t) cc[t][jt] = bc;
    }
#pragma unroll
    for (int s = 0; s < P::KC; ++s) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const float a = pk[P::wc(i) + (jt * P::KC + s) * 64 + lane];
#pragma unroll
        for (int t = 0; t < NT; ++t)
          cc[t][jt] = XRD_MFMA4(a, c[t][s >> 2][s & 3], cc[t][jt]);
      }
      if ((s & 7) == 7) XRD_SB();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      // the layer's 8 ReLU bits are collected in a 32-bit word with constant
      // shifts; ONE 64-bit shift a layer places them (the layer index is a
      // run-time value: 8 variable 64-bit shifts a layer were measurable in
      // the one-wave-a-SIMD tracking kernels)
      uint32_t m8 = 0;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = acc[t][jt][r];
          if (SAVE_MASK && a > 0.f) m8 |= 1u << (jt * 4 + r);
          h[t][jt][r] = fmaxf(a, 0.f) + cc[t][jt][r];
        }
        if (SAVE_H) hs[i][jt] = h[t][jt];
        if (EMIT_H) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            hsc[i * 512 + (16 * jt + 4 * q + r) * 16 + (lane & 15)] =
                h[t][jt][r];
        }
      }
      if (SAVE_MASK) mask[t] |= (uint64_t)m8 << (i * 8);
    }
    if (i < 4) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(
            pk + P::B + (i + 1) * 32 + 16 * jt + 4 * q);
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t][jt] = (i + 1 == 3) ? acc3[t][jt] : b;
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const float a = pk[P::wh(i + 1) + (jt * 8 + s) * 64 + lane];
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t][jt] = XRD_MFMA4(a, h[t][s >> 2][s & 3], acc[t][jt]);
        }
      }
      XRD_SB();
    }
  }
  // output layer on the VALU + reduction over the 4 lane groups
#pragma unroll
  for (int o = 0; o < OD; ++o) {
    const f32x4 w0 =
        *reinterpret_cast<const f32x4*>(pk + P::WOUT + o * 32 + 4 * q);
    const f32x4 w1 =
        *reinterpret_cast<const f32x4*>(pk + P::WOUT + o * 32 + 16 + 4 * q);
    const float bo = pk[P::BOUT + o];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) v += w0[r] * h[t][0][r] + w1[r] * h[t][1][r];
      out[t][o] = group4_sum(v) + bo;
    }
  }
}

// mlp_fwd for one tile a wave with the A fragments READ AHEAD (round 6, the
// mapping kernel).  mlp_fwd leaves the placement of its ds_read_b32 to the
// compiler, which keeps each group of reads right in front of the MFMAs that
// consume it: an exposed LDS round trip per group of 8-16 MFMAs, and after a
// staging barrier the three waves of a SIMD run in step, so all of them wait
// at the same time (phase stamps, tools/nice_map_stamps.py: the slowest wave
// of a block needs 17 / 21 / 18 us for the middle / fine / colour forward
// against 9.6 / 13.4 / 9.6 us of MFMA issue).  Here the layer loop is
// unrolled and every batch of fragments is requested one MFMA batch early:
// fc_c.(i+1)'s fragments before the 16 MFMAs of pts_linears.(i+1), those of
// pts_linears.(i+2) before the MFMAs of fc_c.(i+1).  Same MFMA order per
// accumulator as mlp_fwd; the outputs differ from mlp_fwd's by rounding only
// (<= 1e-6: the compiler contracts the VALU output layer differently).
template <int CD, int OD, bool SAVE_MASK, bool EMIT_H>
__device__ __forceinline__ void mlp_fwd_ra(const float* __restrict__ pk,
                                           int lane, const float (&p)[3],
                                           const f32x4 (&c)[CD / 16],
                                           float (&out)[OD], uint64_t& mask,
                                           float* __restrict__ hsc) {
  using P = MlpPack<CD, OD>;
  constexpr int KC = P::KC;
  const int q = lane >> 4;
  f32x4 acc[2], acc3[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    acc[jt] =
        *reinterpret_cast<const f32x4*>(pk + P::B + 0 * 32 + 16 * jt + 4 * q);
    acc3[jt] =
        *reinterpret_cast<const f32x4*>(pk + P::B + 3 * 32 + 16 * jt + 4 * q);
  }
  // Fourier embedding feeds layer 0 and the skip part of layer 3.  A K-step
  // is one sine (~26 VALU instructions) and 4 MFMAs; written step by step a
  // wave runs "26 VALU, then 4 MFMAs" and the three waves of a SIMD, in step
  // behind the staging barrier, do not fill each other's gaps (stamps: the 96
  // MFMAs of this loop took as long as the 144 of the layers).  So the loop
  // is software-pipelined by hand — step s+1's sine is computed in the
  // shadow of step s's MFMAs, fragments and B rows are requested a step
  // ahead — with one MFMA between two stages of the sine (sin_stage_a..d).
  // (EMB row 96 + q of the last step's look-ahead lies in the WOUT rows
  // behind the table: a finite dummy whose sine is never used)
  f32x4 bkn = *reinterpret_cast<const f32x4*>(pk + P::EMB + emap(1, q) * 4);
  float e = sin_cw(embed_arg(
      p, *reinterpret_cast<const f32x4*>(pk + P::EMB + emap(0, q) * 4)));
  float a0[2], a3[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    a0[jt] = pk[P::W0 + (jt * kEmbS) * 64 + lane];
    a3[jt] = pk[P::W3E + (jt * kEmbS) * 64 + lane];
  }
  XRD_SB();
#pragma unroll 4
  for (int s = 0; s < kEmbS; ++s) {
    const int sn = s + 1 < kEmbS ? s + 1 : s;   // (last step: re-read, unused)
    float n0[2], n3[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      n0[jt] = pk[P::W0 + (jt * kEmbS + sn) * 64 + lane];
      n3[jt] = pk[P::W3E + (jt * kEmbS + sn) * 64 + lane];
    }
    const f32x4 bk2 =
        *reinterpret_cast<const f32x4*>(pk + P::EMB + emap(s + 2, q) * 4);
    // one MFMA of step s, one stage of step s+1's sine, ... (full scheduling
    // fences: the order below is the issue order)
    SinStages st;
    XRD_SB();
    acc[0] = XRD_MFMA4(a0[0], e, acc[0]);
    XRD_SB();
    sin_stage_a(embed_arg(p, bkn), st);
    XRD_SB();
    acc3[0] = XRD_MFMA4(a3[0], e, acc3[0]);
    XRD_SB();
    sin_stage_b(st);
    XRD_SB();
    acc[1] = XRD_MFMA4(a0[1], e, acc[1]);
    XRD_SB();
    sin_stage_c(st);
    XRD_SB();
    acc3[1] = XRD_MFMA4(a3[1], e, acc3[1]);
    XRD_SB();
    const float en = sin_stage_d(st);
    XRD_SB();
    e = en;
    bkn = bk2;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      a0[jt] = n0[jt];
      a3[jt] = n3[jt];
    }
  }
  float ac[2][KC];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int s = 0; s < KC; ++s)
      ac[jt][s] = pk[P::wc(0) + (jt * KC + s) * 64 + lane];
  XRD_SB();
  f32x4 h[2];
  mask = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    // (accumulator start values are read BEFORE a batch of fragments is
    // requested: LDS reads return in order, a bias read behind the batch
    // would make its consumer wait for the whole batch)
    f32x4 cc[2], accn[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      cc[jt] = *reinterpret_cast<const f32x4*>(pk + P::BC + i * 32 + 16 * jt +
                                               4 * q);
      if (i < 4 && i + 1 != 3)
        accn[jt] = *reinterpret_cast<const f32x4*>(pk + P::B + (i + 1) * 32 +
                                                   16 * jt + 4 * q);
    }
    float ah[2][8];
    if (i < 4) {  // pts_linears.(i+1): lands under the MFMAs of fc_c.i
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int s = 0; s < 8; ++s)
          ah[jt][s] = pk[P::wh(i + 1) + (jt * 8 + s) * 64 + lane];
    }
    XRD_SB();
#pragma unroll
    for (int s = 0; s < KC; ++s)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
        cc[jt] = XRD_MFMA4(ac[jt][s], c[s >> 2][s & 3], cc[jt]);
    XRD_SB();
    if (i < 4) {  // fc_c.(i+1): lands under the MFMAs of pts_linears.(i+1)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int s = 0; s < KC; ++s)
          ac[jt][s] = pk[P::wc(i + 1) + (jt * KC + s) * 64 + lane];
    }
    uint32_t m8 = 0;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = acc[jt][r];
        if (SAVE_MASK && a > 0.f) m8 |= 1u << (jt * 4 + r);
        h[jt][r] = fmaxf(a, 0.f) + cc[jt][r];
      }
      if (EMIT_H) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          hsc[i * 512 + (16 * jt + 4 * q + r) * 16 + (lane & 15)] = h[jt][r];
      }
    }
    if (SAVE_MASK) mask |= (uint64_t)m8 << (i * 8);
    if (i < 4) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
        acc[jt] = (i + 1 == 3) ? acc3[jt] : accn[jt];
      XRD_SB();
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
          acc[jt] = XRD_MFMA4(ah[jt][s], h[s >> 2][s & 3], acc[jt]);
      XRD_SB();
    }
  }
  // output layer on the VALU + reduction over the 4 lane groups
#pragma unroll
  for (int o = 0; o < OD; ++o) {
    const f32x4 w0 =
        *reinterpret_cast<const f32x4*>(pk + P::WOUT + o * 32 + 4 * q);
    const f32x4 w1 =
        *reinterpret_cast<const f32x4*>(pk + P::WOUT + o * 32 + 16 + 4 * q);
    const float bo = pk[P::BOUT + o];
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) v += w0[r] * h[0][r] + w1[r] * h[1][r];
    out[o] = group4_sum(v) + bo;
  }
}

// MLP_no_xyz (coarse, decoder_nice.py:308-320): h=c; 5x(Linear+ReLU), skip
// cat[c,h] after layer 2; Linear(32,1).
// One layer, its index a template parameter: the layer's A fragments (16 or
// 32 ds_read_b32) are all issued before the first MFMA.  With the layer index
// a run-time value (rounds 1-4) every MFMA waited for its own fragment read —
// a chain of 16-32 LDS latencies a layer: the coarse mapper's forward was
// 11.8 k cycles for 96 MFMAs (phase stamps, profiles/r05_scatter_contention.txt).
template <int NT, bool SAVE_MASK, int I>
__device__ __forceinline__ void noxyz_fwd_layer(
    const float* __restrict__ pk, int lane, const f32x4 (&c)[NT][2],
    f32x4 (&h)[NT][2], uint64_t (&mask)[NT]) {
  using P = NoXyzPack;
  constexpr int KS = P::ks(I);
  const int q = lane >> 4;
  float a[2][KS];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int s = 0; s < KS; ++s)
      a[jt][s] = pk[P::w(I) + (jt * KS + s) * 64 + lane];
  f32x4 acc[NT][2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const f32x4 b =
        *reinterpret_cast<const f32x4*>(pk + P::B + I * 32 + 16 * jt + 4 * q);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][jt] = b;
  }
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        // layer 3: K-steps 0..7 read c, 8..15 read h
        const float b = (I == 3 && s < 8) ? c[t][(s & 7) >> 2][s & 3]
                                          : h[t][(s & 7) >> 2][s & 3];
        acc[t][jt] = XRD_MFMA4(a[jt][s], b, acc[t][jt]);
      }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[t][jt][r];
        if (SAVE_MASK && v > 0.f)
          mask[t] |= (uint64_t)1 << (I * 8 + jt * 4 + r);
        h[t][jt][r] = fmaxf(v, 0.f);
      }
}

template <int NT, bool SAVE_MASK>
__device__ __forceinline__ void noxyz_fwd(const float* __restrict__ pk,
                                          int lane, const f32x4 (&c)[NT][2],
                                          float (&out)[NT],
                                          uint64_t (&mask)[NT]) {
  using P = NoXyzPack;
  const int q = lane >> 4;
  f32x4 h[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    h[t][0] = c[t][0];
    h[t][1] = c[t][1];
    mask[t] = 0;
  }
  noxyz_fwd_layer<NT, SAVE_MASK, 0>(pk, lane, c, h, mask);
  noxyz_fwd_layer<NT, SAVE_MASK, 1>(pk, lane, c, h, mask);
  noxyz_fwd_layer<NT, SAVE_MASK, 2>(pk, lane, c, h, mask);
  noxyz_fwd_layer<NT, SAVE_MASK, 3>(pk, lane, c, h, mask);
  noxyz_fwd_layer<NT, SAVE_MASK, 4>(pk, lane, c, h, mask);
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(pk + P::WOUT + 4 * q);
  const f32x4 w1 = *reinterpret_cast<const f32x4*>(pk + P::WOUT + 16 + 4 * q);
  const float bo = pk[P::BOUT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) v += w0[r] * h[t][0][r] + w1[r] * h[t][1][r];
    out[t] = group4_sum(v) + bo;
  }
}

// ---------------------------------------------------------------------------
// device: MLP decoder backward
// ---------------------------------------------------------------------------
template <int NT, int CD, int OD, bool NEED_E, bool NEED_DP>
__device__ __forceinline__ void mlp_bwd(
    const float* __restrict__ pk, int lane, const float (&p)[NT][3],
    const f32x4 (&c)[NT][CD / 16], const float (&gout)[NT][OD],
    const uint64_t (&mask)[NT], f32x4 (&gc)[NT][CD / 16],
    float (&gp)[NT][3]) {
  using P = MlpPack<CD, OD>;
  const int q = lane >> 4;
  f32x4 gh[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    gh[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    gh[t][1] = gh[t][0];
#pragma unroll
    for (int kt = 0; kt < CD / 16; ++kt) gc[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int o = 0; o < OD; ++o) {
    const f32x4 w0 =
        *reinterpret_cast<const f32x4*>(pk + P::WOUT + o * 32 + 4 * q);
    const f32x4 w1 =
        *reinterpret_cast<const f32x4*>(pk + P::WOUT + o * 32 + 16 + 4 * q);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      gh[t][0] += w0 * gout[t][o];
      gh[t][1] += w1 * gout[t][o];
    }
  }
  // masked gradients entering layers 3 and 0: the Fourier features feed both
  // (kept until the embedding backward after the loop; 16 registers instead
  // of 24 accumulators live across the whole layer loop)
  f32x4 ga3[NT][2], ga0[NT][2];
#pragma unroll 1
  for (int i = 4; i >= 0; --i) {
    f32x4 ga[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool on = (mask[t] >> (i * 8 + jt * 4 + r)) & 1;
          ga[t][jt][r] = on ? gh[t][jt][r] : 0.f;
        }
    }
    // g_c += Wc_i^T gh
#pragma unroll
    for (int kt = 0; kt < P::KTC; ++kt)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float a = pk[P::wct(i) + (kt * 8 + s) * 64 + lane];
#pragma unroll
        for (int t = 0; t < NT; ++t)
          gc[t][kt] = XRD_MFMA4(a, gh[t][s >> 2][s & 3], gc[t][kt]);
        if (s == 7) XRD_SB();
      }
    if (NEED_E) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          if (i == 3) ga3[t][jt] = ga[t][jt];
          if (i == 0) ga0[t][jt] = ga[t][jt];
        }
    }
    if (i >= 1) {
      f32x4 gprev[NT][2];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        gprev[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        gprev[t][1] = gprev[t][0];
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const float a = pk[P::wht(i) + (kt * 8 + s) * 64 + lane];
#pragma unroll
          for (int t = 0; t < NT; ++t)
            gprev[t][kt] = XRD_MFMA4(a, ga[t][s >> 2][s & 3], gprev[t][kt]);
          if (s == 7) XRD_SB();
        }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        gh[t][0] = gprev[t][0];
        gh[t][1] = gprev[t][1];
      }
    }
  }
  if (NEED_E) {
    // d loss / d sin(p.B) = W0^T ga0 + W3e^T ga3, one 16-feature tile at a
    // time; then through sin: lane group q owns feature k = emap(4kt+r, q)
#pragma unroll 1
    for (int kt = 0; kt < 6; ++kt) {
      f32x4 ge[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) ge[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float a3 = pk[P::W3ET + (kt * 8 + s) * 64 + lane];
        const float a0 = pk[P::W0T + (kt * 8 + s) * 64 + lane];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          ge[t] = XRD_MFMA4(a3, ga3[t][s >> 2][s & 3], ge[t]);
          ge[t] = XRD_MFMA4(a0, ga0[t][s >> 2][s & 3], ge[t]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = emap(4 * kt + r, q);
        const f32x4 bk = *reinterpret_cast<const f32x4*>(pk + P::EMB + k * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float garg = ge[t][r] * cos_cw(embed_arg(p[t], bk));
          if (NEED_DP) {
#pragma unroll
            for (int a = 0; a < 3; ++a) gp[t][a] += garg * bk[a];
          }
        }
      }
    }
  }
}

// mlp_bwd for one tile a wave with the A fragments read ahead (round 6, see
// mlp_fwd_ra): the layer loop is unrolled; pts_linears.i's transposed
// fragments are requested before the MFMAs of fc_c.i's, fc_c.(i-1)'s before
// the MFMAs of pts_linears.i; the embedding backward requests column tile
// kt+1 before the MFMAs of tile kt.  Same MFMA order per accumulator as
// mlp_bwd (results equal to rounding).
template <int CD, int OD, bool NEED_E, bool NEED_DP>
__device__ __forceinline__ void mlp_bwd_ra(
    const float* __restrict__ pk, int lane, const float (&p)[3],
    const float (&gout)[OD], const uint64_t mask, f32x4 (&gc)[CD / 16],
    float (&gp)[3]) {
  using P = MlpPack<CD, OD>;
  constexpr int KTC = P::KTC;
  const int q = lane >> 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 gh[2] = {z4, z4};
#pragma unroll
  for (int kt = 0; kt < KTC; ++kt) gc[kt] = z4;
  float awc[KTC][8];
#pragma unroll
  for (int kt = 0; kt < KTC; ++kt)
#pragma unroll
    for (int s = 0; s < 8; ++s)
      awc[kt][s] = pk[P::wct(4) + (kt * 8 + s) * 64 + lane];
#pragma unroll
  for (int o = 0; o < OD; ++o) {
    const f32x4 w0 =
        *reinterpret_cast<const f32x4*>(pk + P::WOUT + o * 32 + 4 * q);
    const f32x4 w1 =
        *reinterpret_cast<const f32x4*>(pk + P::WOUT + o * 32 + 16 + 4 * q);
    gh[0] += w0 * gout[o];
    gh[1] += w1 * gout[o];
  }
  f32x4 ga3[2] = {z4, z4}, ga0[2] = {z4, z4};
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    float awh[2][8];
    if (i >= 1) {  // lands under the MFMAs of fc_c.i
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 8; ++s)
          awh[kt][s] = pk[P::wht(i) + (kt * 8 + s) * 64 + lane];
    }
    f32x4 ga[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on = (mask >> (i * 8 + jt * 4 + r)) & 1;
        ga[jt][r] = on ? gh[jt][r] : 0.f;
      }
    XRD_SB();
    // g_c += Wc_i^T gh
#pragma unroll
    for (int kt = 0; kt < KTC; ++kt)
#pragma unroll
      for (int s = 0; s < 8; ++s)
        gc[kt] = XRD_MFMA4(awc[kt][s], gh[s >> 2][s & 3], gc[kt]);
    XRD_SB();
    if (i >= 1) {  // fc_c.(i-1): lands under the MFMAs of pts_linears.i
#pragma unroll
      for (int kt = 0; kt < KTC; ++kt)
#pragma unroll
        for (int s = 0; s < 8; ++s)
          awc[kt][s] = pk[P::wct(i - 1) + (kt * 8 + s) * 64 + lane];
    }
    if (NEED_E) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        if (i == 3) ga3[jt] = ga[jt];
        if (i == 0) ga0[jt] = ga[jt];
      }
    }
    if (i >= 1) {
      f32x4 gprev[2] = {z4, z4};
      XRD_SB();
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 8; ++s)
          gprev[kt] = XRD_MFMA4(awh[kt][s], ga[s >> 2][s & 3], gprev[kt]);
      XRD_SB();
      gh[0] = gprev[0];
      gh[1] = gprev[1];
    }
  }
  if (NEED_E) {
    // d loss / d sin(p.B) = W0^T ga0 + W3e^T ga3, one 16-feature tile at a
    // time; then through sin: lane group q owns feature k = emap(4kt+r, q)
    float a3[8], a0[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      a3[s] = pk[P::W3ET + s * 64 + lane];
      a0[s] = pk[P::W0T + s * 64 + lane];
    }
#pragma unroll 1
    for (int kt = 0; kt < 6; ++kt) {
      float n3[8], n0[8];
      if (kt < 5) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          n3[s] = pk[P::W3ET + ((kt + 1) * 8 + s) * 64 + lane];
          n0[s] = pk[P::W0T + ((kt + 1) * 8 + s) * 64 + lane];
        }
      }
      XRD_SB();
      f32x4 ge = z4;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        ge = XRD_MFMA4(a3[s], ga3[s >> 2][s & 3], ge);
        ge = XRD_MFMA4(a0[s], ga0[s >> 2][s & 3], ge);
      }
      XRD_SB();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = emap(4 * kt + r, q);
        const f32x4 bk = *reinterpret_cast<const f32x4*>(pk + P::EMB + k * 4);
        const float garg = ge[r] * cos_cw(embed_arg(p, bk));
        if (NEED_DP) {
#pragma unroll
          for (int a = 0; a < 3; ++a) gp[a] += garg * bk[a];
        }
      }
      if (kt < 5) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          a3[s] = n3[s];
          a0[s] = n0[s];
        }
      }
    }
  }
}

// coarse decoder backward: only d/d(c) is needed (grid_coarse is the only
// parameter optimised in the coarse stage; no pose gradient, no decoder grads).
// One layer with its index a template parameter, fragments read ahead of the
// MFMAs like noxyz_fwd_layer.
template <int NT, int I>
__device__ __forceinline__ void noxyz_bwd_layer(
    const float* __restrict__ pk, int lane, const uint64_t (&mask)[NT],
    f32x4 (&gh)[NT][2], f32x4 (&gc)[NT][2]) {
  using P = NoXyzPack;
  // layer 3: tiles 0,1 -> c part, tiles 2,3 -> h part
  constexpr int KTS = (I == 3) ? 4 : 2;
  float a[KTS][8];
#pragma unroll
  for (int kt = 0; kt < KTS; ++kt)
#pragma unroll
    for (int s = 0; s < 8; ++s)
      a[kt][s] = pk[P::wt(I) + (kt * 8 + s) * 64 + lane];
  f32x4 ga[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ga[t][jt][r] =
            ((mask[t] >> (I * 8 + jt * 4 + r)) & 1) ? gh[t][jt][r] : 0.f;
  f32x4 gprev[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    gprev[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    gprev[t][1] = gprev[t][0];
  }
#pragma unroll
  for (int kt = 0; kt < KTS; ++kt)
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (I == 3 && kt < 2)
          gc[t][kt] = XRD_MFMA4(a[kt][s], ga[t][s >> 2][s & 3], gc[t][kt]);
        else
          gprev[t][kt & 1] =
              XRD_MFMA4(a[kt][s], ga[t][s >> 2][s & 3], gprev[t][kt & 1]);
      }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    gh[t][0] = gprev[t][0];
    gh[t][1] = gprev[t][1];
  }
}

template <int NT>
__device__ __forceinline__ void noxyz_bwd(const float* __restrict__ pk,
                                          int lane, const float (&gout)[NT],
                                          const uint64_t (&mask)[NT],
                                          f32x4 (&gc)[NT][2]) {
  using P = NoXyzPack;
  const int q = lane >> 4;
  f32x4 gh[NT][2];
  const f32x4 w0 = *reinterpret_cast<const f32x4*>(pk + P::WOUT + 4 * q);
  const f32x4 w1 = *reinterpret_cast<const f32x4*>(pk + P::WOUT + 16 + 4 * q);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    gh[t][0] = w0 * gout[t];
    gh[t][1] = w1 * gout[t];
    gc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    gc[t][1] = gc[t][0];
  }
  noxyz_bwd_layer<NT, 4>(pk, lane, mask, gh, gc);
  noxyz_bwd_layer<NT, 3>(pk, lane, mask, gh, gc);
  noxyz_bwd_layer<NT, 2>(pk, lane, mask, gh, gc);
  noxyz_bwd_layer<NT, 1>(pk, lane, mask, gh, gc);
  noxyz_bwd_layer<NT, 0>(pk, lane, mask, gh, gc);
  // layer 0 consumes c directly
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    gc[t][0] += gh[t][0];
    gc[t][1] += gh[t][1];
  }
}

// ---------------------------------------------------------------------------
// per-ray sampling (conv_onet.py:391-484)
// ---------------------------------------------------------------------------
struct RayCtx {
  float o[3], d[3];
  float gd;     // sensor depth of the ray (0 = invalid)
  bool has_d;   // depth-guided sampling active
};

// returns this lane's sorted z (lane < S) through LDS arrays zu/zs [64]
template <int S>
__device__ __forceinline__ double sample_z(const xrd_nice_scene& sc,
                                           const RayCtx& rc, float dmax,
                                           int lane, double* zu, double* zs) {
  double far = 1e300;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double t0 = (sc.bound[2 * a] - (double)rc.o[a]) / (double)rc.d[a];
    const double t1 = (sc.bound[2 * a + 1] - (double)rc.o[a]) / (double)rc.d[a];
    far = fmin(far, fmax(t0, t1));
  }
  far += 0.01;
  if (rc.has_d) far = fmin(fmax(far, 0.0), (double)(dmax * 1.2f));
  const int nu = sc.n_samples;
  double z = 1e300;
  if (lane < nu) {
    const float nearf = rc.has_d ? rc.gd * 0.01f : 0.01f;
    const float tv = sc.t_uniform[lane];
    z = (double)(nearf * (1.f - tv)) + far * (double)tv;
  } else if (lane < S) {
    const double ts = sc.t_surface[lane - nu];
    if (rc.gd > 0.f)
      z = (double)(0.95f * rc.gd) * (1.0 - ts) + (double)(1.05f * rc.gd) * ts;
    else
      z = 0.001 * (1.0 - ts) + (double)dmax * ts;
  }
  zu[lane] = z;
  wave_lds_sync();
  int rank = 0;
#pragma unroll 8
  for (int j = 0; j < S; ++j) {
    const double zj = zu[j];
    rank += (zj < z || (zj == z && j < lane)) ? 1 : 0;
  }
  if (lane < S) zs[rank] = z;
  wave_lds_sync();
  return lane < S ? zs[lane] : 0.0;
}

__device__ __forceinline__ bool in_bound(const double (&p)[3],
                                         const double* bd) {
  return p[0] < bd[1] && p[0] > bd[0] && p[1] < bd[3] && p[1] > bd[2] &&
         p[2] < bd[5] && p[2] > bd[4];
}

template <int NT>
__device__ __forceinline__ float pick_tile(const float (&v)[NT], int q) {
  float r = v[0];
#pragma unroll
  for (int t = 1; t < NT; ++t) r = (q == t) ? v[t] : r;
  return r;
}


// ---------------------------------------------------------------------------
// kernels: one wave = one 16-sample tile of one ray; a block holds RPB rays
// (RPB*NT waves).  Waves of a ray meet in LDS for the compositing scan.
// ---------------------------------------------------------------------------
constexpr int RPB = 2;   // rays per block (forward)
constexpr int RPBB = 1;  // rays per block (backward: register heavy)
constexpr int kColorFlat = MlpFlat<32, 4>::LEN;
constexpr int kMaxBwdBlocks = 1 << 20;

struct TileGeom {
  double p64[3];
  float p32[3];
  double z;
  bool inb;
};

__device__ __forceinline__ void tile_geom(const RayCtx& rc, double z,
                                          const double* bd, TileGeom& g) {
  g.z = z;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    g.p64[a] = (double)rc.o[a] + (double)rc.d[a] * z;
    g.p32[a] = (float)g.p64[a];
  }
  g.inb = in_bound(g.p64, bd);
}

__device__ __forceinline__ void load_ray(const float* __restrict__ rays_o,
                                         const float* __restrict__ rays_d,
                                         const float* __restrict__ gt_depth,
                                         int ray, bool use_depth, RayCtx& rc) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    rc.o[a] = rays_o[ray * 3 + a];
    rc.d[a] = rays_d[ray * 3 + a];
  }
  rc.has_d = use_depth;
  rc.gd = use_depth ? gt_depth[ray] : 0.f;
}

constexpr int kCoarseRep = 32;  // replicas of the coarse-grid gradient

// grad += sum of the replicas; the replicas are left zeroed for the next call
__global__ __launch_bounds__(256) void coarse_rep_reduce_kernel(
    float* __restrict__ rep, int64_t n, float* __restrict__ grad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
#pragma unroll 8
  for (int r = 0; r < kCoarseRep; ++r) {
    const float v = rep[(size_t)r * n + i];
    if (v != 0.f) {
      s += v;
      rep[(size_t)r * n + i] = 0.f;
    }
  }
  if (s != 0.f) grad[i] += s;
}

// Compositing backward (utils.py:189-244) of one ray from the saved raw: lane
// l is sample l.  Returns d loss / d occupancy logit of the lane's sample, its
// weight, and the ray's colour gradient.
template <int S>
__device__ __forceinline__ void composite_bwd(
    const float* __restrict__ raw, int ray, int lane, double zl,
    const double* __restrict__ g_depth, const double* __restrict__ g_var,
    const float* __restrict__ g_rgb, float& gocc_s, float& w,
    float (&grgb)[3]) {
  const bool valid = lane < S;
  f32x4 rw = {0.f, 0.f, 0.f, 0.f};
  if (valid)
    rw = *reinterpret_cast<const f32x4*>(raw + ((size_t)ray * S + lane) * 4);
  // alpha = sigmoid(10 occ) AND 1 - alpha = sigmoid(-10 occ), each to full
  // relative precision: at a sharp surface alpha -> 1 and the reference's
  // f32 "1 - alpha" (transmittance factor, sigmoid derivative) keeps only
  // eps / (1 - alpha) relative accuracy — two f32 evaluations then disagree
  // at 1e-4 on exactly the rays the tracking loss weights most
  float alpha = 0.f, oma = 1.f;
  if (valid) {
    const float e = expf(-10.f * fabsf(rw[3]));  // <= 1
    const float hi = 1.f / (1.f + e), lo = e / (1.f + e);
    alpha = rw[3] >= 0.f ? hi : lo;
    oma = rw[3] >= 0.f ? lo : hi;
  }
  const double f = (double)oma + 1e-10;
  // transmittance, weights and the sums below in f64: the weight gradient
  // subtracts nearly equal sums of gw*w, which turns the ~1e-6 rounding of an
  // f32 product scan into 1e-4 of the result
  double incl = valid ? f : 1.0;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double u = __shfl_up(incl, o);
    if (lane >= o) incl *= u;
  }
  double T = __shfl_up(incl, 1);
  if (lane == 0) T = 1.0;
  const double wd = (double)alpha * T;
  w = (float)wd;
  const double dep = wave_sum(valid ? wd * zl : 0.0);
  const double tmp = zl - dep;
  const double gd_in = g_depth ? g_depth[ray] : 0.0;
  const double gv_in = g_var ? g_var[ray] : 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) grgb[a] = g_rgb ? g_rgb[ray * 3 + a] : 0.f;
  const double sw_tmp = wave_sum(valid ? wd * tmp : 0.0);
  const double gdep = gd_in - 2.0 * gv_in * sw_tmp;
  // d loss / d weight, and the suffix sums of the product-scan backward, in
  // f64: galpha = gw*T - (sum_{j>i} gw_j w_j)/f subtracts two numbers that
  // agree to ~2-3 digits when the depth term dominates (the tracking loss
  // scales it by 1/sqrt(var)), so f32 sums lose the 1e-4 bar there — torch's
  // own f32 evaluation does (tests/test_nice_hip.py compares both with an f64
  // evaluation of the same formulas)
  double gw = 0.0;
  if (valid)
    gw = gdep * zl + gv_in * tmp * tmp +
         (double)(grgb[0] * rw[0] + grgb[1] * rw[1] + grgb[2] * rw[2]);
  double suf = valid ? gw * wd : 0.0;  // inclusive suffix sum
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double u = __shfl_down(suf, o);
    if (lane + o < 64) suf += u;
  }
  double sexc = __shfl_down(suf, 1);
  if (lane == 63) sexc = 0.0;
  const float galpha =
      valid ? (float)(gw * T - sexc / f) : 0.f;
  gocc_s = galpha * 10.f * alpha * oma;
}

// Coarse stage backward (grid_coarse is its only parameter; no pose gradient,
// conv_onet.py:187-195): one wave = one tile, a block = 4 rays = 8 tile waves
// with the decoder's fragments (forward + transposed, 50 KB) staged in LDS once
// per block, like nice_map_coarse_kernel (rounds 1-4: one ray a block, the
// fragments read from L2 behind every MFMA).
constexpr int kBwdCoarseRPB = 4;
constexpr int kBwdCoarseWaves = 2 * kBwdCoarseRPB;
constexpr size_t kBwdCoarseLds =
    ((size_t)NoXyzPack::LEN + kBwdCoarseWaves * (256 + kScatterFloats)) *
    sizeof(float);
__global__ __launch_bounds__(kBwdCoarseWaves * 64, 2) void
nice_bwd_coarse_kernel(
    xrd_nice_scene sc, int n, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ raw,
    const double* __restrict__ g_depth, const double* __restrict__ g_var,
    const float* __restrict__ g_rgb, float* gg_coarse,
    float* __restrict__ ws) {
  constexpr int S = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);  // staged fragments
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = wave >> 1, tile = wave & 1;
  const int q = lane >> 4, li = lane & 15;
  float* R = wl + NoXyzPack::LEN + wave * (256 + kScatterFloats);
  double* zbuf = reinterpret_cast<double*>(R);
  ScatterLds SL;
  SL.gt = R + 256;
  SL.off = reinterpret_cast<int*>(SL.gt + 16 * 33);
  SL.w = SL.gt + 16 * 33 + 16 * 8;
  for (int i = threadIdx.x * 4; i < NoXyzPack::LEN; i += blockDim.x * 4)
    *reinterpret_cast<f32x4*>(wl + i) =
        *reinterpret_cast<const f32x4*>(sc.dec[0] + i);
  __syncthreads();
  const int ngroups = (n + kBwdCoarseRPB - 1) / kBwdCoarseRPB;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int ray = __builtin_amdgcn_readfirstlane(grp * kBwdCoarseRPB + slot);
    if (ray >= n) continue;   // (no block barrier inside the loop)
    RayCtx rc;
    load_ray(rays_o, rays_d, nullptr, ray, false, rc);
    const double zl = sample_z<S>(sc, rc, 0.f, lane, zbuf, zbuf + 64);
    float gocc_s, w, grgb[3];
    composite_bwd<S>(raw, ray, lane, zl, g_depth, g_var, g_rgb, gocc_s, w,
                     grgb);
    const int src = 16 * tile + li;
    TileGeom tg;
    tile_geom(rc, zbuf[64 + src], sc.bound, tg);
    float gocc = __shfl(gocc_s, src);
    if (!tg.inb) gocc = 0.f;  // occupancy was overridden to 100
    f32x4 c_a[1][2], gc[1][2];
    float o1[1];
    uint64_t mask[1];
    const float go[1] = {gocc};
    Tri tr;
    tri_prepare(tg.p64, sc.bound, sc.coarse_enlarge, sc.gdim + 0, tr);
    tri_gather(sc.grid[0], tr, q, c_a[0]);
    noxyz_fwd<1, true>(wl, lane, c_a, o1, mask);
    noxyz_bwd<1>(wl, lane, go, mask, gc);
    tri_prepare(tg.p64, sc.bound, sc.coarse_enlarge, sc.gdim + 0, tr);
    // coarse grid: ~1.3e3 cells and every ray starts in the camera's cell, so
    // the atomics of 1000 rays serialise on a few lines; rays spread over
    // kCoarseRep private replicas (ws), summed afterwards
    float* ggc = gg_coarse;
    if (ws != nullptr && gg_coarse != nullptr)
      ggc = ws + (size_t)(ray & (kCoarseRep - 1)) *
                     ((size_t)sc.gdim[0] * sc.gdim[1] * sc.gdim[2] * 32);
    grid_scatter(ggc, sc.gmask[0], tr, lane, gc[0], SL);
  }
}

// ---------------------------------------------------------------------------
// Fused backward of the middle / fine / colour stages: ONE launch
// back-propagates every decoder of the stage (round 1: one launch per decoder,
// each re-running the ray set-up, + a staging pass through HBM and a separate
// dW kernel for the colour decoder's weight gradients).
//
// A block = FWV consecutive 16-sample tiles (tiles of a ray are consecutive, a
// ray may straddle blocks); a wave owns one tile through all decoder phases:
//   compositing backward of its ray -> middle -> fine -> colour, each phase
//   = gather, forward recompute (ReLU masks), backward on transposed weight
//   fragments, grid-gradient scatter.  Ray-gradient partial sums go to a
//   workspace row per tile (f64) and are added up by nice_bwd_finish_kernel.
//
// WEIGHTS IN LDS.  Measured on MI355X (profiles/r02_weight_fetch.txt): with
// the fragments read from L2 the backward is bound by their fetch — the same
// kernels with every fragment load folded onto 4 KB (L1 hits) run 1.7-2.4x
// faster.  The packed parameters of a decoder do not fit a CU's L1 (32 KB),
// but one PASS of one decoder fits LDS (forward <= 85 KB, backward <= 84 KB,
// nice_layout.h keeps each one contiguous): the block stages the pass's
// fragments once per group of tiles (block barrier, cooperative b128 copy,
// barrier) and every wave reads them with ds_read — L2 traffic per tile drops
// by the number of tiles per block, fragment latency to LDS latency, and the
// registers the compiler spent on prefetching global fragments are free.
//
// Colour-decoder weight gradients (NEED_DW): dW = sum over points of
// (gradient) x (layer input) as v_mfma_f32_16x16x4_f32 with the POINTS on the
// K dimension.  The 68 16x16 blocks of the flat gradient are split over the
// eight waves of a block; every wave keeps ITS blocks in MFMA accumulators
// across all tiles the block ever processes (persistent blocks) and adds them
// to one of kDwRep replicas once, at the end.  Operands are exchanged through
// LDS (DwLds): layer by layer each wave publishes its tile's gh_i, ReLU mask
// and h_{i-1} (kept in registers since the forward pass), the block
// synchronises, and every wave contracts its block over the eight tiles.
// Nothing is staged through HBM (round 1: 90 MB per launch).
// ---------------------------------------------------------------------------
// tiles (= waves) per block.  Measured at 1000 rays (colour stage, grid
// gradients only): 4 waves 218 us, 8 waves 185 us, 16 waves 151 us — more
// tiles share one staging of the fragments and more waves hide the gathers;
// 16 waves leave 128 registers a lane, which the variants without pose
// gradients fit and the others do not (they spill and lose: 263 vs 242 us).
// Small batches (tracking: 200 rays = 600 tiles) fill more CUs with narrow
// blocks — a block's staging costs only a few microseconds — so the width is
// a launch-time choice: 4, 8 or 16 waves (fused_width()).
constexpr int FW = 8;
constexpr int FW_WIDE = 16;
constexpr int FWD = 8;           // ... with weight gradients (2 per SIMD too)
constexpr int kDwRep = 8;        // replicas the blocks add their dW into
constexpr int kRS = 36;          // row stride of a point-major LDS matrix
constexpr int kMat = 16 * kRS;   // 576 floats
// LDS map (floats).  [0, kWMax) the staged fragments of the current pass
// (largest: fine decoder forward 21316); the waves' dW exchange regions start
// behind the largest COLOUR pass (16196) and overlap the tail of the fragment
// region, which only the fine decoder uses; the scratch of a wave (z values,
// scatter tiles) aliases its exchange region, idle outside the colour
// backward.
constexpr int kWMax = 21696;
constexpr int kDwBase = 16256;
constexpr int kScratch = 256 + kScatterFloats;  // per wave
// Exchange region of one wave = one tile: point-major matrices [16 points][32
// features] with a row stride of 36 floats — the producer writes its
// accumulator-layout registers as b128, a consumer lane (m = l&15, q = l>>4)
// reads feature 16jt+m of point 4q+s for K-step s; with stride 36 the four
// lane groups q fall into four disjoint 16-bank ranges.
struct DwLds {
  static constexpr int TC = 0;             // grid features c (later garg 0..31)
  static constexpr int TH = kMat;          // h_{i-1} (later garg 32..63)
  static constexpr int TG = 2 * kMat;      // gh_i; masked ga_0 (garg 64..95)
  static constexpr int TX = 3 * kMat;      // h_4; masked ga_3
  static constexpr int TM = 4 * kMat;      // ReLU masks [5][16] (bit f)
  static constexpr int TP = TM + 80;       // sample positions [16][4]
  static constexpr int TGO = TP + 64;      // d loss / d decoder output [16][4]
  static constexpr int LEN = TGO + 64;     // 2512 floats
};
static_assert(kScratch <= DwLds::LEN, "scratch aliases the exchange region");
static_assert(MlpPack<64, 1>::WHT <= kWMax, "fine forward fits");
static_assert(MlpPack<64, 1>::LEN - MlpPack<64, 1>::EMB <= kWMax, "fine bwd");
static_assert(MlpPack<32, 4>::WHT <= kDwBase, "colour forward below dW region");
static_assert(MlpPack<32, 4>::LEN - MlpPack<32, 4>::EMB <= kDwBase, "");

constexpr size_t fused_lds_floats(bool dw) {
  return dw ? ((size_t)kDwBase + FWD * DwLds::LEN > (size_t)kWMax
                   ? (size_t)kDwBase + FWD * DwLds::LEN
                   : (size_t)kWMax)
            : (size_t)kWMax + FW_WIDE * kScratch;  // widest variant
}
static_assert(fused_lds_floats(true) * 4 <= 163840, "LDS per CU");
static_assert(fused_lds_floats(false) * 4 <= 163840, "LDS per CU");

// stage n floats (multiple of 4, 16-byte aligned) of packed parameters
__device__ __forceinline__ void stage_weights(float* __restrict__ wl,
                                              const float* __restrict__ src,
                                              int n) {
  __syncthreads();  // everybody is done with the previous pass's fragments
  for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4)
    *reinterpret_cast<f32x4*>(wl + i) =
        *reinterpret_cast<const f32x4*>(src + i);
  __syncthreads();
}

// stage_weights in two halves for a block's FIRST pass: the loads are issued
// into registers (stage_issue), the caller runs its ray set-up and grid
// gathers under their latency, then the LDS stores and the barrier
// (stage_commit).  T = threads of the block, NV = ceil(n / (4 T)).
template <int T, int NV>
__device__ __forceinline__ void stage_issue(const float* __restrict__ src,
                                            int n, f32x4 (&r)[NV]) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = ((int)threadIdx.x + k * T) * 4;
    if (i < n) r[k] = *reinterpret_cast<const f32x4*>(src + i);
  }
}
template <int T, int NV>
__device__ __forceinline__ void stage_commit(float* __restrict__ wl, int n,
                                             const f32x4 (&r)[NV]) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = ((int)threadIdx.x + k * T) * 4;
    if (i < n) *reinterpret_cast<f32x4*>(wl + i) = r[k];
  }
  __syncthreads();
}

// The 68 16x16 blocks of the colour decoder's flat gradient over the FWD = 8
// waves of a block (w = wave, jt = (w >> 1) & 1, kt = w & 1):
//   layer i:  w < 4  fc_c.i.weight block (jt, kt)        [A = gh_i, B = c]
//             w >= 4 pts_linears.i hidden block (jt, kt)  [A = ga_i, B = h_{i-1}]
//             (i = 0 has no hidden block); bias rows jt by the kt == 0 waves
//   layer 4 also: output_linear.weight, columns 16(w&1).. by waves 6, 7
//   Fourier parts: w < 4 of pts_linears.0, w >= 4 of pts_linears.3:
//             rows jt, column tiles 3kt..3kt+2
//   embedder._B: column tile w by waves 0..5
struct DwAcc {
  f32x4 lay[5];   // the wave's block of layer i
  f32x4 emb[3];   // its three Fourier-part blocks
  f32x4 x;        // output_linear (w = 6,7) / embedder._B (w < 6) block
  float bias[5];  // kt == 0 waves: fc_c.i.bias (w < 4) / pts_linears.i.bias rows jt
  float bout;     // wave 6: output_linear.bias
};

__device__ __forceinline__ void dw_acc_zero(DwAcc& A) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    A.lay[i] = z;
    A.bias[i] = 0.f;
    if (i < 3) A.emb[i] = z;
  }
  A.x = z;
  A.bout = 0.f;
}

// layer I of the exchange: gh_I in TG, mask_I in TM, c in TC, h_{I-1} in TH
// (and h_4 in TX for the output layer) of every active tile
template <int I>
__device__ __forceinline__ void dw_layer_step(const float* __restrict__ lds,
                                              int nact, int wave, int lane,
                                              DwAcc& A) {
  const int m = lane & 15, q = lane >> 4;
  const int jt = (wave >> 1) & 1, kt = wave & 1;
  const bool hid = wave >= 4;
  if (!(I == 0 && hid && kt != 0)) {
    const int bslot = hid ? DwLds::TH : DwLds::TC;
    for (int t = 0; t < nact; ++t) {
      const float* R = lds + t * DwLds::LEN;
      const uint32_t* M =
          reinterpret_cast<const uint32_t*>(R + DwLds::TM) + I * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int pt = 4 * q + s;
        float a = R[DwLds::TG + pt * kRS + 16 * jt + m];
        if (hid && !((M[pt] >> (16 * jt + m)) & 1u)) a = 0.f;  // ga = masked gh
        if (I >= 1 || !hid) {
          const float b = R[bslot + pt * kRS + 16 * kt + m];
          A.lay[I] = XRD_MFMA4(a, b, A.lay[I]);
        }
        if (kt == 0) A.bias[I] += a;
      }
    }
  }
  if (I == 4 && wave >= 6) {  // output layer: rows = output o, cols = h_4
    for (int t = 0; t < nact; ++t) {
      const float* R = lds + t * DwLds::LEN;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int pt = 4 * q + s;
        const float ao = (m < 4) ? R[DwLds::TGO + pt * 4 + (m & 3)] : 0.f;
        const float bh = R[DwLds::TX + pt * kRS + 16 * kt + m];
        A.x = XRD_MFMA4(ao, bh, A.x);
        if (wave == 6) A.bout += ao;
      }
    }
  }
}

// Fourier-feature weights of layers 0 and 3 against the recomputed sin(p.B):
// masked ga_0 is in TG, masked ga_3 in TX
__device__ __forceinline__ void dw_emb_step(const float* __restrict__ w,
                                            const float* __restrict__ lds,
                                            int nact, int wave, int lane,
                                            DwAcc& A) {
  using P = MlpPack<32, 4>;
  const int m = lane & 15, q = lane >> 4;
  const int jt = (wave >> 1) & 1, kb = 3 * (wave & 1);
  const int aslot = (wave < 4 ? DwLds::TG : DwLds::TX) + 16 * jt + m;
  for (int t = 0; t < nact; ++t) {
    const float* R = lds + t * DwLds::LEN;
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
      const int pt = 4 * q + s;
      const float a = R[aslot + pt * kRS];
      const f32x4 pp = *reinterpret_cast<const f32x4*>(R + DwLds::TP + pt * 4);
      const float pv[3] = {pp[0], pp[1], pp[2]};
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) {
        const f32x4 bk = *reinterpret_cast<const f32x4*>(
            w + P::EMB + (16 * (kb + k3) + m) * 4);
        A.emb[k3] = XRD_MFMA4(a, sin_cw(embed_arg(pv, bk)), A.emb[k3]);
      }
    }
  }
}

// embedder._B: rows = axis a (lane m < 3), cols = Fourier feature; the
// per-point d loss / d (p.B) sits in TC / TH / TG (feature f -> matrix f>>5,
// column f&31)
__device__ __forceinline__ void dw_embB_step(const float* __restrict__ lds,
                                             int nact, int wave, int lane,
                                             DwAcc& A) {
  const int m = lane & 15, q = lane >> 4;
  if (wave >= 6) return;
  const int f = 16 * wave + m;
  for (int t = 0; t < nact; ++t) {
    const float* R = lds + t * DwLds::LEN;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int pt = 4 * q + s;
      const float ap = (m < 3) ? R[DwLds::TP + pt * 4 + (m & 3)] : 0.f;
      const float gb = R[(f >> 5) * kMat + pt * kRS + (f & 31)];
      A.x = XRD_MFMA4(ap, gb, A.x);
    }
  }
}

// add the wave's dW blocks to one replica of the flat gradient
__device__ __forceinline__ void dw_flush(float* __restrict__ rep, int wave,
                                         int lane, DwAcc& A) {
  using F = MlpFlat<32, 4>;
  const int n = lane & 15, q = lane >> 4;
  const int jt = (wave >> 1) & 1, kt = wave & 1;
  const bool hid = wave >= 4;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    if (!hid || i >= 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jt + 4 * q + r;
        float* dst = hid ? rep + F::pw(i) + j * F::pstride(i) + F::pcol(i)
                         : rep + F::fcw(i) + j * 32;
        atomicAdd(dst + 16 * kt + n, A.lay[i][r]);
      }
    }
    // bias rows: lane (m = n, q) holds the sum over points 4q..4q+3
    const float b = group4_sum(A.bias[i]);
    if (kt == 0 && q == 0)
      atomicAdd(rep + (hid ? F::pb(i) : F::fcb(i)) + 16 * jt + n, b);
  }
#pragma unroll
  for (int k3 = 0; k3 < 3; ++k3) {
    const int k = 16 * (3 * kt + k3) + n;
    if (k < kEmbK) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jt + 4 * q + r;
        float* dst = hid ? rep + F::P3W + j * (kEmbK + 32)
                         : rep + F::P0W + j * kEmbK;
        atomicAdd(dst + k, A.emb[k3][r]);
      }
    }
  }
  if (wave >= 6) {
    if (q == 0) {  // rows 0..3 of the accumulator = lane group 0
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(rep + F::OW + r * 32 + 16 * kt + n, A.x[r]);
    }
    if (wave == 6) {
      const float b = group4_sum(A.bout);
      if (q == 0 && n < 4) atomicAdd(rep + F::OB + n, b);
    }
  } else if (q == 0) {
    const int k = 16 * wave + n;
    if (k < kEmbK) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
        atomicAdd(rep + F::EB + r * kEmbK + k, A.x[r]);
    }
  }
}

// write a D-layout register pair (features 16jt+4q+r of point li) to a matrix
__device__ __forceinline__ void lds_put(float* M, int lane,
                                        const f32x4 (&v)[2]) {
  const int q = lane >> 4, li = lane & 15;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
    *reinterpret_cast<f32x4*>(M + li * kRS + 16 * jt + 4 * q) = v[jt];
}

// ReLU mask of layer i -> one word per point (bit f = feature f active)
__device__ __forceinline__ void lds_put_mask(float* R, int i, int lane,
                                             uint64_t mask) {
  const int q = lane >> 4, li = lane & 15;
  uint32_t word = 0;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if ((mask >> (i * 8 + jt * 4 + r)) & 1)
        word |= 1u << (16 * jt + 4 * q + r);
  word |= (uint32_t)__shfl_xor((int)word, 16);
  word |= (uint32_t)__shfl_xor((int)word, 32);
  if (q == 0) reinterpret_cast<uint32_t*>(R + DwLds::TM)[i * 16 + li] = word;
}

// Colour decoder backward with the weight-gradient exchange.  EVERY wave of
// the block runs this (block barriers); `active` waves also back-propagate
// their own tile: gc = d loss / d grid features, gp += d loss / d position.
// w: the staged backward fragments (MlpPack offsets), lds: exchange regions of
// the block's waves, hs: the tile's layer outputs h_0..h_4 (registers).
template <bool NEED_DP>
__device__ __forceinline__ void color_bwd_dw(
    const float* __restrict__ w, float* __restrict__ lds, int wave, int lane,
    bool active, int nact, const float (&p)[1][3], const f32x4 (&c)[1][2],
    const float (&go)[1][4], uint64_t mask, const f32x4 (&hs)[5][2],
    f32x4 (&gc)[1][2], float (&gp)[1][3], DwAcc& A) {
  using P = MlpPack<32, 4>;
  const int q = lane >> 4, li = lane & 15;
  float* R = lds + wave * DwLds::LEN;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 gh[2] = {z4, z4}, ga[2] = {z4, z4}, ga3[2] = {z4, z4};
  gc[0][0] = z4;
  gc[0][1] = z4;
  __syncthreads();  // the scratch aliases of every wave are idle
  if (active) {
    lds_put(R + DwLds::TC, lane, c[0]);
    lds_put(R + DwLds::TX, lane, hs[4]);
    if (q == 0) {
      *reinterpret_cast<f32x4*>(R + DwLds::TP + li * 4) =
          f32x4{p[0][0], p[0][1], p[0][2], 0.f};
      *reinterpret_cast<f32x4*>(R + DwLds::TGO + li * 4) =
          f32x4{go[0][0], go[0][1], go[0][2], go[0][3]};
    }
    // gh_4 = Wout^T go
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const f32x4 w0 =
          *reinterpret_cast<const f32x4*>(w + P::WOUT + o * 32 + 4 * q);
      const f32x4 w1 =
          *reinterpret_cast<const f32x4*>(w + P::WOUT + o * 32 + 16 + 4 * q);
      gh[0] += w0 * go[0][o];
      gh[1] += w1 * go[0][o];
    }
  }
  // one layer: publish (gh_i, mask_i, h_{i-1}), contract the block's share,
  // back-propagate the own tile
  auto layer = [&](auto IC) {
    constexpr int i = decltype(IC)::value;
    if (active) {
      lds_put(R + DwLds::TG, lane, gh);
      lds_put_mask(R, i, lane, mask);
      if (i >= 1) lds_put(R + DwLds::TH, lane, hs[i >= 1 ? i - 1 : 0]);
    }
    __syncthreads();
    dw_layer_step<i>(lds, nact, wave, lane, A);
    if (active) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ga[jt][r] = ((mask >> (i * 8 + jt * 4 + r)) & 1) ? gh[jt][r] : 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const float a = w[P::wct(i) + (kt * 8 + s) * 64 + lane];
          gc[0][kt] = XRD_MFMA4(a, gh[s >> 2][s & 3], gc[0][kt]);
        }
      if (i >= 1) {
        f32x4 gprev[2] = {z4, z4};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const float a =
                w[P::wht(i >= 1 ? i : 1) + (kt * 8 + s) * 64 + lane];
            gprev[kt] = XRD_MFMA4(a, ga[s >> 2][s & 3], gprev[kt]);
          }
        gh[0] = gprev[0];
        gh[1] = gprev[1];
      }
    }
    __syncthreads();  // TG / TH may be rewritten
  };
  layer(std::integral_constant<int, 4>{});
  layer(std::integral_constant<int, 3>{});
  ga3[0] = ga[0];
  ga3[1] = ga[1];
  layer(std::integral_constant<int, 2>{});
  layer(std::integral_constant<int, 1>{});
  layer(std::integral_constant<int, 0>{});
  // ga now holds the masked ga_0
  if (active) {
    lds_put(R + DwLds::TG, lane, ga);
    lds_put(R + DwLds::TX, lane, ga3);
  }
  __syncthreads();
  dw_emb_step(w, lds, nact, wave, lane, A);
  __syncthreads();  // TC / TH / TG may be rewritten
  if (active) {
    // d loss / d sin(p.B) = W0^T ga0 + W3e^T ga3, through the sine; lane
    // group q owns feature k = emap(4kt+r, q)
#pragma unroll 1
    for (int kt = 0; kt < 6; ++kt) {
      f32x4 ge = z4;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float a3 = w[P::W3ET + (kt * 8 + s) * 64 + lane];
        const float a0 = w[P::W0T + (kt * 8 + s) * 64 + lane];
        ge = XRD_MFMA4(a3, ga3[s >> 2][s & 3], ge);
        ge = XRD_MFMA4(a0, ga[s >> 2][s & 3], ge);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = emap(4 * kt + r, q);
        const f32x4 bk = *reinterpret_cast<const f32x4*>(w + P::EMB + k * 4);
        const float garg = ge[r] * cos_cw(embed_arg(p[0], bk));
        if (NEED_DP) {
#pragma unroll
          for (int a = 0; a < 3; ++a) gp[0][a] += garg * bk[a];
        }
        R[(k >> 5) * kMat + li * kRS + (k & 31)] = garg;
      }
    }
  }
  __syncthreads();
  dw_embB_step(lds, nact, wave, lane, A);
  __syncthreads();  // the regions (and their scratch aliases) may be reused
}

}  // namespace
}  // namespace xrd
