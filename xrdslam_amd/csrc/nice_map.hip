// NICE-SLAM mapping iteration as ONE launch per stage (gfx950): forward render,
// the mapping loss and the backward of everything it reaches.
//
// Why this is possible: the mapping losses are plain sums over rays
// (slam/models/conv_onet.py:178-184: L1 depth on the rays with a valid sensor
// depth, + 0.2 * L1 colour in the colour stage), so d loss / d (depth, rgb) of
// a ray is known as soon as that ray is composited — no batch statistic like
// the tracking loss's median sits between the forward and the backward.
// Round 2 ran a forward launch, a loss launch, two torch elementwise launches
// (the autograd scaling of the loss gradients) and a backward launch that
// recomputed the whole forward; here a block owns WHOLE rays, runs every
// decoder forward ONCE (keeping the ReLU masks, and the colour decoder's layer
// outputs when its weights train), composites its rays in LDS, and walks the
// decoders backwards: colour -> fine -> middle.
//
// Block geometry: one 16-sample tile per wave, 3 tiles per ray, 4 rays = 12
// waves = 3 per SIMD a block; 1000 rays = 250 blocks = ONE round of the 256
// CUs (round 2's 8-wave backward blocks took two: 375 groups on 256 CUs).
//
// Colour-decoder weight gradients (dW = sum over points of gradient x layer
// input, the POINTS on the MFMA K dimension).  Round 2 contracted them inside
// the layer loop through an LDS exchange (two block barriers per layer, the
// layer outputs h_0..h_4 and 36+ accumulators alive in every tile wave: 256
// registers = 2 waves per SIMD, +170 us).  Here the tile waves stay as light
// as the variant without weight gradients: while a tile goes forward /
// backward it drops the contraction's operands (c, h_0..h_4, dL/dh_0..4,
// ReLU masks, positions: 23 KB a tile) into a per-tile scratch as
// feature-major matrices [32 features][16 points], and after the last decoder
// of the group the block's 12 waves contract the group's 12 tiles — 62 16x16
// output blocks, 4-8 accumulators a wave — and write the block's row of the
// partial gradient.  Round 6: the operands come back through an LDS ring
// filled by LDS-DMA loads (each fetched once; rounds 3-5 read them from the
// scratch per unit, 2-10 times each, behind six dependent round trips a
// unit), see dw_contract.  embedder._B (3 x 93) is reduced on the VALU (row
// reductions + LDS adds) instead of 6 padded MFMA blocks.
// Reference maths restated (never copied): conv_onet.py:339-524 (sampling,
// eval_points, the stage's decoders), decoder_nice.py:103-234,
// utils.py:189-244 (raw2outputs_nerf_color), conv_onet.py:145-185 (losses).
#include <hip/hip_runtime.h>

#include "common.h"
#include "nice_device.h"
#include "nice_layout.h"

namespace xrd {
namespace {

// Compositing of one ray (lane l = sample l) from its raw values in LDS, the
// ray's mapping loss and the compositing backward.  Returns d loss / d
// occupancy logit of the lane's sample, its weight and the ray's colour
// gradient.  Same arithmetic as composite_bwd (f64 transmittance / suffix sums)
// with g_var = 0 and g_depth, g_rgb = the signs the L1 terms produce.
template <int S>
__device__ __forceinline__ void map_composite(
    const float* __restrict__ rawray, int lane, double zl, float gt_d,
    const float* __restrict__ tgt, bool kept, bool use_color, float w_color,
    float& gocc_s, float& w, float (&grgb)[3], double& loss) {
  const bool valid = lane < S;
  f32x4 rw = {0.f, 0.f, 0.f, 0.f};
  if (valid) rw = *reinterpret_cast<const f32x4*>(rawray + lane * 4);
  float alpha = 0.f, oma = 1.f;
  if (valid) {
    const float e = expf(-10.f * fabsf(rw[3]));  // <= 1
    const float hi = 1.f / (1.f + e), lo = e / (1.f + e);
    alpha = rw[3] >= 0.f ? hi : lo;
    oma = rw[3] >= 0.f ? lo : hi;
  }
  const double f = (double)oma + 1e-10;
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
  const double diff = (double)gt_d - dep;
  const bool m = kept && gt_d > 0.f;
  const double gdep = m ? (diff > 0 ? -1.0 : (diff < 0 ? 1.0 : 0.0)) : 0.0;
  loss = m ? fabs(diff) : 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) grgb[a] = 0.f;
  if (use_color) {
    float lc = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float c = wave_sum(valid ? w * rw[a] : 0.f);
      if (kept) {
        const float dc = tgt[a] - c;
        lc += fabsf(dc);
        grgb[a] = w_color * (dc > 0.f ? -1.f : (dc < 0.f ? 1.f : 0.f));
      }
    }
    loss += (double)w_color * (double)lc;
  }
  double gw = 0.0;
  if (valid)
    gw = gdep * zl +
         (double)(grgb[0] * rw[0] + grgb[1] * rw[1] + grgb[2] * rw[2]);
  double suf = valid ? gw * wd : 0.0;  // inclusive suffix sum
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double u = __shfl_down(suf, o);
    if (lane + o < 64) suf += u;
  }
  double sexc = __shfl_down(suf, 1);
  if (lane == 63) sexc = 0.0;
  const float galpha = valid ? (float)(gw * T - sexc / f) : 0.f;
  gocc_s = galpha * 10.f * alpha * oma;
}

// ---- tracking iteration in the same launch (round 4) ------------------------
// The tracking loss (slam/models/conv_onet.py:145-176) puts a BATCH statistic
// between the forward and the backward: the residual |d - depth| / sqrt(var)
// of a ray counts only while it is below 10 x the batch's (lower) median.
// Round 3 therefore ran forward launch, loss launch (one block: sort), and a
// backward launch that recomputed the whole forward (43 + 14 + 79 us for 200
// rays).  Here a block still owns whole rays and keeps the forward's state in
// registers; the blocks meet at ONE grid barrier: every ray publishes its
// residual, every block selects the median from the n residuals itself, and
// the backward continues from the registers.  All blocks must be resident
// (n <= 4 x 256 rays: one block a CU) — the caller checks.
struct TrackArgs {
  double* res;             // [n] residuals (1e300: ray not kept)
  double* ray_lossc;       // [n] sum |d colour| of the rays that count
  unsigned* bar;           // grid barrier counter (zero on entry, re-zeroed
                           // by the finishing launch)
  int use_color, handle_dynamic;
  // mapping, optional: the sample points [n*S,3] and d loss / d occupancy
  // logit [n*S] of every sample (xrd_nice_map_iter_export)
  float* exp_p;
  float* exp_g;
};

// forward half of the compositing (lane l = sample l): weights, depth, the
// depth variance and the colour of the ray
template <int S>
__device__ __forceinline__ void track_forward(
    const float* __restrict__ rawray, int lane, double zl, f32x4& rw,
    float& alpha, float& oma, double& T, double& wd, double& dep,
    double& var, float (&c)[3]) {
  const bool valid = lane < S;
  rw = f32x4{0.f, 0.f, 0.f, 0.f};
  if (valid) rw = *reinterpret_cast<const f32x4*>(rawray + lane * 4);
  alpha = 0.f;
  oma = 1.f;
  if (valid) {
    const float e = expf(-10.f * fabsf(rw[3]));  // <= 1
    const float hi = 1.f / (1.f + e), lo = e / (1.f + e);
    alpha = rw[3] >= 0.f ? hi : lo;
    oma = rw[3] >= 0.f ? lo : hi;
  }
  const double f = (double)oma + 1e-10;
  double incl = valid ? f : 1.0;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double u = __shfl_up(incl, o);
    if (lane >= o) incl *= u;
  }
  T = __shfl_up(incl, 1);
  if (lane == 0) T = 1.0;
  wd = (double)alpha * T;
  // the rendered values the loss sees: the arithmetic of the forward launch
  // this replaces (nice_fwd_kernel: f32 sigmoid, f32 product scan like the
  // reference's cumprod, f64 depth / variance sums) — the f64 weights above
  // serve the backward (composite_bwd)
  {
    const float a32 = valid ? 1.f / (1.f + expf(-10.f * rw[3])) : 0.f;
    const float f32f = valid ? (1.f - a32 + 1e-10f) : 1.f;
    float inc = f32f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float u = __shfl_up(inc, o);
      if (lane >= o) inc *= u;
    }
    float T32 = __shfl_up(inc, 1);
    if (lane == 0) T32 = 1.f;
    const float w = a32 * T32;
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = wave_sum(w * rw[a]);
    dep = wave_sum(valid ? (double)w * zl : 0.0);
    const double dz = zl - dep;
    var = wave_sum(valid ? (double)w * dz * dz : 0.0);
  }
}
// backward half: d loss / d occupancy logit of the lane's sample from
// d loss / d depth (the variance is detached, conv_onet.py:157) and d colour
template <int S>
__device__ __forceinline__ float track_backward(
    int lane, double zl, const f32x4 rw, float alpha, float oma, double T,
    double wd, double gdep, const float (&grgb)[3]) {
  const bool valid = lane < S;
  const double f = (double)oma + 1e-10;
  double gw = 0.0;
  if (valid)
    gw = gdep * zl +
         (double)(grgb[0] * rw[0] + grgb[1] * rw[1] + grgb[2] * rw[2]);
  double suf = valid ? gw * wd : 0.0;  // inclusive suffix sum
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double u = __shfl_down(suf, o);
    if (lane + o < 64) suf += u;
  }
  double sexc = __shfl_down(suf, 1);
  if (lane == 63) sexc = 0.0;
  const float galpha = valid ? (float)(gw * T - sexc / f) : 0.f;
  return galpha * 10.f * alpha * oma;
}
// all blocks of the launch (every one resident) meet once
__device__ __forceinline__ void grid_barrier_once(unsigned* bar) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) <
           gridDim.x)
      __builtin_amdgcn_s_sleep(2);
    __threadfence();
  }
  __syncthreads();
}
// 10 x the lower median of the kept rays' residuals (torch.median), selected by
// rank counting over the n residuals by the whole block; R: n doubles of LDS
__device__ __forceinline__ double track_threshold(const double* __restrict__ res,
                                                  int n, double* R,
                                                  double* s_thr) {
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    R[i] = __hip_atomic_load(res + i, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) *s_thr = 1e300;
  __syncthreads();
  int cnt = 0;
  for (int j = 0; j < n; ++j) cnt += R[j] < 1e300;   // (uniform: broadcasts)
  if (cnt > 0) {
    const int k = (cnt - 1) / 2;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double v = R[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        const double u = R[j];
        rank += (u < v) || (u == v && j < i);
      }
      if (rank == k) *s_thr = 10.0 * v;
    }
  }
  __syncthreads();
  return *s_thr;
}

// per-tile operand scratch of the deferred weight-gradient contraction
// (floats): feature-major matrices M[f][pt] = 512 floats
constexpr int SC_C = 0;                 // grid features c
constexpr int SC_H = 512;               // h_0..h_4
constexpr int SC_G = SC_H + 5 * 512;    // dL/dh_0..4 (gh_i)
constexpr int SC_M = SC_G + 5 * 512;    // ReLU masks: [5][32] words, bit = pt
constexpr int SC_P = SC_M + 160;        // positions [16][4]
constexpr int SC_GO = SC_P + 64;        // dL/d decoder output [16][4]
constexpr int SC_TILE = SC_GO + 64;     // 5920 floats = 23 680 B
static_assert(SC_TILE % 4 == 0, "16-byte aligned tiles");

// write a D-layout register pair (features 16jt+4q+r of point li) to a
// feature-major matrix
__device__ __forceinline__ void scr_put(float* __restrict__ M, int lane,
                                        const f32x4 (&v)[2]) {
  const int q = lane >> 4, li = lane & 15;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) M[(16 * jt + 4 * q + r) * 16 + li] = v[jt][r];
}

// Colour decoder backward of one tile (gc = d loss / d grid features, gp += d
// loss / d position) that leaves the weight-gradient operands in the tile's
// scratch and adds the tile's embedder._B gradient to embB (LDS [3][96]).
// w: the staged backward fragments (MlpPack offsets).
template <bool NEED_DP>
__device__ __forceinline__ void color_bwd_emit(
    const float* __restrict__ w, int lane, const float (&p)[1][3],
    const float (&go)[1][4], uint64_t mask, float* __restrict__ tsc,
    float* __restrict__ embB, float* __restrict__ park,
    f32x4 (&gc)[1][2], float (&gp)[1][3]) {
  using P = MlpPack<32, 4>;
  const int q = lane >> 4, li = lane & 15;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // ga_3 waits for the embedding backward in the wave's LDS scatter tile
  // (park, 512 floats, idle until grid_scatter) instead of 8 registers held
  // across three layers
  f32x4 gh[2] = {z4, z4}, ga[2] = {z4, z4};
  gc[0][0] = z4;
  gc[0][1] = z4;
  if (q == 0) {
    *reinterpret_cast<f32x4*>(tsc + SC_P + li * 4) =
        f32x4{p[0][0], p[0][1], p[0][2], 0.f};
    *reinterpret_cast<f32x4*>(tsc + SC_GO + li * 4) =
        f32x4{go[0][0], go[0][1], go[0][2], go[0][3]};
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const f32x4 w0 =
        *reinterpret_cast<const f32x4*>(w + P::WOUT + o * 32 + 4 * q);
    const f32x4 w1 =
        *reinterpret_cast<const f32x4*>(w + P::WOUT + o * 32 + 16 + 4 * q);
    gh[0] += w0 * go[0][o];
    gh[1] += w1 * go[0][o];
  }
  uint32_t* mw = reinterpret_cast<uint32_t*>(tsc + SC_M);
  // fragments read ahead as in mlp_bwd_ra: pts_linears.i's before the MFMAs
  // of fc_c.i's, fc_c.(i-1)'s before the MFMAs of pts_linears.i
  float awc[2][8];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 8; ++s)
      awc[kt][s] = w[P::wct(4) + (kt * 8 + s) * 64 + lane];
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    float awh[2][8];
    if (i >= 1) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 8; ++s)
          awh[kt][s] = w[P::wht(i) + (kt * 8 + s) * 64 + lane];
    }
    scr_put(tsc + SC_G + 512 * i, lane, gh);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on = (mask >> (i * 8 + jt * 4 + r)) & 1;
        ga[jt][r] = on ? gh[jt][r] : 0.f;
        // lane (q, li) -> ballot bit 16q+li: the 16 points of feature
        // 16jt+4q+r are bits [16q, 16q+16)
        const uint64_t b = __ballot(on);
        if (li == 0)
          mw[i * 32 + 16 * jt + 4 * q + r] = (uint32_t)(b >> (16 * q)) & 0xffffu;
      }
    XRD_SB();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 8; ++s)
        gc[0][kt] = XRD_MFMA4(awc[kt][s], gh[s >> 2][s & 3], gc[0][kt]);
    XRD_SB();
    if (i >= 1) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 8; ++s)
          awc[kt][s] = w[P::wct(i - 1) + (kt * 8 + s) * 64 + lane];
    }
    if (i == 3) {
      *reinterpret_cast<f32x4*>(park + lane * 8) = ga[0];
      *reinterpret_cast<f32x4*>(park + lane * 8 + 4) = ga[1];
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
  // ga holds the masked ga_0.  d loss / d sin(p.B) = W0^T ga0 + W3e^T ga3,
  // through the sine; lane group q owns feature k = emap(4kt+r, q)
  const f32x4 ga3[2] = {*reinterpret_cast<const f32x4*>(park + lane * 8),
                        *reinterpret_cast<const f32x4*>(park + lane * 8 + 4)};
  float a3[8], a0[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    a3[s] = w[P::W3ET + s * 64 + lane];
    a0[s] = w[P::W0T + s * 64 + lane];
  }
#pragma unroll 1
  for (int kt = 0; kt < 6; ++kt) {
    float n3[8], n0[8];
    if (kt < 5) {   // column tile kt+1 lands under the MFMAs of tile kt
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        n3[s] = w[P::W3ET + ((kt + 1) * 8 + s) * 64 + lane];
        n0[s] = w[P::W0T + ((kt + 1) * 8 + s) * 64 + lane];
      }
    }
    XRD_SB();
    f32x4 ge = z4;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      ge = XRD_MFMA4(a3[s], ga3[s >> 2][s & 3], ge);
      ge = XRD_MFMA4(a0[s], ga[s >> 2][s & 3], ge);
    }
    XRD_SB();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = emap(4 * kt + r, q);
      const f32x4 bk = *reinterpret_cast<const f32x4*>(w + P::EMB + k * 4);
      const float garg = ge[r] * cos_cw(embed_arg(p[0], bk));
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (NEED_DP) gp[0][a] += garg * bk[a];
        // embedder._B[a][k] += sum over the tile's points of p_a * garg
        const float v = row16_sum_dpp(p[0][a] * garg);
        if (li == 0 && k < kEmbK) atomicAdd(embB + a * 96 + k, v);
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

// ---- deferred weight-gradient contraction ---------------------------------
// lane (m = l & 15, q = l >> 4): an A operand row is feature 16jt+m, a B
// operand column feature 16kt+m; K-step s of a tile = point 4q+s, i.e.
// element s of the lane's 16-byte read from a feature-major matrix.
//
// Round 6: the operands come through LDS.  Rounds 3-5 let every unit read its
// operands from the tile scratch with global loads, two tiles a batch: the
// scratch was written a few microseconds earlier by this CU, but 9 MB per XCD
// are in flight (L2: 4 MB), so a batch is a round trip to the Infinity Cache
// (~2-3 us under the launch's own load) — six dependent round trips a unit,
// two units a wave: 35 us of a 182 us block for 10 us of MFMA work
// (profiles/r04_nice_map_phases.txt), and every operand was fetched 2-10
// times (c by ten units).  The registers of a 168-VGPR wave cannot hold more
// tiles in flight; the LDS can: after the last decoder pass the whole LDS of
// the block is idle.  So the block copies the group's tiles into a ring of
// two buffers of three tiles (2 x 70 KB) with LDS-DMA loads
// (global_load_lds_dwordx4: 1 KB a wave instruction, no registers), round
// r+1 in flight while round r is contracted, each operand fetched ONCE, and
// every repeated read is a ds_read_b128.  One barrier a round.
//
// The 62 16x16 output blocks over the 12 waves so that every SIMD (waves w,
// w+4, w+8) issues the same number of MFMAs a tile (64, 64, 64, 56):
//   waves 0-5:  Fourier column tile kt6 = wave: pts_linears.0 / .3 Fourier
//               parts, both row tiles (4 blocks sharing one B = sin(p.B))
//   wave 8 / 9: fc_c.0 + fc_c.1 / fc_c.2 + fc_c.3 (8 blocks sharing B = c)
//   wave 6:     fc_c.4 (4 blocks) + output_linear (2 blocks, B = h_4)
//   wave 10:    pts_linears.1 hidden (4) + pts_linears.4 rows 0-15 (2)
//   wave 7:     pts_linears.2 hidden (4) + pts_linears.4 rows 16-31 (2)
//   wave 11:    pts_linears.3 hidden (4)
// Every block accumulates its tiles in the order of rounds 3-5 (tile 0, 1,
// ..., K-steps 0-3 each): the sums are bit for bit the earlier ones.
__device__ __forceinline__ f32x4 ldm(const float* __restrict__ M, int f,
                                     int q) {
  return *reinterpret_cast<const f32x4*>(M + f * 16 + 4 * q);
}

constexpr int kRingTiles = 3;                      // tiles a round
constexpr int kRingFloats = kRingTiles * SC_TILE;  // 17 760 floats = 71 040 B
constexpr int kRingRounds = 12 / kRingTiles;       // a group = 12 tiles
constexpr int kRingChunks = (kRingFloats + 255) / 256;  // 70 x 1 KB
constexpr int kRingStride = kRingChunks * 256;     // floats between buffers
static_assert(12 % kRingTiles == 0 && kRingFloats % 4 == 0, "whole rounds");

// this wave's share of one round: src[0, kRingFloats) -> LDS byte address dst.
// Whole 1 KB chunks, chunk c by wave c % 12; the last chunk is partial: its
// lanes beyond the round re-read the round's last 16 bytes (they land in the
// pad between the two buffers) — no lane mask, no read beyond the scratch
__device__ __forceinline__ void ring_issue(const float* __restrict__ src,
                                           uint32_t dst, int wave, int lane) {
  constexpr uint32_t kLastBytes = (kRingFloats - (kRingChunks - 1) * 256) * 4;
  constexpr int kLast = kRingChunks - 1;
  const uint32_t voff = (uint32_t)lane * 16u;
#pragma unroll
  for (int k = 0; k < (kRingChunks + 11) / 12; ++k) {
    const int c = wave + 12 * k;   // wave-uniform
    if (k == kLast / 12 && c == kLast) {
      uint32_t vl = voff < kLastBytes ? voff : kLastBytes - 16u;
      lds_dma16(src + c * 256, vl, dst + (uint32_t)c * 1024u);
    } else if (c < kRingChunks) {
      lds_dma16(src + c * 256, voff, dst + (uint32_t)c * 1024u);
    }
  }
}
__device__ __forceinline__ void ring_landed() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// A block adds its weight-gradient blocks to ITS row of a [blocks][flat]
// partial buffer with plain loads and stores (nobody else touches the row; the
// finishing launch sums the rows): measured, the same adds as float atomics
// into 8 shared replicas cost 29 us a launch.
// first: the block's first group of this launch — a plain store, no read of
// a line that sits in HBM.  Every entry of a block's row is stored by its
// first group (62 blocks + bias rows + the output rows; embedder._B at the end
// of the kernel), so the rows need no zeroing between launches
// (tests/test_nice_hip.py::test_partial_rows_need_no_zeroing poisons them)
__device__ __forceinline__ void padd(float* __restrict__ p, float v,
                                     bool first) {
  if (first)
    *p = v;
  else
    *p += v;
}

// Operand addresses inside a tile are (one per-lane base) + (a compile-time
// offset): Tl = T + 16 m + 4 q (the lane's 16 bytes of row m of a feature-major
// matrix), Tm = T + SC_M + m (mask words), Tq = T + 16 q (point rows).  The
// blocks of a wave are template parameters: with run-time block tables the
// compiler formed every block's address per lane ahead of the round loop and
// spilled them — and a scratch reload inside the loop waits on vmcnt, i.e.
// for the LDS-DMA loads in flight.
__device__ __forceinline__ f32x4 ld4(const float* __restrict__ p) {
  return *reinterpret_cast<const f32x4*>(p);
}

// Fourier parts of pts_linears.0 / .3, column tile kt6 (features 16kt6+m of
// sin(p.B), recomputed): A = ga_0 / ga_3, both row tiles = 4 blocks sharing
// one B.  kt6 == 0 also carries pts_linears.0.bias = sum ga_0.
__device__ __forceinline__ void dwl_fourier_tile(const float* __restrict__ Tl,
                                                 const float* __restrict__ Tm,
                                                 const float* __restrict__ Tq,
                                                 int q, const f32x4 bk,
                                                 f32x4 (&acc)[8],
                                                 float (&bias)[4]) {
  f32x4 pp[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) pp[s] = ld4(Tq + SC_P + 4 * s);
  const f32x4 a00 = ld4(Tl + SC_G), a01 = ld4(Tl + SC_G + 256);
  const f32x4 a30 = ld4(Tl + SC_G + 3 * 512);
  const f32x4 a31 = ld4(Tl + SC_G + 3 * 512 + 256);
  const uint32_t* mw = reinterpret_cast<const uint32_t*>(Tm);
  const uint32_t m00 = mw[0] >> (4 * q), m01 = mw[16] >> (4 * q);
  const uint32_t m30 = mw[3 * 32] >> (4 * q);
  const uint32_t m31 = mw[3 * 32 + 16] >> (4 * q);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float pv[3] = {pp[s][0], pp[s][1], pp[s][2]};
    const float e = sin_cw(embed_arg(pv, bk));
    const float v00 = ((m00 >> s) & 1u) ? a00[s] : 0.f;
    const float v01 = ((m01 >> s) & 1u) ? a01[s] : 0.f;
    const float v30 = ((m30 >> s) & 1u) ? a30[s] : 0.f;
    const float v31 = ((m31 >> s) & 1u) ? a31[s] : 0.f;
    acc[0] = XRD_MFMA4(v00, e, acc[0]);
    acc[1] = XRD_MFMA4(v01, e, acc[1]);
    acc[2] = XRD_MFMA4(v30, e, acc[2]);
    acc[3] = XRD_MFMA4(v31, e, acc[3]);
    bias[0] += v00;
    bias[1] += v01;
  }
}
__device__ __forceinline__ void dwl_fourier_flush(int kt6, int m, int q,
                                                  const f32x4 (&acc)[8],
                                                  const float (&bias)[4],
                                                  float* __restrict__ rep,
                                                  bool first) {
  using F = MlpFlat<32, 4>;
  const int k = 16 * kt6 + m;
  if (k < kEmbK) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jt + 4 * q + r;
        padd(rep + F::P0W + j * kEmbK + k, acc[jt][r], first);
        padd(rep + F::P3W + j * (kEmbK + 32) + k, acc[2 + jt][r], first);
      }
  }
  if (kt6 == 0) {
    const float s0 = group4_sum(bias[0]), s1 = group4_sum(bias[1]);
    if (q == 0) {
      padd(rep + F::P0B + m, s0, first);
      padd(rep + F::P0B + 16 + m, s1, first);
    }
  }
}

// One A row tile against both column tiles of one B matrix (2 blocks):
//   HID = false: fc_c.I rows JT,    A = gh_I,          B = c
//   HID = true:  pts_linears.I,     A = masked gh_I,   B = h_{I-1}
template <bool HID, int I, int JT>
__device__ __forceinline__ void dwl_pair_tile(const float* __restrict__ Tl,
                                              const float* __restrict__ Tm,
                                              int q, f32x4& acc0, f32x4& acc1,
                                              float& bias) {
  constexpr int BOFF = HID ? SC_H + 512 * (I - 1) : SC_C;
  const f32x4 a = ld4(Tl + SC_G + 512 * I + 256 * JT);
  const f32x4 b0 = ld4(Tl + BOFF), b1 = ld4(Tl + BOFF + 256);
  const uint32_t mw =
      HID ? reinterpret_cast<const uint32_t*>(Tm)[I * 32 + 16 * JT] >> (4 * q)
          : 0xfu;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float av = ((mw >> s) & 1u) ? a[s] : 0.f;  // ga: masked gh
    acc0 = XRD_MFMA4(av, b0[s], acc0);
    acc1 = XRD_MFMA4(av, b1[s], acc1);
    bias += av;
  }
}
template <bool HID, int I, int JT>
__device__ __forceinline__ void dwl_pair_flush(int m, int q, const f32x4 acc0,
                                               const f32x4 acc1, float bias,
                                               float* __restrict__ rep,
                                               bool first) {
  using F = MlpFlat<32, 4>;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = 16 * JT + 4 * q + r;
    float* dst = HID ? rep + F::pw(I) + j * F::pstride(I) + F::pcol(I)
                     : rep + F::fcw(I) + j * 32;
    padd(dst + m, acc0[r], first);
    padd(dst + 16 + m, acc1[r], first);
  }
  const float b = group4_sum(bias);
  if (q == 0) padd(rep + (HID ? F::pb(I) : F::fcb(I)) + 16 * JT + m, b, first);
}
// output_linear: rows = the 4 outputs (A = d loss / d output), B = h_4
__device__ __forceinline__ void dwl_output_tile(const float* __restrict__ Tl,
                                                const float* __restrict__ Tq,
                                                int m, f32x4& acc0,
                                                f32x4& acc1, float& bout) {
  const f32x4 b0 = ld4(Tl + SC_H + 4 * 512);
  const f32x4 b1 = ld4(Tl + SC_H + 4 * 512 + 256);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float av = m < 4 ? Tq[SC_GO + 4 * s + (m & 3)] : 0.f;
    acc0 = XRD_MFMA4(av, b0[s], acc0);
    acc1 = XRD_MFMA4(av, b1[s], acc1);
    bout += av;
  }
}
__device__ __forceinline__ void dwl_output_flush(int m, int q, const f32x4 acc0,
                                                 const f32x4 acc1, float bout,
                                                 float* __restrict__ rep,
                                                 bool first) {
  using F = MlpFlat<32, 4>;
  if (q == 0) {  // rows 0..3 of the accumulator = lane group 0
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      padd(rep + F::OW + r * 32 + m, acc0[r], first);
      padd(rep + F::OW + r * 32 + 16 + m, acc1[r], first);
    }
  }
  const float b = group4_sum(bout);
  if (q == 0 && m < 4) padd(rep + F::OB + m, b, first);
}

// the blocks of layer wave 6 + ROLE on one tile / at the end
template <int ROLE>
__device__ __forceinline__ void dwl_role_tile(const float* __restrict__ Tl,
                                              const float* __restrict__ Tm,
                                              const float* __restrict__ Tq,
                                              int m, int q, f32x4 (&a)[8],
                                              float (&b)[4]) {
  if (ROLE == 0) {         // wave 6: fc_c.4 + output_linear
    dwl_pair_tile<false, 4, 0>(Tl, Tm, q, a[0], a[1], b[0]);
    dwl_pair_tile<false, 4, 1>(Tl, Tm, q, a[2], a[3], b[1]);
    dwl_output_tile(Tl, Tq, m, a[4], a[5], b[2]);
  } else if (ROLE == 1) {  // wave 7: pts_linears.2, .4 rows 16-31
    dwl_pair_tile<true, 2, 0>(Tl, Tm, q, a[0], a[1], b[0]);
    dwl_pair_tile<true, 2, 1>(Tl, Tm, q, a[2], a[3], b[1]);
    dwl_pair_tile<true, 4, 1>(Tl, Tm, q, a[4], a[5], b[2]);
  } else if (ROLE == 2) {  // wave 8: fc_c.0, fc_c.1
    dwl_pair_tile<false, 0, 0>(Tl, Tm, q, a[0], a[1], b[0]);
    dwl_pair_tile<false, 0, 1>(Tl, Tm, q, a[2], a[3], b[1]);
    dwl_pair_tile<false, 1, 0>(Tl, Tm, q, a[4], a[5], b[2]);
    dwl_pair_tile<false, 1, 1>(Tl, Tm, q, a[6], a[7], b[3]);
  } else if (ROLE == 3) {  // wave 9: fc_c.2, fc_c.3
    dwl_pair_tile<false, 2, 0>(Tl, Tm, q, a[0], a[1], b[0]);
    dwl_pair_tile<false, 2, 1>(Tl, Tm, q, a[2], a[3], b[1]);
    dwl_pair_tile<false, 3, 0>(Tl, Tm, q, a[4], a[5], b[2]);
    dwl_pair_tile<false, 3, 1>(Tl, Tm, q, a[6], a[7], b[3]);
  } else if (ROLE == 4) {  // wave 10: pts_linears.1, .4 rows 0-15
    dwl_pair_tile<true, 1, 0>(Tl, Tm, q, a[0], a[1], b[0]);
    dwl_pair_tile<true, 1, 1>(Tl, Tm, q, a[2], a[3], b[1]);
    dwl_pair_tile<true, 4, 0>(Tl, Tm, q, a[4], a[5], b[2]);
  } else {                 // wave 11: pts_linears.3
    dwl_pair_tile<true, 3, 0>(Tl, Tm, q, a[0], a[1], b[0]);
    dwl_pair_tile<true, 3, 1>(Tl, Tm, q, a[2], a[3], b[1]);
  }
}
template <int ROLE>
__device__ __forceinline__ void dwl_role_flush(int m, int q,
                                               const f32x4 (&a)[8],
                                               const float (&b)[4],
                                               float* __restrict__ rep,
                                               bool first) {
  if (ROLE == 0) {
    dwl_pair_flush<false, 4, 0>(m, q, a[0], a[1], b[0], rep, first);
    dwl_pair_flush<false, 4, 1>(m, q, a[2], a[3], b[1], rep, first);
    dwl_output_flush(m, q, a[4], a[5], b[2], rep, first);
  } else if (ROLE == 1) {
    dwl_pair_flush<true, 2, 0>(m, q, a[0], a[1], b[0], rep, first);
    dwl_pair_flush<true, 2, 1>(m, q, a[2], a[3], b[1], rep, first);
    dwl_pair_flush<true, 4, 1>(m, q, a[4], a[5], b[2], rep, first);
  } else if (ROLE == 2) {
    dwl_pair_flush<false, 0, 0>(m, q, a[0], a[1], b[0], rep, first);
    dwl_pair_flush<false, 0, 1>(m, q, a[2], a[3], b[1], rep, first);
    dwl_pair_flush<false, 1, 0>(m, q, a[4], a[5], b[2], rep, first);
    dwl_pair_flush<false, 1, 1>(m, q, a[6], a[7], b[3], rep, first);
  } else if (ROLE == 3) {
    dwl_pair_flush<false, 2, 0>(m, q, a[0], a[1], b[0], rep, first);
    dwl_pair_flush<false, 2, 1>(m, q, a[2], a[3], b[1], rep, first);
    dwl_pair_flush<false, 3, 0>(m, q, a[4], a[5], b[2], rep, first);
    dwl_pair_flush<false, 3, 1>(m, q, a[6], a[7], b[3], rep, first);
  } else if (ROLE == 4) {
    dwl_pair_flush<true, 1, 0>(m, q, a[0], a[1], b[0], rep, first);
    dwl_pair_flush<true, 1, 1>(m, q, a[2], a[3], b[1], rep, first);
    dwl_pair_flush<true, 4, 0>(m, q, a[4], a[5], b[2], rep, first);
  } else {
    dwl_pair_flush<true, 3, 0>(m, q, a[0], a[1], b[0], rep, first);
    dwl_pair_flush<true, 3, 1>(m, q, a[2], a[3], b[1], rep, first);
  }
}

// ring: the block's LDS from offset 0 (every other use of it is over: the
// caller's barrier).  scr: the group's 12 tile slots, contiguous.
__device__ __forceinline__ void dw_contract(const float* __restrict__ scr,
                                            int ntiles, int wave, int lane,
                                            const float* __restrict__ dec,
                                            float* __restrict__ ring,
                                            float* __restrict__ rep,
                                            bool first) {
  using P = MlpPack<32, 4>;
  const int m = lane & 15, q = lane >> 4;
  const uint32_t ring_b = lds_addr(ring);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[8] = {z4, z4, z4, z4, z4, z4, z4, z4};
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 bk = z4;
  if (wave < 6)
    bk = *reinterpret_cast<const f32x4*>(dec + P::EMB + (16 * wave + m) * 4);
  ring_issue(scr, ring_b, wave, lane);
#pragma unroll 1
  for (int r = 0; r < kRingRounds; ++r) {
    ring_landed();
    __syncthreads();  // round r is in its buffer; round r-1's buffer is free
    if (r + 1 < kRingRounds && (r + 1) * kRingTiles < ntiles)
      ring_issue(scr + (size_t)(r + 1) * kRingFloats,
                 ring_b + ((r + 1) & 1) * (kRingStride * 4), wave, lane);
    const float* buf = ring + (r & 1) * kRingStride;
#pragma unroll 1
    for (int t = 0; t < kRingTiles; ++t) {
      if (r * kRingTiles + t >= ntiles) break;  // tiles beyond the group
      const float* T = buf + t * SC_TILE;
      const float* Tl = T + 16 * m + 4 * q;
      const float* Tm = T + SC_M + m;
      const float* Tq = T + 16 * q;
      switch (wave) {
        case 6: dwl_role_tile<0>(Tl, Tm, Tq, m, q, acc, bias); break;
        case 7: dwl_role_tile<1>(Tl, Tm, Tq, m, q, acc, bias); break;
        case 8: dwl_role_tile<2>(Tl, Tm, Tq, m, q, acc, bias); break;
        case 9: dwl_role_tile<3>(Tl, Tm, Tq, m, q, acc, bias); break;
        case 10: dwl_role_tile<4>(Tl, Tm, Tq, m, q, acc, bias); break;
        case 11: dwl_role_tile<5>(Tl, Tm, Tq, m, q, acc, bias); break;
        default: dwl_fourier_tile(Tl, Tm, Tq, q, bk, acc, bias); break;
      }
    }
  }
  switch (wave) {
    case 6: dwl_role_flush<0>(m, q, acc, bias, rep, first); break;
    case 7: dwl_role_flush<1>(m, q, acc, bias, rep, first); break;
    case 8: dwl_role_flush<2>(m, q, acc, bias, rep, first); break;
    case 9: dwl_role_flush<3>(m, q, acc, bias, rep, first); break;
    case 10: dwl_role_flush<4>(m, q, acc, bias, rep, first); break;
    case 11: dwl_role_flush<5>(m, q, acc, bias, rep, first); break;
    default: dwl_fourier_flush(wave, m, q, acc, bias, rep, first); break;
  }
}

template <int NT>
struct MapGeom {
  static constexpr int RPBM = 4;             // rays per block
  static constexpr int NW = RPBM * NT;       // waves = tiles of a group
  static constexpr int RAW = kWMax;          // raw [RPBM][64][4]
  static constexpr int WAVE0 = kWMax + RPBM * 256;  // per-wave scratch
  // embedder._B sums: behind the per-wave scratch AND behind the weight-
  // gradient contraction's operand ring, which takes the LDS from offset 0
  static constexpr int EMBB = WAVE0 + NW * kScratch > 2 * kRingStride
                                  ? WAVE0 + NW * kScratch
                                  : 2 * kRingStride;
  static constexpr size_t LDS = (size_t)EMBB + 288;
};
static_assert(MapGeom<3>::LDS * 4 <= 163840, "LDS per CU");
static_assert(MapGeom<3>::NW == 12, "the contraction's unit table: 12 waves");

template <int STAGE, int NT, bool NEED_DP, bool NEED_DW, bool TRACK = false>
__global__ __launch_bounds__((MapGeom<NT>::NW * 64),
                             ((MapGeom<NT>::NW + 3) / 4)) void
nice_map_fused_kernel(
    xrd_nice_scene sc, int n, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ gt_depth,
    const float* __restrict__ dmax_p, const float* __restrict__ tgt_rgb,
    const uint8_t* __restrict__ keep, float w_color, float* gg_middle,
    float* gg_fine, float* gg_color, double* __restrict__ part,
    float* __restrict__ dw_rep, float* __restrict__ dw_scr,
    double* __restrict__ ray_loss, TrackArgs trk) {
  static_assert(!NEED_DW || STAGE == XRD_STAGE_COLOR, "dW: colour stage");
  static_assert(!TRACK || (STAGE == XRD_STAGE_COLOR && NEED_DP && !NEED_DW),
                "tracking: colour stage, ray gradients only");
  using G = MapGeom<NT>;
  constexpr int S = NT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);  // staged fragments
  float* rawbuf = wl + G::RAW;
  const int wave = threadIdx.x >> 6, lane0 = threadIdx.x & 63;
  const int slot = wave / NT, tile = wave % NT;
  float* scratch = wl + G::WAVE0 + wave * kScratch;
  float* embB = wl + G::EMBB;
  double* zbuf = reinterpret_cast<double*>(scratch);
  ScatterLds SL;
  SL.gt = scratch + 256;
  SL.off = reinterpret_cast<int*>(SL.gt + 16 * 33);
  SL.w = SL.gt + 16 * 33 + 16 * 8;
  if (NEED_DW) {
    for (int i = threadIdx.x; i < 288; i += blockDim.x) embB[i] = 0.f;
    // (visible to every wave behind the first staging barrier)
  }
  using PM = MlpPack<32, 1>;
  using PF = MlpPack<64, 1>;
  using PC = MlpPack<32, 4>;
  const int ngroups = (n + G::RPBM - 1) / G::RPBM;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int ray = __builtin_amdgcn_readfirstlane(grp * G::RPBM + slot);
    const bool active = ray < n;
    // the lane index is re-materialised per group and per decoder phase
    // (empty asm): without it hipcc hoists the phases' LDS / scratch address
    // arithmetic out of the group loop and keeps it in registers across every
    // phase (0.7-0.8 KB of spills a lane in the weight-gradient variants)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int q = lane >> 4, li = lane & 15;
    // this tile's operand scratch (weight-gradient contraction)
    float* tsc = NEED_DW ? dw_scr + ((size_t)grp * G::NW + wave) * SC_TILE
                         : nullptr;
    double zl = 0.0;
    float gd = 0.f;
    TileGeom tg = {};
    float p32[1][3] = {{0.f, 0.f, 0.f}};
    float xn[3] = {0.f, 0.f, 0.f};   // the point in the bound's [-1, 1]^3
    float occ = 0.f, col[3] = {0.f, 0.f, 0.f};
    uint64_t mask_m[1] = {0}, mask_f[1] = {0}, mask_c[1] = {0};
    f32x4 c_m[1][2], c_c[1][2];
    Tri tr;
    // ---- forward: middle -> fine -> colour, each decoder ONCE ------------
    if (active) {
      RayCtx rc;
      load_ray(rays_o, rays_d, gt_depth, ray, true, rc);
      gd = rc.gd;
      zl = sample_z<S>(sc, rc, dmax_p[0], lane, zbuf, zbuf + 64);
      tile_geom(rc, zbuf[64 + 16 * tile + li], sc.bound, tg);
#pragma unroll
      for (int a = 0; a < 3; ++a) p32[0][a] = tg.p32[a];
      tri_norm(tg.p64, sc.bound, xn);   // once for all six lookups
      tri_prepare_n<NEED_DP>(xn, sc.bound, sc.gdim + 3, tr);
      tri_gather(sc.grid[1], tr, q, c_m[0]);
    }
    stage_weights(wl, sc.dec[1], PM::WHT);
    asm volatile("" : "+v"(lane));
    if (active) {
      float om[1];
      mlp_fwd_ra<32, 1, true, false>(wl, lane, p32[0], c_m[0], om, mask_m[0],
                                     nullptr);
      occ = om[0];
    }
    if (STAGE >= XRD_STAGE_FINE) {
      f32x4 c_f[1][4];
      if (active) {
        f32x4 cf[2];
        tri_prepare_n<NEED_DP>(xn, sc.bound, sc.gdim + 6, tr);
        tri_gather(sc.grid[2], tr, q, cf);
        c_f[0][0] = cf[0];
        c_f[0][1] = cf[1];
        c_f[0][2] = c_m[0][0];
        c_f[0][3] = c_m[0][1];
      }
      stage_weights(wl, sc.dec[2], PF::WHT);
      asm volatile("" : "+v"(lane));
      if (active) {
        float of[1];
        mlp_fwd_ra<64, 1, true, false>(wl, lane, p32[0], c_f[0], of, mask_f[0],
                                       nullptr);
        occ = of[0] + occ;  // NICE.forward: fine_occ + middle_occ
      }
    }
    if (STAGE == XRD_STAGE_COLOR) {
      if (active) {
        tri_prepare_n<NEED_DP>(xn, sc.bound, sc.gdim + 9, tr);
        tri_gather(sc.grid[3], tr, q, c_c[0]);
        if (NEED_DW) scr_put(tsc + SC_C, lane, c_c[0]);
      }
      stage_weights(wl, sc.dec[3], PC::WHT);
      asm volatile("" : "+v"(lane));
      if (active) {
        float oc[4];
        mlp_fwd_ra<32, 4, true, NEED_DW>(wl, lane, p32[0], c_c[0], oc,
                                         mask_c[0], tsc + SC_H);
        col[0] = oc[0];
        col[1] = oc[1];
        col[2] = oc[2];
      }
    }
    if (active) {
      if (!tg.inb) occ = 100.f;  // conv_onet.py:370
      if (q == 0)
        *reinterpret_cast<f32x4*>(rawbuf + (slot * 64 + 16 * tile + li) * 4) =
            f32x4{col[0], col[1], col[2], occ};
    }
    __syncthreads();
    // ---- compositing, loss, compositing backward (every wave: its ray) ----
    float gocc = 0.f, gcol[3] = {0.f, 0.f, 0.f};
    if constexpr (TRACK) {
      // forward half of every ray, the residuals meet at the grid barrier,
      // the block selects the batch median, the backward half follows
      f32x4 rw = {0.f, 0.f, 0.f, 0.f};
      float alpha = 0.f, oma = 1.f, crgb[3] = {0.f, 0.f, 0.f};
      double T = 1.0, wd = 0.0, dep = 0.0, var = 0.0;
      const bool kept = active && (keep == nullptr || keep[ray] != 0);
      if (active)
        track_forward<S>(rawbuf + slot * 256, lane, zl, rw, alpha, oma, T, wd,
                         dep, var, crgb);
      const double inv = 1.0 / sqrt(var + 1e-10);
      const double diff = (double)gd - dep;
      const double res = fabs(diff) * inv;
      if (active && tile == 0 && lane == 0)
        __hip_atomic_store(trk.res + ray, kept ? res : 1e300,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      grid_barrier_once(trk.bar);
      // (the forward's staged colour-decoder fragments are dead: their LDS
      // holds the residuals until the backward stages its own)
      const double thr =
          trk.handle_dynamic
              ? track_threshold(trk.res, n, reinterpret_cast<double*>(wl),
                                reinterpret_cast<double*>(embB))
              : 1e300;
      if (active) {
        const bool m = kept && gd > 0.f &&
                       (!trk.handle_dynamic || res < thr);
        const double gdep =
            m ? (diff > 0 ? -1.0 : (diff < 0 ? 1.0 : 0.0)) * inv : 0.0;
        float grgb[3] = {0.f, 0.f, 0.f};
        float lc = 0.f;
        if (m && trk.use_color) {
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const float dc = tgt_rgb[ray * 3 + a] - crgb[a];
            lc += fabsf(dc);
            grgb[a] = w_color * (dc > 0.f ? -1.f : (dc < 0.f ? 1.f : 0.f));
          }
        }
        const float gocc_s =
            track_backward<S>(lane, zl, rw, alpha, oma, T, wd, gdep, grgb);
        const int src = 16 * tile + li;
        gocc = __shfl(gocc_s, src);
        if (!tg.inb) gocc = 0.f;  // occupancy was overridden to 100
        const float wsrc = __shfl((float)wd, src);
#pragma unroll
        for (int a = 0; a < 3; ++a) gcol[a] = grgb[a] * wsrc;
        if (tile == 0 && lane == 0) {
          ray_loss[ray] = m ? res : 0.0;
          trk.ray_lossc[ray] = (double)lc;
        }
      }
    } else if (active) {
      float gocc_s, w, grgb[3];
      double loss;
      const bool kept = keep == nullptr || keep[ray] != 0;
      map_composite<S>(rawbuf + slot * 256, lane, zl, gd, tgt_rgb + ray * 3,
                       kept, STAGE == XRD_STAGE_COLOR, w_color, gocc_s, w,
                       grgb, loss);
      const int src = 16 * tile + li;
      gocc = __shfl(gocc_s, src);
      if (!tg.inb) gocc = 0.f;  // occupancy was overridden to 100
      const float wsrc = __shfl(w, src);
#pragma unroll
      for (int a = 0; a < 3; ++a) gcol[a] = grgb[a] * wsrc;
      if (tile == 0 && lane == 0 && ray_loss != nullptr) ray_loss[ray] = loss;
    }
    if (!TRACK && trk.exp_g != nullptr && active && q == 0) {
      const size_t sidx = (size_t)ray * S + 16 * tile + li;
      trk.exp_g[sidx] = gocc;
#pragma unroll
      for (int a = 0; a < 3; ++a) trk.exp_p[sidx * 3 + a] = p32[0][a];
    }
    // ---- backward: colour -> fine -> middle --------------------------------
    asm volatile("" : "+v"(lane));
    double gp64[3] = {0.0, 0.0, 0.0};
    float gp32[1][3] = {{0.f, 0.f, 0.f}};
    if (STAGE == XRD_STAGE_COLOR) {
      f32x4 gc[1][2];
      // channel 3 is overwritten by fine+middle occupancy -> no gradient
      const float go[1][4] = {{gcol[0], gcol[1], gcol[2], 0.f}};
      stage_weights(wl, sc.dec[3] + PC::EMB,
                    ((NEED_DP || NEED_DW) ? PC::LEN : PC::W0T) - PC::EMB);
      if (NEED_DW) {
        if (active)
          color_bwd_emit<NEED_DP>(wl - PC::EMB, lane, p32, go, mask_c[0], tsc,
                                  embB, SL.gt, gc, gp32);
      } else if (active) {
        mlp_bwd_ra<32, 4, NEED_DP, NEED_DP>(wl - PC::EMB, lane, p32[0], go[0],
                                            mask_c[0], gc[0], gp32[0]);
      }
      if (active) {
        tri_prepare_n<NEED_DP>(xn, sc.bound, sc.gdim + 9, tr);
        if (NEED_DP) tri_backward_dp(sc.grid[3], tr, q, gc[0], gp64);
        if constexpr (!TRACK)
          grid_scatter(gg_color, sc.gmask[3], tr, lane, gc[0], SL);
      }
    }
    if (STAGE >= XRD_STAGE_FINE) {
      f32x4 c_f[1][4], gc[1][4];
      const float go[1][1] = {{gocc}};
      stage_weights(wl, sc.dec[2] + PF::EMB,
                    (NEED_DP ? PF::LEN : PF::W0T) - PF::EMB);
      asm volatile("" : "+v"(lane));
      if (active) {
        mlp_bwd_ra<64, 1, NEED_DP, NEED_DP>(wl - PF::EMB, lane, p32[0], go[0],
                                            mask_f[0], gc[0], gp32[0]);
        const f32x4 g2[2] = {gc[0][0], gc[0][1]};  // c_middle is no_grad
        tri_prepare_n<NEED_DP>(xn, sc.bound, sc.gdim + 6, tr);
        if (NEED_DP) tri_backward_dp(sc.grid[2], tr, q, g2, gp64);
        if constexpr (!TRACK)
          grid_scatter(gg_fine, sc.gmask[2], tr, lane, g2, SL);
      }
    }
    {
      const float go[1][1] = {{gocc}};
      f32x4 gc[1][2];
      stage_weights(wl, sc.dec[1] + PM::EMB,
                    (NEED_DP ? PM::LEN : PM::W0T) - PM::EMB);
      asm volatile("" : "+v"(lane));
      if (active) {
        mlp_bwd_ra<32, 1, NEED_DP, NEED_DP>(wl - PM::EMB, lane, p32[0], go[0],
                                            mask_m[0], gc[0], gp32[0]);
        tri_prepare_n<NEED_DP>(xn, sc.bound, sc.gdim + 3, tr);
        if (NEED_DP) tri_backward_dp(sc.grid[1], tr, q, gc[0], gp64);
        if constexpr (!TRACK)
          grid_scatter(gg_middle, sc.gmask[1], tr, lane, gc[0], SL);
      }
    }
    if (NEED_DW) {
      // every tile of the group has left its operands in the scratch.
      // Writers and readers are waves of ONE workgroup: the block barrier's
      // workgroup-scope release / acquire is enough (the lines were never
      // read before: no stale L1 copy).  An agent-scope __threadfence() here
      // makes every block write its XCD's dirty L2 lines back (the L2s of the
      // 8 XCDs are not coherent with each other): measured +55 us a launch.
      __syncthreads();
      const int rays_here = n - grp * G::RPBM < G::RPBM ? n - grp * G::RPBM
                                                          : G::RPBM;
      int lane_b = lane;
      asm volatile("" : "+v"(lane_b));  // keep the contraction's address
      // arithmetic inside the group loop (no hoisting into live registers)
      dw_contract(dw_scr + (size_t)grp * G::NW * SC_TILE, rays_here * NT,
                  __builtin_amdgcn_readfirstlane(wave), lane_b, sc.dec[3], wl,
                  dw_rep + (size_t)blockIdx.x * kColorFlat,
                  grp == (int)blockIdx.x);
      // (a further group's set-up writes the LDS the last round is read from)
      if (grp + (int)gridDim.x < ngroups) __syncthreads();
    }
    if (NEED_DP && active) {
      const int tile_id = ray * NT + tile;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double g = gp64[a] + (double)gp32[0][a];
        const double so = wave_sum(g), sd = wave_sum(g * tg.z);
        if (lane == 0) {
          part[(size_t)tile_id * 6 + a] = so;
          part[(size_t)tile_id * 6 + 3 + a] = sd;
        }
      }
    }
  }
  if (NEED_DW) {
    __syncthreads();  // the LDS sums of embedder._B are complete
    float* rep = dw_rep + (size_t)blockIdx.x * kColorFlat;
    for (int i = threadIdx.x; i < 288; i += blockDim.x) {
      const int a = i / 96, k = i % 96;
      // (once a launch, by the block that owns the row: a plain store)
      if (k < kEmbK) rep[MlpFlat<32, 4>::EB + a * kEmbK + k] = embB[i];
    }
  }
}

// Coarse stage (grid_coarse is its only parameter, conv_onet.py:187-195; 32
// uniform samples, no depth guidance).  A block = 4 rays = 8 tile waves with
// the decoder's fragments (forward + transposed, 50 KB) staged in LDS once per
// block; the gradient is scattered into one of kCoarseRep replicas
// (nice_map_coarse_finish_kernel).  Round 3 ran one ray per block with the
// fragments read from L2: a ray's forward + backward is a chain of ten layer
// passes, each behind an L2 round trip — 47 us a launch whether 200 or 1000
// rays were in it.  The coarse mapper runs next to the mapper on a side
// stream (NiceSLAM._coarse_on_side_stream); measured, it still cost 2.9 ms
// per mapping call + the 4 tracking frames after it (13 % of the frame time).
constexpr int kCoarseRPB = 4;                       // rays per block
constexpr int kCoarseWaves = 2 * kCoarseRPB;        // 2 tiles a ray
constexpr int kCoarseWaveLds = 256 + kScatterFloats;
constexpr size_t kCoarseLds =
    ((size_t)NoXyzPack::LEN + kCoarseWaves * kCoarseWaveLds +
     kCoarseRPB * 256) * sizeof(float);
static_assert(NoXyzPack::LEN % 4 == 0, "16-byte staging");
static_assert(kCoarseLds <= 163840, "LDS per CU");

__global__ __launch_bounds__(kCoarseWaves * 64, 2) void nice_map_coarse_kernel(
    xrd_nice_scene sc, int n, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ gt_depth,
    const uint8_t* __restrict__ keep, float* gg_coarse,
    float* __restrict__ rep, double* __restrict__ ray_loss) {
  constexpr int S = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);  // staged fragments
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = wave >> 1, tile = wave & 1;
  const int q = lane >> 4, li = lane & 15;
  float* R = wl + NoXyzPack::LEN + wave * kCoarseWaveLds;
  float* rawbuf =
      wl + NoXyzPack::LEN + kCoarseWaves * kCoarseWaveLds + slot * 256;
  double* zbuf = reinterpret_cast<double*>(R);
  ScatterLds SL;
  SL.gt = R + 256;
  SL.off = reinterpret_cast<int*>(SL.gt + 16 * 33);
  SL.w = SL.gt + 16 * 33 + 16 * 8;
  for (int i = threadIdx.x * 4; i < NoXyzPack::LEN; i += blockDim.x * 4)
    *reinterpret_cast<f32x4*>(wl + i) =
        *reinterpret_cast<const f32x4*>(sc.dec[0] + i);
  __syncthreads();
  const int ngroups = (n + kCoarseRPB - 1) / kCoarseRPB;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int ray = __builtin_amdgcn_readfirstlane(grp * kCoarseRPB + slot);
    const bool active = ray < n;
    double zl = 0.0;
    TileGeom tg = {};
    f32x4 c_a[1][2], gc[1][2];
    float o1[1] = {0.f};
    uint64_t mask[1] = {0};
    Tri tr;
    if (active) {
      RayCtx rc;
      load_ray(rays_o, rays_d, nullptr, ray, false, rc);
      zl = sample_z<S>(sc, rc, 0.f, lane, zbuf, zbuf + 64);
      tile_geom(rc, zbuf[64 + 16 * tile + li], sc.bound, tg);
      tri_prepare(tg.p64, sc.bound, sc.coarse_enlarge, sc.gdim + 0, tr);
      tri_gather(sc.grid[0], tr, q, c_a[0]);
      noxyz_fwd<1, true>(wl, lane, c_a, o1, mask);
      if (q == 0)
        *reinterpret_cast<f32x4*>(rawbuf + (16 * tile + li) * 4) =
            f32x4{0.f, 0.f, 0.f, tg.inb ? o1[0] : 100.f};
    }
    __syncthreads();
    if (active) {
      float gocc_s, w, grgb[3];
      double loss;
      const bool kept = keep == nullptr || keep[ray] != 0;
      map_composite<S>(rawbuf, lane, zl, gt_depth[ray], nullptr, kept, false,
                       0.f, gocc_s, w, grgb, loss);
      if (tile == 0 && lane == 0 && ray_loss != nullptr) ray_loss[ray] = loss;
      float gocc = __shfl(gocc_s, 16 * tile + li);
      if (!tg.inb) gocc = 0.f;
      const float go[1] = {gocc};
      noxyz_bwd<1>(wl, lane, go, mask, gc);
      tri_prepare(tg.p64, sc.bound, sc.coarse_enlarge, sc.gdim + 0, tr);
      float* ggc = gg_coarse;
      if (rep != nullptr && gg_coarse != nullptr)
        ggc = rep + (size_t)(ray & (kCoarseRep - 1)) *
                        ((size_t)sc.gdim[0] * sc.gdim[1] * sc.gdim[2] * 32);
      grid_scatter(ggc, sc.gmask[0], tr, lane, gc[0], SL);
    }
    __syncthreads();  // rawbuf is rewritten by the next group
  }
}

// after the fused launch: decoder gradient = sum of the blocks' partial rows
// (64 columns a block, the rows split over its
// four waves), ray gradients = sum of the ray's tile partials in a fixed
// order, loss = sum of the per-ray losses (last block)
constexpr int kFinishThreads = 1024;
__global__ __launch_bounds__(kFinishThreads) void nice_map_finish_kernel(
    float* __restrict__ rep, int n_rep, int len, int dec_blocks,
    float* __restrict__ g_dec, const double* __restrict__ part, int n_dp,
    int nt, float* __restrict__ g_rays_o, float* __restrict__ g_rays_d,
    const double* __restrict__ ray_loss, int n, double* __restrict__ loss) {
  __shared__ double sh[4];
  __shared__ float shf[kFinishThreads / 64][64];
  if (blockIdx.x == gridDim.x - 1) {
    // (four waves, as ever: the loss keeps its summation order)
    double s = 0.0;
    if (ray_loss != nullptr && threadIdx.x < 256)
      for (int i = threadIdx.x; i < n; i += 256) s += ray_loss[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256)
      sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && loss != nullptr)
      loss[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    return;
  }
  if ((int)blockIdx.x < dec_blocks) {
    // 64 columns a block, the blocks' partial rows split over its 16 waves
    // (round 4: over 4 waves — a chain of 64 load / store pairs a thread,
    // 12 us for the colour decoder's 250 x 15899 floats)
    constexpr int W = kFinishThreads / 64;
    const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + c;
    float s = 0.f;
    if (i < len) {
#pragma unroll 4
      for (int r = w; r < n_rep; r += W) s += rep[(size_t)r * len + i];
    }
    shf[w][c] = s;
    __syncthreads();
    if (w == 0 && i < len) {
      float t[W];
#pragma unroll
      for (int k = 0; k < W; ++k) t[k] = shf[k][c];
#pragma unroll
      for (int span = W / 2; span >= 1; span >>= 1)
#pragma unroll
        for (int k = 0; k < span; ++k) t[k] = t[2 * k] + t[2 * k + 1];
      g_dec[i] = t[0];
    }
    return;
  }
  if (threadIdx.x >= 256) return;   // 256 partial-row sums a block, as before
  const int j = ((int)blockIdx.x - dec_blocks) * 256 + threadIdx.x;
  if (j >= n_dp * 6) return;
  const int ray = j / 6, a = j % 6;
  double s = 0.0;
  for (int t = 0; t < nt; ++t) s += part[((size_t)ray * nt + t) * 6 + a];
  float* dst = a < 3 ? g_rays_o + ray * 3 + a : g_rays_d + ray * 3 + a - 3;
  *dst = (float)s;
}

// grad += sum of the coarse replicas (left zeroed), loss = sum of ray losses
__global__ __launch_bounds__(256) void nice_map_coarse_finish_kernel(
    float* __restrict__ rep, int64_t ne, float* __restrict__ grad,
    const double* __restrict__ ray_loss, int n, double* __restrict__ loss) {
  if (blockIdx.x == gridDim.x - 1) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += ray_loss[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && loss != nullptr)
      loss[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    return;
  }
  // a block = 256 consecutive elements (ne is a multiple of 32: channel-last
  // cells): thread (part, j) sums the float4 j of a quarter of the replicas —
  // 8 independent 16-byte loads instead of a chain of 32 scalar ones
  // (round 4: 10 us a launch for 5.4 MB) — the quarters meet in LDS
  static_assert(kCoarseRep % 4 == 0, "four quarters");
  __shared__ f32x4 quarter[4][64];
  const int j = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 256 + 4 * j;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < ne && rep != nullptr) {
    f32x4 v[kCoarseRep / 4];
#pragma unroll
    for (int r = 0; r < kCoarseRep / 4; ++r)
      v[r] = *reinterpret_cast<const f32x4*>(
          rep + (size_t)(part * (kCoarseRep / 4) + r) * ne + i);
#pragma unroll
    for (int r = 0; r < kCoarseRep / 4; ++r) {
      if (v[r][0] != 0.f || v[r][1] != 0.f || v[r][2] != 0.f ||
          v[r][3] != 0.f) {
        s += v[r];
        *reinterpret_cast<f32x4*>(
            rep + (size_t)(part * (kCoarseRep / 4) + r) * ne + i) =
            f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  quarter[part][j] = s;
  __syncthreads();
  if (part == 0 && i < ne && rep != nullptr) {
    const f32x4 t = (quarter[0][j] + quarter[1][j]) +
                    (quarter[2][j] + quarter[3][j]);
    if (t[0] != 0.f || t[1] != 0.f || t[2] != 0.f || t[3] != 0.f) {
      f32x4* g = reinterpret_cast<f32x4*>(grad + i);
      *g = *g + t;
    }
  }
}

constexpr int kMapBlocks = 256;  // persistent: one block per CU

template <int ST, bool DP, bool DW>
int launch_map(const xrd_nice_scene* scene, int n, const float* rays_o,
               const float* rays_d, const float* gt_depth, const float* dmax,
               const float* tgt_rgb, const uint8_t* keep, float w_color,
               float* const gg[4], double* part, float* dw_rep,
               float* dw_scr, double* ray_loss, float* exp_p, float* exp_g,
               hipStream_t st) {
  using G = MapGeom<3>;
  auto kern = nice_map_fused_kernel<ST, 3, DP, DW>;
  const size_t lds = G::LDS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute");
    attr_set = true;
  }
  if (n == 0) return XRD_OK;  // warm-up call: attributes only
  const int ngroups = (n + G::RPBM - 1) / G::RPBM;
  const int nb = ngroups < kMapBlocks ? ngroups : kMapBlocks;
  hipLaunchKernelGGL(kern, dim3(nb), dim3(G::NW * 64), lds, st, *scene, n,
                     rays_o, rays_d, gt_depth, dmax, tgt_rgb, keep, w_color,
                     gg[1], gg[2], gg[3], part, dw_rep, dw_scr, ray_loss,
                     TrackArgs{nullptr, nullptr, nullptr, 0, 0, exp_p, exp_g});
  return check_launch("xrd_nice_map_iter");
}

// tracking: ray gradients = sum of the ray's tile partials, loss = depth term
// + (float) w_color x (float) colour term like xrd_nice_loss; the grid barrier
// counter is left zeroed for the next call
__global__ __launch_bounds__(256) void nice_track_finish_kernel(
    const double* __restrict__ part, int n, float* __restrict__ g_rays_o,
    float* __restrict__ g_rays_d, const double* __restrict__ ray_loss,
    const double* __restrict__ ray_lossc, float w_color, unsigned* bar,
    double* __restrict__ loss) {
  __shared__ double sh[2][4];
  if (blockIdx.x == gridDim.x - 1) {
    double sd = 0.0, sc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
      sd += ray_loss[i];
      sc += ray_lossc[i];
    }
    sd = wave_sum(sd);
    sc = wave_sum(sc);
    if ((threadIdx.x & 63) == 0) {
      sh[0][threadIdx.x >> 6] = sd;
      sh[1][threadIdx.x >> 6] = sc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double d = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
      const double c = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
      if (loss != nullptr) loss[0] = d + (double)(w_color * (float)c);
      *bar = 0u;
    }
    return;
  }
  const int j = (int)blockIdx.x * 256 + threadIdx.x;
  if (j >= n * 6) return;
  const int ray = j / 6, a = j % 6;
  double s = 0.0;
  for (int t = 0; t < 3; ++t) s += part[((size_t)ray * 3 + t) * 6 + a];
  float* dst = a < 3 ? g_rays_o + ray * 3 + a : g_rays_d + ray * 3 + a - 3;
  *dst = (float)s;
}

#define MAP_ARGS                                                             \
  scene, n_rays, rays_o, rays_d, gt_depth, dmax, tgt_rgb, keep, w_color, gg, \
      part, dw_rep, dw_scr, ray_loss, exp_p, exp_g, st

int map_dispatch(int stage, bool dp, bool dw, const xrd_nice_scene* scene,
                 int n_rays, const float* rays_o, const float* rays_d,
                 const float* gt_depth, const float* dmax,
                 const float* tgt_rgb, const uint8_t* keep, float w_color,
                 float* const gg[4], double* part, float* dw_rep,
                 float* dw_scr, double* ray_loss, float* exp_p, float* exp_g,
                 hipStream_t st) {
  switch (stage) {
    case XRD_STAGE_MIDDLE:
      if (dp) return launch_map<XRD_STAGE_MIDDLE, true, false>(MAP_ARGS);
      return launch_map<XRD_STAGE_MIDDLE, false, false>(MAP_ARGS);
    case XRD_STAGE_FINE:
      if (dp) return launch_map<XRD_STAGE_FINE, true, false>(MAP_ARGS);
      return launch_map<XRD_STAGE_FINE, false, false>(MAP_ARGS);
    case XRD_STAGE_COLOR:
      if (dw) {
        if (dp) return launch_map<XRD_STAGE_COLOR, true, true>(MAP_ARGS);
        return launch_map<XRD_STAGE_COLOR, false, true>(MAP_ARGS);
      }
      if (dp) return launch_map<XRD_STAGE_COLOR, true, false>(MAP_ARGS);
      return launch_map<XRD_STAGE_COLOR, false, false>(MAP_ARGS);
  }
  return XRD_ERR_ARG;
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int64_t xrd_nice_map_ws_floats(const xrd_nice_scene* scene, int stage,
                               int n_rays) {
  if (scene == nullptr || n_rays < 0) return -1;
  if (stage == XRD_STAGE_COARSE)
    return 2 * (int64_t)n_rays + 4 +
           (int64_t)kCoarseRep * scene->gdim[0] * scene->gdim[1] *
               scene->gdim[2] * 32;
  // [n*3][6] f64 tile partials | [n] f64 ray losses | the blocks' dW rows | (colour
  // stage) the contraction's operand scratch, one tile slot per wave and group
  int64_t need = (int64_t)n_rays * 3 * 6 * 2 + 2 * (int64_t)n_rays + 4 +
                 (int64_t)kMapBlocks * kColorFlat;
  need = (need + 3) / 4 * 4;
  if (stage == XRD_STAGE_COLOR)
    need += ((int64_t)n_rays + 3) / 4 * 12 * SC_TILE;
  return need;
}

int xrd_nice_map_iter(const xrd_nice_scene* scene, int stage, int n_rays,
                      const float* rays_o, const float* rays_d,
                      const float* gt_depth, const float* dmax,
                      const float* tgt_rgb, const uint8_t* keep,
                      float w_color, float* g_rays_o, float* g_rays_d,
                      float* const g_grid[4], float* g_dec_color, float* ws,
                      double* loss, xrd_stream_t stream) {
  return xrd_nice_map_iter_export(scene, stage, n_rays, rays_o, rays_d,
                                  gt_depth, dmax, tgt_rgb, keep, w_color,
                                  g_rays_o, g_rays_d, g_grid, g_dec_color,
                                  nullptr, nullptr, ws, loss, stream);
}

int xrd_nice_map_iter_export(const xrd_nice_scene* scene, int stage,
                             int n_rays, const float* rays_o,
                             const float* rays_d, const float* gt_depth,
                             const float* dmax, const float* tgt_rgb,
                             const uint8_t* keep, float w_color,
                             float* g_rays_o, float* g_rays_d,
                             float* const g_grid[4], float* g_dec_color,
                             float* sample_points, float* g_occ, float* ws,
                             double* loss, xrd_stream_t stream) {
  if (scene == nullptr || n_rays < 0 || stage < 0 || stage > 3)
    return XRD_ERR_ARG;
  if ((sample_points == nullptr) != (g_occ == nullptr)) return XRD_ERR_ARG;
  if (g_occ != nullptr && stage == XRD_STAGE_COARSE)
    return XRD_ERR_UNSUPPORTED;
  float* exp_p = sample_points;
  float* exp_g = g_occ;
  if (!rays_o || !rays_d || !gt_depth || !ws) return XRD_ERR_ARG;
  if ((g_rays_o == nullptr) != (g_rays_d == nullptr)) return XRD_ERR_ARG;
  if (scene->t_uniform == nullptr) return XRD_ERR_ARG;
  const int need[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 1, 1, 0}, {0, 1, 1, 1}};
  for (int g = 0; g < 4; ++g)
    if (need[stage][g] && (scene->grid[g] == nullptr ||
                           scene->dec[g] == nullptr))
      return XRD_ERR_ARG;
  float* gg[4] = {nullptr, nullptr, nullptr, nullptr};
  if (g_grid)
    for (int g = 0; g < 4; ++g) gg[g] = g_grid[g];
  hipStream_t st = (hipStream_t)stream;
  if (stage == XRD_STAGE_COARSE) {
    if (scene->n_samples != 32) return XRD_ERR_UNSUPPORTED;
    if (g_rays_o != nullptr || g_dec_color != nullptr)
      return XRD_ERR_UNSUPPORTED;  // the coarse stage never reaches them
    if (n_rays == 0) return XRD_OK;
    double* ray_loss = reinterpret_cast<double*>(ws);
    // (16-byte aligned behind the n f64 ray losses, inside the 2n + 4 floats
    // xrd_nice_map_ws_floats reserves in front of the replicas)
    float* rep = ws + (2 * (size_t)n_rays + 3) / 4 * 4;
    const int64_t ne =
        (int64_t)scene->gdim[0] * scene->gdim[1] * scene->gdim[2] * 32;
    static bool coarse_attr = false;
    if (!coarse_attr) {
      if (hipFuncSetAttribute(
              reinterpret_cast<const void*>(nice_map_coarse_kernel),
              hipFuncAttributeMaxDynamicSharedMemorySize,
              (int)kCoarseLds) != hipSuccess)
        return check_launch("hipFuncSetAttribute");
      coarse_attr = true;
    }
    const int cgroups = (n_rays + kCoarseRPB - 1) / kCoarseRPB;
    hipLaunchKernelGGL(nice_map_coarse_kernel,
                       dim3(cgroups < 2 * kMapBlocks ? cgroups : 2 * kMapBlocks),
                       dim3(kCoarseWaves * 64), kCoarseLds, st, *scene, n_rays,
                       rays_o, rays_d, gt_depth, keep, gg[0],
                       gg[0] ? rep : nullptr, ray_loss);
    int rc = check_launch("xrd_nice_map_iter/coarse");
    if (rc != XRD_OK) return rc;
    const int64_t nb = (gg[0] ? (ne + 255) / 256 : 0) + 1;
    hipLaunchKernelGGL(nice_map_coarse_finish_kernel, dim3((unsigned)nb),
                       dim3(256), 0, st, gg[0] ? rep : nullptr, ne, gg[0],
                       ray_loss, n_rays, loss);
    return check_launch("xrd_nice_map_iter/coarse_finish");
  }
  if (scene->n_samples != 32 || scene->n_surface != 16)
    return XRD_ERR_UNSUPPORTED;  // 48 samples a ray = 3 tiles
  if (!dmax || scene->t_surface == nullptr) return XRD_ERR_ARG;
  const bool dw = g_dec_color != nullptr;
  if (dw && stage != XRD_STAGE_COLOR) return XRD_ERR_ARG;
  if (stage == XRD_STAGE_COLOR && !tgt_rgb) return XRD_ERR_ARG;
  const bool dp = g_rays_o != nullptr;
  if (n_rays == 0) return XRD_OK;
  double* part = reinterpret_cast<double*>(ws);
  double* ray_loss = part + (size_t)n_rays * 3 * 6;
  const size_t rep_off = (size_t)n_rays * 3 * 6 * 2 + 2 * (size_t)n_rays + 4;
  float* dw_rep = ws + rep_off;
  float* dw_scr =
      ws + (rep_off + (size_t)kMapBlocks * kColorFlat + 3) / 4 * 4;
  int rc = map_dispatch(stage, dp, dw, scene, n_rays, rays_o, rays_d, gt_depth,
                        dmax, tgt_rgb, keep, w_color, gg, part, dw_rep, dw_scr,
                        ray_loss, exp_p, exp_g, st);
  if (rc != XRD_OK) return rc;
  const int len = dw ? kColorFlat : 0;
  const int ngroups_map = (n_rays + 3) / 4;   // MapGeom::RPBM rays a group
  const int nb_map = ngroups_map < kMapBlocks ? ngroups_map : kMapBlocks;
  const int dec_blocks = (len + 63) / 64;
  const int ray_blocks = dp ? (n_rays * 6 + 255) / 256 : 0;
  hipLaunchKernelGGL(nice_map_finish_kernel,
                     dim3(dec_blocks + ray_blocks + 1), dim3(kFinishThreads),
                     0, st,
                     dw_rep, nb_map, len, dec_blocks, g_dec_color, part,
                     dp ? n_rays : 0, 3, g_rays_o, g_rays_d, ray_loss, n_rays,
                     loss);
  return check_launch("xrd_nice_map_iter/finish");
}

int64_t xrd_nice_track_ws_floats(int n_rays) {
  if (n_rays < 0) return -1;
  // [n*3][6] f64 tile partials | [n] f64 depth terms | [n] f64 colour terms |
  // [n] f64 residuals | barrier counter
  return (int64_t)n_rays * 3 * 6 * 2 + 6 * (int64_t)n_rays + 4;
}

int xrd_nice_track_iter(const xrd_nice_scene* scene, int n_rays,
                        const float* rays_o, const float* rays_d,
                        const float* gt_depth, const float* dmax,
                        const float* tgt_rgb, const uint8_t* keep,
                        int use_color, int handle_dynamic, float w_color,
                        float* g_rays_o, float* g_rays_d, float* ws,
                        double* loss, xrd_stream_t stream) {
  using G = MapGeom<3>;
  if (scene == nullptr || n_rays < 0) return XRD_ERR_ARG;
  if (!rays_o || !rays_d || !gt_depth || !dmax || !tgt_rgb || !g_rays_o ||
      !g_rays_d || !ws)
    return XRD_ERR_ARG;
  if (scene->t_uniform == nullptr || scene->t_surface == nullptr)
    return XRD_ERR_ARG;
  for (int g = 1; g < 4; ++g)
    if (scene->grid[g] == nullptr || scene->dec[g] == nullptr)
      return XRD_ERR_ARG;
  if (scene->n_samples != 32 || scene->n_surface != 16)
    return XRD_ERR_UNSUPPORTED;  // 48 samples a ray = 3 tiles
  // the grid barrier needs every block resident: one block (4 rays) a CU
  if (n_rays > G::RPBM * kMapBlocks) return XRD_ERR_UNSUPPORTED;
  if (n_rays == 0) return XRD_OK;
  hipStream_t st = (hipStream_t)stream;
  auto kern = nice_map_fused_kernel<XRD_STAGE_COLOR, 3, true, false, true>;
  const size_t lds = G::LDS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute");
    attr_set = true;
  }
  double* part = reinterpret_cast<double*>(ws);
  double* ray_loss = part + (size_t)n_rays * 3 * 6;
  TrackArgs trk = {};
  trk.ray_lossc = ray_loss + n_rays;
  trk.res = trk.ray_lossc + n_rays;
  trk.bar = reinterpret_cast<unsigned*>(trk.res + n_rays);
  trk.use_color = use_color;
  trk.handle_dynamic = handle_dynamic;
  const int ngroups = (n_rays + G::RPBM - 1) / G::RPBM;
  hipLaunchKernelGGL(kern, dim3(ngroups), dim3(G::NW * 64), lds, st, *scene,
                     n_rays, rays_o, rays_d, gt_depth, dmax, tgt_rgb, keep,
                     w_color, (float*)nullptr, (float*)nullptr,
                     (float*)nullptr, part, (float*)nullptr, (float*)nullptr,
                     ray_loss, trk);
  int rc = check_launch("xrd_nice_track_iter");
  if (rc != XRD_OK) return rc;
  hipLaunchKernelGGL(nice_track_finish_kernel,
                     dim3((n_rays * 6 + 255) / 256 + 1), dim3(256), 0, st,
                     part, n_rays, g_rays_o, g_rays_d, ray_loss,
                     trk.ray_lossc, w_color, trk.bar, loss);
  return check_launch("xrd_nice_track_iter/finish");
}

int xrd_nice_map_warmup(void) {
  float* gg[4] = {nullptr, nullptr, nullptr, nullptr};
  xrd_nice_scene sc = {};
  for (int stage = XRD_STAGE_MIDDLE; stage <= XRD_STAGE_COLOR; ++stage)
    for (int dp = 0; dp < 2; ++dp)
      for (int dw = 0; dw < 2; ++dw) {
        if (dw && stage != XRD_STAGE_COLOR) continue;
        int rc = map_dispatch(stage, dp, dw, &sc, 0, nullptr, nullptr, nullptr,
                              nullptr, nullptr, nullptr, 0.f, gg, nullptr,
                              nullptr, nullptr, nullptr, nullptr, nullptr,
                              nullptr);
        if (rc != XRD_OK) return rc;
      }
  return XRD_OK;
}

}  // extern "C"
