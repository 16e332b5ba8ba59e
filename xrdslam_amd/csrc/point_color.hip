// Point-SLAM colour path on gfx950: per-neighbour F_theta, inverse-distance
// interpolation and the 5 x 128 softplus colour decoder as one kernel each way
// (MLP_color.get_feature_at_pos / forward, MLP_col_neighbor,
// slam/model_components/decoder_pointslam.py:276-291,408-542; layouts:
// point_layout.h).
//
// One wave = 16 sample points, activations stay in registers in D layout
// (rows = features, columns = the 16 points; the accumulators of a layer are
// the B operand of the next).  A block = 4 or 8 waves (pc_waves_*); it stages
// one layer's fragments in LDS at a time (<= 103 KB forward, 123 KB backward)
// and loops over groups of 16 points per wave.
// Per point the 8 neighbours run through F_theta one after the other
// (13 + 32 K-steps x 8 / 2 output tiles), their outputs are combined with the
// interpolation weights, then the trunk follows (10/32/32/42/32 K-steps x 8
// tiles + the feature injection, 8 K-steps x 8 tiles, per layer).
//
// The backward recomputes F_theta per neighbour, reads the trunk's layer
// outputs back from HBM, returns d loss / d positions (Fourier features of p,
// relative-position features, neighbour distances) and scatters the colour
// feature gradients with atomics (tiles transposed through LDS: 128-byte rows
// per instruction).  Two kernels: point_color_bwd_kernel (tracking: no
// parameter gradients) and point_color_bwd_w_kernel (mapping: the weight
// gradients contracted inside the block, see there).
//
// Reference behaviour restated, never copied; parity: tests/test_pointslam_hip.py.
#include <hip/hip_runtime.h>

#include "common.h"
#include "point_common.h"
#include "point_layout.h"

namespace xrd {
namespace {

constexpr int kPcBlocks = 256;    // persistent blocks
constexpr float kBeta = 100.f;
constexpr int kTailLen = PcPack::FWD_LEN - PcPack::OW + 32;
constexpr int kPcLds = (PcPack::STAGE_MAX + kTailLen) * (int)sizeof(float);
constexpr float kTwoPi = 6.283185307179586f;

// global -> LDS copy of n floats (n % 4 == 0) by the whole block: the loads of
// a batch are all in flight before the first store (a load-wait-store loop is
// latency-bound with 128..512 threads and ~100 KB per stage)
__device__ __forceinline__ void pc_copy(float* __restrict__ wl,
                                        const float* __restrict__ src, int n) {
  constexpr int U = 8;
  const int step = blockDim.x * 4;
  int i = threadIdx.x * 4;
  for (; i + (U - 1) * step < n; i += U * step) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = *reinterpret_cast<const f32x4*>(src + i + u * step);
#pragma unroll
    for (int u = 0; u < U; ++u)
      *reinterpret_cast<f32x4*>(wl + i + u * step) = v[u];
  }
  for (; i < n; i += step)
    *reinterpret_cast<f32x4*>(wl + i) =
        *reinterpret_cast<const f32x4*>(src + i);
}

__device__ __forceinline__ void pc_stage(float* __restrict__ wl,
                                         const float* __restrict__ src,
                                         int n) {
  __syncthreads();
  pc_copy(wl, src, n);
  __syncthreads();
}

// element k of a register array without dynamic indexing (no scratch)
template <class T>
__device__ __forceinline__ T pick8(const T (&v)[8], int k) {
  T r = v[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) r = k == j ? v[j] : r;
  return r;
}

// torch.nn.Softplus(beta=100): x if beta x > 20 else log1p(exp(beta x)) / beta,
// evaluated in the stable form max(x, 0) + log1p(exp(-|beta x|)) / beta on the
// hardware exp2 / log2 (1 ulp each; the two forms differ by < 1e-8 absolute,
// the library expf / log1pf pair costs ~10x the instructions and made the
// kernel VALU-bound)
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
__device__ __forceinline__ float softplus100(float x) {
  const float t = __builtin_amdgcn_exp2f(-fabsf(kBeta * x) * kLog2e);
  return fmaf(__builtin_amdgcn_logf(1.f + t), kLn2 / kBeta, fmaxf(x, 0.f));
}
// d softplus / dx = sigmoid(beta x)
__device__ __forceinline__ float softplus100_grad(float x) {
  return __builtin_amdgcn_rcpf(
      1.f + __builtin_amdgcn_exp2f(-kBeta * kLog2e * x));
}
// the same from the VALUE y = softplus(x): 1 - exp(-beta y)
__device__ __forceinline__ float softplus100_grad_of_value(float y) {
  return 1.f - __builtin_amdgcn_exp2f(-kBeta * kLog2e * fmaxf(y, 0.f));
}

__device__ __forceinline__ f32x4 softplus4(const f32x4 a) {
  return f32x4{softplus100(a[0]), softplus100(a[1]), softplus100(a[2]),
               softplus100(a[3])};
}

// acc[jt] += frag(jt, s0 + s) * in(s) for s < KS, JT output tiles, fragments
// laid out (jt * KTOT + s); in: D-layout registers (dense_h) or one float per
// K-step (dense_e).
// A stage spans up to 123 KB of LDS while a ds_read reaches 64 KB beyond its
// address register; left alone the compiler materialises one address VGPR per
// read of the far part, runs out of registers and issues read - wait - MFMA
// one at a time.  Two opaque lane offsets (tiles 0..3 / 4..7) keep every read
// an immediate-offset read.
extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];

template <int JT, int KTOT>
struct FragBase {
  int lo, hi;
  __device__ __forceinline__ FragBase(const float* w, int lane) {
    lo = (int)(w - reinterpret_cast<const float*>(pc_smem)) + lane;
    hi = lo + (JT > 4 ? 4 * KTOT * 64 : 0);
    asm volatile("" : "+v"(lo), "+v"(hi));
  }
  __device__ __forceinline__ float operator()(int jt, int s) const {
    const float* base = reinterpret_cast<const float*>(pc_smem);
    return JT > 4 && jt >= 4 ? base[hi + ((jt - 4) * KTOT + s) * 64]
                             : base[lo + (jt * KTOT + s) * 64];
  }
};

template <int JT, int KTOT, int KS>
__device__ __forceinline__ void dense_h(const float* __restrict__ w, int lane,
                                        int s0, const f32x4* in, f32x4* acc) {
  const FragBase<JT, KTOT> frag(w, lane);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float v = in[s >> 2][s & 3];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
      acc[jt] = XRD_MFMA4(frag(jt, s0 + s), v, acc[jt]);
  }
}
template <int JT, int KTOT, int KS>
__device__ __forceinline__ void dense_e(const float* __restrict__ w, int lane,
                                        int s0, const float* in, f32x4* acc) {
  const FragBase<JT, KTOT> frag(w, lane);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float v = in[s];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
      acc[jt] = XRD_MFMA4(frag(jt, s0 + s), v, acc[jt]);
  }
}

__device__ __forceinline__ f32x4 bias4(const float* __restrict__ b, int jt,
                                       int q) {
  return *reinterpret_cast<const f32x4*>(b + 16 * jt + 4 * q);
}

// [sin, cos]((2 pi x) . B) features of an N-column embedding matrix stored
// [N][4]; x arrives multiplied by 2 pi (the reference's order of operations):
// feature f < N: sin(x . B_f), f >= N: cos(x . B_{f-N})
template <int N>
__device__ __forceinline__ float emb_feature(const float* __restrict__ B4,
                                             const float (&x)[3], int f) {
  const f32x4 b = *reinterpret_cast<const f32x4*>(B4 + (f < N ? f : f - N) * 4);
  float a = x[0] * b[0];
  a = fmaf(x[1], b[1], a);
  a = fmaf(x[2], b[2], a);
  return f < N ? sin_cw(a) : cos_cw(a);
}

// F_theta of one neighbour: y (D layout, 2 tiles) from rel = c_k - p and the
// neighbour's colour feature row; h (8 tiles) is returned for the backward
__device__ __forceinline__ void ftheta_fwd(const float* __restrict__ wl,
                                           int lane, int q,
                                           const float (&rel)[3],
                                           const f32x4 (&f)[2], f32x4 (&h)[8],
                                           f32x4 (&y)[2]) {
  using K = PcPack;
  float e[5];
#pragma unroll
  for (int s = 0; s < 5; ++s)
    e[s] = emb_feature<10>(wl + (K::BREL - K::FT), rel, 4 * s + q);
#pragma unroll
  for (int jt = 0; jt < 8; ++jt) h[jt] = bias4(wl + (K::B1 - K::FT), jt, q);
  dense_e<8, 13, 5>(wl + (K::W1 - K::FT), lane, 0, e, h);
  dense_h<8, 13, 8>(wl + (K::W1 - K::FT), lane, 5, f, h);
#pragma unroll
  for (int jt = 0; jt < 8; ++jt) h[jt] = softplus4(h[jt]);
  y[0] = bias4(wl + (K::B2 - K::FT), 0, q);
  y[1] = bias4(wl + (K::B2 - K::FT), 1, q);
  dense_h<2, 32, 32>(wl + (K::W2 - K::FT), lane, 0, h, y);
}

template <int TILES>
__device__ __forceinline__ void save_rows(float* __restrict__ dst, int width,
                                          int64_t pt, int q, const f32x4* v) {
#pragma unroll
  for (int jt = 0; jt < TILES; ++jt)
    *reinterpret_cast<f32x4*>(dst + pt * width + 16 * jt + 4 * q) = v[jt];
}

template <int PW>
__global__ __launch_bounds__(PW * 64, 2) void point_color_fwd_kernel(
    int64_t n, const float* __restrict__ pts, const int64_t* __restrict__ nbr,
    const int* __restrict__ n_nb, const float* __restrict__ cloud,
    const float* __restrict__ feats, const float* __restrict__ radius,
    float radius_all, int min_nn, const float* __restrict__ empty,
    const float* pk, float* __restrict__ rgb,
    float* __restrict__ save_c, float* __restrict__ save_h,
    float* __restrict__ save_y) {
  using K = PcPack;
  float* wl = reinterpret_cast<float*>(pc_smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // small operands that every group reads: output layer, embedding matrix,
  // the call's empty feature (LDS tail, staged once)
  float* tail = wl + K::STAGE_MAX;
  for (int i = threadIdx.x; i < kTailLen; i += blockDim.x)
    tail[i] = i < K::FWD_LEN - K::OW ? pk[K::OW + i]
                                     : empty[i - (K::FWD_LEN - K::OW)];
  const float* ow = tail;
  const float* ob = tail + (K::OB - K::OW);
  const float* bemb = tail + (K::BEMB - K::OW);
  const float* emp = tail + (K::FWD_LEN - K::OW);
  const int64_t ngroups = (n + 16 * PW - 1) / (16 * PW);
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t pt = (grp * PW + wave) * 16 + li;
    const bool valid = pt < n;
    // keep the staging addresses inside the loop (hoisted, they cost ~100 VGPRs)
    asm volatile("" : "+s"(pk));
    float p[3] = {0.f, 0.f, 0.f};
    if (valid) {
#pragma unroll
      for (int a = 0; a < 3; ++a) p[a] = pts[pt * 3 + a];
    }
    PointNb nb;
    point_neighbors(nbr, cloud, n_nb, radius, radius_all, min_nn, pt, valid, p,
                    nb);
    // ---- F_theta over the neighbours, interpolation ----------------------------
    pc_stage(wl, pk + K::FT, K::FT_LEN);
    f32x4 c[2] = {z4, z4};
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      // the fragments are re-read from LDS per neighbour (kept in registers
      // across the loop they would not fit)
      asm volatile("" ::: "memory");
      const float uk = pick8(nb.u, k);
      const bool live = nb.has && uk != 0.f;
      // (uniform work: the MFMA chain runs for every neighbour slot; slots
      // without weight contribute nothing)
      float rel[3] = {0.f, 0.f, 0.f};
      f32x4 f[2] = {z4, z4};
      if (live) {
        const int64_t id = pick8(nb.id, k);
#pragma unroll
        for (int a = 0; a < 3; ++a)
          rel[a] = kTwoPi * (cloud[id * 3 + a] - p[a]);
        f[0] = *reinterpret_cast<const f32x4*>(feats + id * 32 + 4 * q);
        f[1] = *reinterpret_cast<const f32x4*>(feats + id * 32 + 16 + 4 * q);
      }
      f32x4 h[8], y[2];
      ftheta_fwd(wl, lane, q, rel, f, h, y);
      const float w = live ? uk / nb.den : 0.f;
      c[0] += y[0] * w;
      c[1] += y[1] * w;
      if (valid && save_y) save_rows<2>(save_y, 32, pt * 8 + k, q, y);
    }
    if (!nb.has) {
      c[0] = *reinterpret_cast<const f32x4*>(emp + 4 * q);
      c[1] = *reinterpret_cast<const f32x4*>(emp + 16 + 4 * q);
    }
    if (valid && save_c) save_rows<2>(save_c, 32, pt, q, c);
    // ---- trunk --------------------------------------------------------------------
    float e[10];
    {
      const float p2[3] = {kTwoPi * p[0], kTwoPi * p[1], kTwoPi * p[2]};
#pragma unroll
      for (int s = 0; s < 10; ++s)
        e[s] = emb_feature<20>(bemb, p2, 4 * s + q);
    }
    f32x4 h[8];
#pragma unroll 1
    for (int i = 0; i < 5; ++i) {
      pc_stage(wl, pk + K::tw(i), K::tlen(i));
      const float* W = wl;
      const float* Bv = wl + (K::tb(i) - K::tw(i));
      const float* FCw = wl + (K::tfc(i) - K::tw(i));
      const float* FCb = wl + (K::tfb(i) - K::tw(i));
      f32x4 acc[8], cc[8];
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) {
        acc[jt] = bias4(Bv, jt, q);
        cc[jt] = bias4(FCb, jt, q);
      }
      if (i == 0) {
        dense_e<8, 10, 10>(W, lane, 0, e, acc);
      } else if (i == 3) {
        dense_e<8, 42, 10>(W, lane, 0, e, acc);
        dense_h<8, 42, 32>(W, lane, 10, h, acc);
      } else {
        dense_h<8, 32, 32>(W, lane, 0, h, acc);
      }
      dense_h<8, 8, 8>(FCw, lane, 0, c, cc);
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) h[jt] = softplus4(acc[jt]) + cc[jt];
      if (valid && save_h)
        save_rows<8>(save_h + (int64_t)i * n * 128, 128, pt, q, h);
    }
    // ---- output layer (3 rows, VALU) + sigmoid ----------------------------------
    float o[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float v = 0.f;
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(ow + r * 128 +
                                                        16 * jt + 4 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) v += w[t] * h[jt][t];
      }
      o[r] = group4_sum(v) + ob[r];
    }
    if (valid && q == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) rgb[pt * 3 + r] = 1.f / (1.f + expf(-o[r]));
    }
  }
}

// ---- backward -------------------------------------------------------------------
constexpr int kBwdFcOff = PcPack::rlen(3);               // forward FC frag + bias
constexpr int kBwdTail = kBwdFcOff + 8 * 8 * 64 + 128;   // OW, OB, BEMB
constexpr int kBwdTailLen = PcPack::FWD_LEN - PcPack::OW;
constexpr int kTileLen = 16 * 33;                        // scatter tile per wave
constexpr int kBwdTile = (kBwdTail + kBwdTailLen + 3) / 4 * 4;
constexpr int kBwdLds = (kBwdTile + 8 * kTileLen) * (int)sizeof(float);
static_assert(kBwdLds <= 160 * 1024, "backward LDS");
constexpr int kBwdRfOff = PcPack::FT_LEN;
static_assert(kBwdRfOff + PcPack::RF_LEN <= kBwdTail, "F_theta stage fits");
static_assert(PcPack::rlen(3) >= PcPack::rlen(4) &&
              PcPack::rlen(3) >= PcPack::rlen(0), "largest backward stage");

template <int TILES>
__device__ __forceinline__ void load_rows(const float* __restrict__ src,
                                          int width, int64_t row, int q,
                                          bool valid, f32x4* v) {
#pragma unroll
  for (int jt = 0; jt < TILES; ++jt)
    v[jt] = valid ? *reinterpret_cast<const f32x4*>(src + row * width +
                                                    16 * jt + 4 * q)
                  : f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int PW>
__global__ __launch_bounds__(PW * 64, 2) void point_color_bwd_kernel(
    int64_t n, const float* __restrict__ pts, const int64_t* __restrict__ nbr,
    const int* __restrict__ n_nb, const float* __restrict__ cloud,
    const float* __restrict__ feats, const float* __restrict__ radius,
    float radius_all, int min_nn, const float* pk,
    const float* __restrict__ rgb, const float* __restrict__ save_c,
    const float* __restrict__ save_h, const float* __restrict__ save_y,
    const float* __restrict__ g_rgb, float* __restrict__ g_pts,
    float* __restrict__ g_feats) {
  using K = PcPack;
  float* wl = reinterpret_cast<float*>(pc_smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  float* tail = wl + kBwdTail;
  for (int i = threadIdx.x; i < kBwdTailLen; i += blockDim.x)
    tail[i] = pk[K::OW + i];
  const float* ow = tail;
  const float* bemb = tail + (K::BEMB - K::OW);
  const int64_t ngroups = (n + 16 * PW - 1) / (16 * PW);
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t pt = (grp * PW + wave) * 16 + li;
    const bool valid = pt < n;
    asm volatile("" : "+s"(pk));
    float p[3] = {0.f, 0.f, 0.f};
    if (valid) {
#pragma unroll
      for (int a = 0; a < 3; ++a) p[a] = pts[pt * 3 + a];
    }
    PointNb nb;
    point_neighbors(nbr, cloud, n_nb, radius, radius_all, min_nn, pt, valid, p,
                    nb);
    // ---- sigmoid, output layer ----------------------------------------------------
    float go[3] = {0.f, 0.f, 0.f};
    if (valid) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float o = rgb[pt * 3 + r];
        go[r] = g_rgb[pt * 3 + r] * o * (1.f - o);
      }
    }
    f32x4 g_h[8];
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
      g_h[jt] = z4;
#pragma unroll
      for (int r = 0; r < 3; ++r)
        g_h[jt] += *reinterpret_cast<const f32x4*>(ow + r * 128 + 16 * jt +
                                                   4 * q) * go[r];
    }
    f32x4 c[2];
    load_rows<2>(save_c, 32, pt, q, valid, c);
    // embedding of p: lane group q holds sin and cos of columns 4s + q, s < 5
    float e[10];
    {
      const float p2[3] = {kTwoPi * p[0], kTwoPi * p[1], kTwoPi * p[2]};
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bemb + (4 * s + q) * 4);
        float a = p2[0] * b[0];
        a = fmaf(p2[1], b[1], a);
        a = fmaf(p2[2], b[2], a);
        sincos_cw(a, e[s], e[s + 5]);
      }
    }
    f32x4 g_c[2] = {z4, z4}, g_e[3] = {z4, z4, z4};
    // ---- trunk, last layer first ---------------------------------------------------
#pragma unroll 1
    for (int i = 4; i >= 0; --i) {
      __syncthreads();
      pc_copy(wl, pk + K::rw(i), K::rlen(i));
      pc_copy(wl + kBwdFcOff, pk + K::tfc(i), 8 * 8 * 64 + 128);
      __syncthreads();
      const float* WT = wl;
      const float* ET = wl + (K::ret(i) - K::rw(i));
      const float* FCT = wl + (K::rfc(i) - K::rw(i));
      const float* FCw = wl + kBwdFcOff;
      const float* FCb = FCw + 8 * 8 * 64;
      dense_h<2, 32, 32>(FCT, lane, 0, g_h, g_c);
      {
        // softplus'(a) = 1 - exp(-beta softplus(a)), softplus(a) = h - FC c
        f32x4 h[8], cc[8];
        load_rows<8>(save_h + (int64_t)i * n * 128, 128, pt, q, valid, h);
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) cc[jt] = bias4(FCb, jt, q);
        dense_h<8, 8, 8>(FCw, lane, 0, c, cc);
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            g_h[jt][t] *= softplus100_grad_of_value(h[jt][t] - cc[jt][t]);
          }
      }
      if (i == 0 || i == 3) dense_h<3, 32, 32>(ET, lane, 0, g_h, g_e);
      if (i >= 1) {
        f32x4 gp[8];
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) gp[jt] = z4;
        dense_h<8, 32, 32>(WT, lane, 0, g_h, gp);
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) g_h[jt] = gp[jt];
      }
    }
    // ---- d / d p through the embedding -------------------------------------------
    float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const float garg = g_e[s >> 2][s & 3] * e[s + 5] -
                         g_e[(s + 5) >> 2][(s + 5) & 3] * e[s];
      const f32x4 b = *reinterpret_cast<const f32x4*>(bemb + (4 * s + q) * 4);
#pragma unroll
      for (int a = 0; a < 3; ++a) gp[a] = fmaf(garg, b[a], gp[a]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) gp[a] = group4_sum(gp[a]) * kTwoPi;
    // ---- interpolation weights ----------------------------------------------------
    __syncthreads();
    pc_copy(wl, pk + K::FT, K::FT_LEN);
    pc_copy(wl + kBwdRfOff, pk + K::RF, K::RF_LEN);
    __syncthreads();
    if (!nb.has) g_c[0] = g_c[1] = z4;
    float gD[8];
    {
      float gw[8], sum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        f32x4 y[2];
        load_rows<2>(save_y, 32, pt * 8 + k, q, valid, y);
        float d = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          d += g_c[0][t] * y[0][t] + g_c[1][t] * y[1][t];
        gw[k] = group4_sum(d);
        sum += gw[k] * (nb.u[k] / nb.den);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        gD[k] = (nb.has && nb.u[k] != 0.f)
                    ? -(nb.u[k] * nb.u[k]) * ((gw[k] - sum) / nb.den)
                    : 0.f;
    }
    // ---- F_theta, one neighbour at a time ---------------------------------------
    const float* W1 = wl + (K::W1 - K::FT);
    const float* brel = wl + (K::BREL - K::FT);
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      asm volatile("" ::: "memory");
      const float uk = pick8(nb.u, k);
      const bool live = nb.has && uk != 0.f;
      const int64_t id = pick8(nb.id, k);
      float raw[3] = {0.f, 0.f, 0.f}, rel[3];
      f32x4 f[2] = {z4, z4};
      if (live) {
#pragma unroll
        for (int a = 0; a < 3; ++a) raw[a] = cloud[id * 3 + a] - p[a];
        f[0] = *reinterpret_cast<const f32x4*>(feats + id * 32 + 4 * q);
        f[1] = *reinterpret_cast<const f32x4*>(feats + id * 32 + 16 + 4 * q);
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) rel[a] = kTwoPi * raw[a];
      // relative-position features of this lane group and their derivatives
      float e5[5], d5[5];
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int j = 4 * s + q, fi = j < 10 ? j : j - 10;
        const f32x4 b = *reinterpret_cast<const f32x4*>(brel + fi * 4);
        float a = rel[0] * b[0];
        a = fmaf(rel[1], b[1], a);
        a = fmaf(rel[2], b[2], a);
        float sn, cs;
        sincos_cw(a, sn, cs);
        e5[s] = j < 10 ? sn : cs;
        d5[s] = j < 10 ? cs : -sn;
      }
      f32x4 sp[8];
      {
        f32x4 a[8];
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) a[jt] = bias4(wl + (K::B1 - K::FT), jt, q);
        dense_e<8, 13, 5>(W1, lane, 0, e5, a);
        dense_h<8, 13, 8>(W1, lane, 5, f, a);
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
          for (int t = 0; t < 4; ++t) sp[jt][t] = softplus100_grad(a[jt][t]);
      }
      const float w = live ? uk / nb.den : 0.f;
      f32x4 g_y[2] = {g_c[0] * w, g_c[1] * w};
      f32x4 g_a[8];
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) g_a[jt] = z4;
      dense_h<8, 8, 8>(wl + kBwdRfOff + (K::W2T - K::RF), lane, 0, g_y, g_a);
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) g_a[jt] *= sp[jt];
      if (g_feats != nullptr) {
        f32x4 g_f[2] = {z4, z4};
        dense_h<2, 32, 32>(wl + kBwdRfOff + (K::W1TF - K::RF), lane, 0, g_a,
                           g_f);
        // through LDS: 32 consecutive lanes add the 32 features of one
        // neighbour (two 128-byte rows per atomic instruction)
        float* T = wl + kBwdTile + wave * kTileLen;   // [16][33]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          T[li * 33 + 4 * q + t] = g_f[0][t];
          T[li * 33 + 16 + 4 * q + t] = g_f[1][t];
        }
        wave_lds_sync();
        const int idl = live ? (int)id : -1;
        const int ff = lane & 31, half = lane >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int pt2 = 2 * i + half;   // lane pt2 = (q 0, li pt2)
          const int id2 = __shfl(idl, pt2);
          if (id2 >= 0)
            atomicAdd(g_feats + (int64_t)id2 * 32 + ff, T[pt2 * 33 + ff]);
        }
        wave_lds_sync();
      }
      f32x4 g_er[2] = {z4, z4};
      dense_h<2, 32, 32>(wl + kBwdRfOff + (K::W1TE - K::RF), lane, 0, g_a,
                         g_er);
      float grel[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int j = 4 * s + q, fi = j < 10 ? j : j - 10;
        const float garg = g_er[s >> 2][s & 3] * d5[s];
        const f32x4 b = *reinterpret_cast<const f32x4*>(brel + fi * 4);
#pragma unroll
        for (int a = 0; a < 3; ++a) grel[a] = fmaf(garg, b[a], grel[a]);
      }
      const float gDk = pick8(gD, k);
#pragma unroll
      for (int a = 0; a < 3; ++a)
        gp[a] -= group4_sum(grel[a]) * kTwoPi + 2.f * raw[a] * gDk;
    }
    if (valid && q == 0 && g_pts != nullptr) {
#pragma unroll
      for (int a = 0; a < 3; ++a) g_pts[pt * 3 + a] = gp[a];
    }
  }
}

// ---- backward WITH weight gradients: contracted inside the block ----------------
// (round 5; replaces the round-2..4 scheme that left 16 KB of weight-gradient
// operands per sample in HBM for a second kernel: 372 MB written + 477 MB read
// at 24.5 k samples, 10 x the algorithmic bytes.)
//
// A block = 8 waves = ONE group of 128 sample points.  The chain is the one
// above; what changes is where the operands of  dW = sum_points g (x) x  go.
// The POINTS are the K dimension of that product while the chain keeps them on
// the lanes, so the 128-wide operand (g: d loss / d pre-activation, d loss / d
// layer output, F_theta's d loss / d pre-activation, the weighted hidden mean)
// is PUBLISHED to LDS in point-major rows (stride 144 floats: the two point
// rows of a 32-lane read group fall on distinct bank halves) and read back as
// the A fragments of  D[out][in] += A[out][point] B[point][in].  The other
// operand is read as B fragments
//   * from LDS where it fits: c rides in the 16 + 16 padding columns of the two
//     trunk buffers, the embedding of p overwrites the consumed d loss / d
//     layer output;
//   * from the rows the forward / this block left in global memory otherwise
//     (the previous layer's output: issued BEFORE the barrier that precedes the
//     contraction, the loads land behind the publish; F_theta's input rows and
//     g_c: 448 floats a point of scratch, L2-hot).
// Every wave owns a fixed set of output tiles and contracts them over ALL 128
// points of the block in MFMA accumulators (registers; no LDS or global
// atomics), then writes each tile as one 1 KB record of the block's partial;
// pc_dw_reduce_kernel sums the records over the blocks into the flat gradient.
// Bias gradients = one more MFMA per K-step against a column of ones.  The
// output layer (3 rows) and the two small sums (b2, B_rel) are lane products
// summed over the wave's 16 points on DPP and filed per wave (same-address
// atomics of every wave of every block cost ~13 us PER INSTRUCTION: 180 us of
// the first version's 670).
//
// LDS does not hold a layer's staged fragments and the published operands at
// once (121 + 147 KB): in the trunk the two 128 x 144 operand buffers OVERLAY
// the layer's weight stage after a barrier says every wave has consumed it
// (4 barriers a layer instead of 2); F_theta keeps its fragments resident
// (60 KB: W2^T is needed once — d loss / d hidden = w_k (W2^T g_c) — and is
// staged separately in front) next to one operand buffer.
// Two identities shrink F_theta's products:  d W2 = sum_s g_c (x) (sum_k w_k
// h_k)  (one product per sample instead of eight) and  g_a,k = w_k u (.)
// softplus'(a_k),  u = W2^T g_c  once per sample.
//
// Every fragment read of this kernel is a PINNED ds_read (common.h) issued
// one to four K-steps ahead by template recursion: next to the kernel's live
// state the scheduler waited on almost every read (104 waits for the 256 MFMAs
// of a trunk layer).  Measured (profiles/r05_pointslam_*): block time line
// 365 us at 24.5 k samples on an otherwise idle GPU; under the sustained load
// of the frame loop the launch group is 534 us (round 4: 566), counter traffic
// in DESIGN.md 4.9b.
constexpr int WB_PW = 8, WB_PTS = 16 * WB_PW, WB_S = 144;
constexpr int WB_GS0 = 0, WB_GS1 = WB_PTS * WB_S;   // trunk: gh | gz (floats)
constexpr int WB_TAIL = 2 * WB_PTS * WB_S;          // OW, OB, BEMB
// F_theta phase: [W1 frag | B1 | BREL | W1TF | W1TE] | operand buffer | tiles
constexpr int WB_F_B1 = 8 * 13 * 64, WB_F_BREL = WB_F_B1 + 128,
              WB_F_W1TF = WB_F_BREL + 40, WB_F_W1TE = WB_F_W1TF + 2 * 32 * 64,
              WB_F_END = WB_F_W1TE + 2 * 32 * 64;
constexpr int WB_GSF = (WB_F_END + 63) / 64 * 64;
constexpr int WB_TILE = WB_GSF + WB_PTS * WB_S;
constexpr int WB_LDS_FLOATS =
    (WB_TILE + WB_PW * kTileLen) > (WB_TAIL + kBwdTailLen + 4)
        ? (WB_TILE + WB_PW * kTileLen) : (WB_TAIL + kBwdTailLen + 4);
constexpr int kWbLds = WB_LDS_FLOATS * (int)sizeof(float);
static_assert(kWbLds <= 160 * 1024, "backward (weights) LDS");
static_assert(kBwdFcOff + 8 * 8 * 64 + 128 <= WB_TAIL, "trunk stage below tail");
static_assert(PcPack::B1 == PcPack::W1 + 8 * 13 * 64, "W1 | B1 contiguous");
static_assert(PcPack::W1TE == PcPack::W1TF + 2 * 32 * 64, "W1TF | W1TE");
static_assert(WB_F_W1TF % 4 == 0, "16-byte staging");

// operands that stay in global memory (floats per point: 448)
struct WbOps {
  float *gc, *fx;
  __host__ __device__ WbOps(float* base, int64_t n) {
    gc = base;              // [n][32]     d loss / d interpolated feature c
    fx = gc + n * 32;       // [8][n][52]  F_theta input [e_rel | f], neighbour major
  }
};
constexpr int64_t kWbOpsPerPoint = 32 + 8 * 52;

// records (16 x 16 tiles, 256 floats: [r][lane] = D-layout registers) of a
// block's partial
struct WbRec {
  static constexpr int p(int i) { return (i - 1) * 64; }          // i = 1..4: ot * 8 + it
  static constexpr int pb(int i) { return 256 + i * 8; }          // i = 0..4: ot
  static constexpr int e(int i) { return 296 + (i == 0 ? 0 : 24); }  // i = 0, 3: ot * 3 + it
  static constexpr int fc(int i) { return 344 + i * 16; }         // ot * 2 + it
  static constexpr int fcb(int i) { return 424 + i * 8; }         // ot
  static constexpr int O = 464;      // (464..472 unused: the output layer's
  static constexpr int OB = 472;     //  gradient is a per-wave VALU sum, OUT)
  static constexpr int W2 = 473;                                  // ot * 2 + it
  static constexpr int W1 = 489;                                  // ot * 4 + it
  static constexpr int B1 = 521;                                  // ot
  // per-wave row sums (no tile): [wave][32] and [wave][64] floats.  Atomics of
  // every wave of every block on the same 32 / 30 addresses cost ~13 us PER
  // INSTRUCTION (measured: 106 + 75 us of a 670 us backward)
  static constexpr int B2 = 529;
  static constexpr int BREL = 530;                                // 2 records
  // output layer: [wave][512] = d OW [3][128], d OB [3]
  static constexpr int OUT = 532;                                 // 16 records
  static constexpr int N = 548;
};
constexpr int kWbBlocks = 256;

struct WbProd {
  // kind 0: out x in, 1: transposed, 2: row bias, 3: [wave][32] sums -> off + j,
  // 4: [wave][64] B_rel sums, 5: [wave][512] output-layer sums
  int rec0, nrec, nit, kind;
  int off, ldo, M, N;
};
constexpr int WB_NPROD = 27;
struct WbProds {
  WbProd p[WB_NPROD];
};

template <int TILES>
__device__ __forceinline__ void publish_rows(float* __restrict__ gs, int lp,
                                             int q, const f32x4* v) {
#pragma unroll
  for (int jt = 0; jt < TILES; ++jt)
    *reinterpret_cast<f32x4*>(gs + lp * WB_S + 16 * jt + 4 * q) = v[jt];
}

// B fragment source: the block's 128 point-major rows in global memory, from
// row `brow` (all 128 exist: the last, partial group is shifted back to rows
// n - 128 .. n - 1 and its already-processed points publish zeros).  Wave-
// uniform base + one 32-bit lane offset: the K-steps differ by a scalar (one
// address pair per load — what a per-load row clamp costs — ran the kernel out
// of registers).  Columns beyond the operand's width read the neighbouring
// row: they only reach output columns the reduction drops.  CLAMP: n < 128
// (one short group): rows beyond n - 1 read row n - 1, their A entries are 0.
// MASKED: rows written by THIS launch (the ops scratch): in the shifted group
// the rows below `first` belong to another block and may not be written yet —
// stale bits (NaN) times a zero A entry would poison the product: read as 0.
template <bool CLAMP, bool MASKED>
struct GlbRowsT {
  const float* __restrict__ base;
  int stride, off, q, lastl, v0;
  __device__ __forceinline__ GlbRowsT(const float* b, int stride_, int64_t brow,
                                      int64_t n, int64_t first, int q_, int li)
      : base(b + brow * stride_), stride(stride_), off(q_ * stride_ + li),
        q(q_), lastl((int)(n - 1 - brow)), v0((int)(first - brow)) {}
  __device__ __forceinline__ float at(int ks, int it) const {
    float v;
    if constexpr (CLAMP) {
      int r = q + 4 * ks;
      r = r < lastl ? r : lastl;
      v = base[r * stride + (off - q * stride) + 16 * it];
    } else {
      v = base[ks * 4 * stride + 16 * it + off];
    }
    if constexpr (MASKED) v = 4 * ks + q >= v0 ? v : 0.f;
    return v;
  }
  // the same rows, in tiles it0, it0 + 1, ...
  __device__ __forceinline__ GlbRowsT tiles_from(int it0) const {
    GlbRowsT r = *this;
    r.off += 16 * it0;
    return r;
  }
};

__device__ __forceinline__ void put_record(float* __restrict__ part, int rec,
                                           int lane, const f32x4& v) {
  // (wave-uniform base + 32-bit lane offset: no 64-bit address pair per record)
  const int off = rec * 256 + lane;
#pragma unroll
  for (int r = 0; r < 4; ++r) part[off + r * 64] = v[r];
}

template <int I, int N, int STRIDE, int OFF>
__device__ __forceinline__ void lds_row(uint32_t addr, float* v) {
  if constexpr (I < N) {
    v[I] = lds_async<OFF + I * STRIDE>(addr);
    lds_row<I + 1, N, STRIDE, OFF>(addr, v);
  }
}

// dense_h / dense_e with the fragment reads PINNED and pipelined one K-step
// ahead (2 JT registers) instead of left to the scheduler: next to this
// kernel's live state the compiler batched the reads 2-3 deep and waited on
// almost every one (104 waits for the 256 MFMAs of a trunk layer; 33 in the
// kernel above).  in_at(s): the B operand of K-step s.
template <int JT, int KTOT, int KS, int S, class In>
__device__ __forceinline__ void dense_p_step(uint32_t addr, const In& in_at,
                                             f32x4* acc, float* fa, float* fb) {
  if constexpr (S + 1 < KS) lds_row<0, JT, KTOT * 256, (S + 1) * 256>(addr, fb);
  lds_landed<(S + 1 < KS) ? JT : 0>();
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) lds_tie(fa[jt]);
  const float v = in_at(S);
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) acc[jt] = XRD_MFMA4(fa[jt], v, acc[jt]);
  if constexpr (S + 1 < KS)
    dense_p_step<JT, KTOT, KS, S + 1>(addr, in_at, acc, fb, fa);
}
template <int JT, int KTOT, int KS, class In>
__device__ __forceinline__ void dense_p(const float* __restrict__ w, int lane,
                                        int s0, const In& in_at, f32x4* acc) {
  static_assert(((JT - 1) * KTOT + KS) * 256 <= 65536, "immediate offsets");
  const uint32_t addr = lds_addr(w + lane) + s0 * 256;
  float fa[JT], fb[JT];
  lds_row<0, JT, KTOT * 256, 0>(addr, fa);
  dense_p_step<JT, KTOT, KS, 0>(addr, in_at, acc, fa, fb);
}
template <int JT, int KTOT, int KS>
__device__ __forceinline__ void dense_hp(const float* __restrict__ w, int lane,
                                         int s0, const f32x4* in, f32x4* acc) {
  dense_p<JT, KTOT, KS>(w, lane, s0,
                        [&](int s) { return in[s >> 2][s & 3]; }, acc);
}
template <int JT, int KTOT, int KS>
__device__ __forceinline__ void dense_ep(const float* __restrict__ w, int lane,
                                         int s0, const float* in, f32x4* acc) {
  dense_p<JT, KTOT, KS>(w, lane, s0, [&](int s) { return in[s]; }, acc);
}
constexpr int WB_KSTEP = 4 * WB_S * 4;     // bytes between K-steps of an operand

// one K-step of contract_1xn: the A fragment of step S + 4 is issued, the one
// of step S waited for (ring of 5 registers; lgkmcnt counts to 15)
template <int NB, bool BIAS, int CH, int S>
__device__ __forceinline__ void c1xn_step(uint32_t g0, uint32_t g1,
                                          const float (*cur)[NB], f32x4* acc,
                                          f32x4& accb, float* a) {
  constexpr int D = 4;
  if constexpr (S + D < CH) {
    if constexpr (S + D < 16)
      a[(S + D) % 5] = lds_async<(S + D) * WB_KSTEP>(g0);
    else
      a[(S + D) % 5] = lds_async<(S + D - 16) * WB_KSTEP>(g1);
  }
  lds_landed<(CH - 1 - S < D) ? CH - 1 - S : D>();
  lds_tie(a[S % 5]);
#pragma unroll
  for (int it = 0; it < NB; ++it)
    acc[it] = XRD_MFMA4(a[S % 5], cur[S][it], acc[it]);
  if (BIAS) accb = XRD_MFMA4(a[S % 5], 1.f, accb);
  if constexpr (S + 1 < CH)
    c1xn_step<NB, BIAS, CH, S + 1>(g0, g1, cur, acc, accb, a);
}

// the B fragments of the first CH K-steps (issued by a caller that knows them
// to be written earlier than the A operand is published)
template <int NB, int CH, class Rows>
__device__ __forceinline__ void load_b(const Rows& B, float (*cur)[NB]) {
#pragma unroll
  for (int s = 0; s < CH; ++s)
#pragma unroll
    for (int it = 0; it < NB; ++it) cur[s][it] = B.at(s, it);
}

template <int NB, bool BIAS, int CH, class Rows, bool PRELOADED = false>
__device__ __forceinline__ void contract_1xn(const float* __restrict__ gs,
                                             const Rows& B, f32x4* acc,
                                             f32x4& accb,
                                             float (*pre)[NB] = nullptr) {
  // A fragments: LDS, pinned reads 4 K-steps ahead; B fragments: global, CH
  // K-steps per batch of loads (one exposed L2 / HBM round trip per batch:
  // CH = 32 where the registers allow), the next batch in flight while this
  // one is contracted
  constexpr int NCH = 32 / CH;
  float cur[CH][NB], nxt[NCH > 1 ? CH : 1][NB], a[5];
  uint32_t ga = lds_addr(gs);
#pragma unroll
  for (int s = 0; s < CH; ++s)
#pragma unroll
    for (int it = 0; it < NB; ++it)
      cur[s][it] = PRELOADED ? pre[s][it] : B.at(s, it);
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    if constexpr (NCH > 1) {
      if (c < NCH - 1) {
#pragma unroll
        for (int s = 0; s < CH; ++s)
#pragma unroll
          for (int it = 0; it < NB; ++it)
            nxt[s][it] = B.at(CH * (c + 1) + s, it);
      }
    }
    a[0] = lds_async<0>(ga);
    a[1] = lds_async<WB_KSTEP>(ga);
    a[2] = lds_async<2 * WB_KSTEP>(ga);
    a[3] = lds_async<3 * WB_KSTEP>(ga);
    c1xn_step<NB, BIAS, CH, 0>(ga, ga + 16 * WB_KSTEP, cur, acc, accb, a);
    if constexpr (NCH > 1) {
      ga += CH * WB_KSTEP;
#pragma unroll
      for (int s = 0; s < CH; ++s)
#pragma unroll
        for (int it = 0; it < NB; ++it) cur[s][it] = nxt[s][it];
    }
  }
}

// acc[it] += A(out tile at ga) x B(in tile it at gb + 64 it bytes), both
// operands in LDS (pinned reads, 2 K-steps ahead), over the 128 points;
// accb += A x 1.
template <int NB, bool BIAS, int S>
__device__ __forceinline__ void cl_step(uint32_t ga0, uint32_t gb0,
                                        uint32_t ga1, uint32_t gb1, f32x4* acc,
                                        f32x4& accb, float (*f)[NB + 1]) {
  constexpr int D = 2;
  if constexpr (S + D < 32) {
    constexpr int T = S + D, O = (T < 16 ? T : T - 16) * WB_KSTEP;
    f[T % 3][0] = lds_async<O>(T < 16 ? ga0 : ga1);
    lds_row<0, NB, 64, O>(T < 16 ? gb0 : gb1, &f[T % 3][1]);
  }
  lds_landed<(31 - S < D ? 31 - S : D) * (NB + 1)>();
#pragma unroll
  for (int j = 0; j <= NB; ++j) lds_tie(f[S % 3][j]);
#pragma unroll
  for (int it = 0; it < NB; ++it)
    acc[it] = XRD_MFMA4(f[S % 3][0], f[S % 3][1 + it], acc[it]);
  if (BIAS) accb = XRD_MFMA4(f[S % 3][0], 1.f, accb);
  if constexpr (S + 1 < 32)
    cl_step<NB, BIAS, S + 1>(ga0, gb0, ga1, gb1, acc, accb, f);
}
template <int NB, bool BIAS>
__device__ __forceinline__ void contract_lds(const float* __restrict__ a_s,
                                             const float* __restrict__ b_s,
                                             f32x4* acc, f32x4& accb) {
  static_assert(2 * (NB + 1) <= 15, "lgkmcnt");
  const uint32_t ga0 = lds_addr(a_s), gb0 = lds_addr(b_s);
  float f[3][NB + 1];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    f[s][0] = s == 0 ? lds_async<0>(ga0) : lds_async<WB_KSTEP>(ga0);
    if (s == 0) lds_row<0, NB, 64, 0>(gb0, &f[s][1]);
    else lds_row<0, NB, 64, WB_KSTEP>(gb0, &f[s][1]);
  }
  cl_step<NB, BIAS, 0>(ga0, gb0, ga0 + 16 * WB_KSTEP, gb0 + 16 * WB_KSTEP, acc,
                       accb, f);
}

// the 8 out tiles + the bias tile of one K-step
struct Frag9 {
  float a[8], ab;
  template <int OFF>
  __device__ __forceinline__ void load(uint32_t ga, uint32_t gb) {
    lds_row<0, 8, 64, OFF>(ga, a);
    ab = lds_async<OFF>(gb);
  }
  template <int PENDING>
  __device__ __forceinline__ void landed() {
    lds_landed<PENDING>();
#pragma unroll
    for (int i = 0; i < 8; ++i) lds_tie(a[i]);
    lds_tie(ab);
  }
  __device__ __forceinline__ void mma(float b, f32x4* acc, f32x4& accb) const {
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) acc[ot] = XRD_MFMA4(a[ot], b, acc[ot]);
    accb = XRD_MFMA4(ab, 1.f, accb);
  }
};

// acc[ot] += A(out tile ot) x B(one in tile), ot < 8; accb += A(out tile
// `wave`) x 1.  gs: this lane's address for out tile 0; b: the B fragments of
// the 32 K-steps (loaded by the caller BEFORE the barrier in front of the
// contraction).  The 9 fragment reads of K-step s + 1 are issued before the
// MFMAs of step s.
template <int S>
__device__ __forceinline__ void c8x1_step(uint32_t ga0, uint32_t gb0,
                                          uint32_t ga1, uint32_t gb1,
                                          const float* b, f32x4* acc,
                                          f32x4& accb, Frag9& fa, Frag9& fb) {
  // (two bases: an immediate offset reaches 64 KB, 32 K-steps span 72 KB)
  if constexpr (S + 1 < 16)
    fb.template load<(S + 1) * WB_KSTEP>(ga0, gb0);
  else if constexpr (S + 1 < 32)
    fb.template load<(S + 1 - 16) * WB_KSTEP>(ga1, gb1);
  if constexpr (S + 1 < 32)
    fa.template landed<9>();
  else
    fa.template landed<0>();
  fa.mma(b[S], acc, accb);
  if constexpr (S + 1 < 32)
    c8x1_step<S + 1>(ga0, gb0, ga1, gb1, b, acc, accb, fb, fa);
}
__device__ __forceinline__ void contract_8x1(const float* __restrict__ gs,
                                             int wave, const float* b,
                                             f32x4* acc, f32x4& accb) {
  const uint32_t ga0 = lds_addr(gs), gb0 = ga0 + 64 * wave;
  const uint32_t ga1 = ga0 + 16 * WB_KSTEP, gb1 = gb0 + 16 * WB_KSTEP;
  Frag9 f0, f1;
  f0.load<0>(ga0, gb0);
  c8x1_step<0>(ga0, gb0, ga1, gb1, b, acc, accb, f0, f1);
}

template <bool CLAMP>
__global__ __launch_bounds__(WB_PW * 64, 1) void point_color_bwd_w_kernel(
    int64_t n, int64_t grp0, const float* __restrict__ pts, const int64_t* __restrict__ nbr,
    const int* __restrict__ n_nb, const float* __restrict__ cloud,
    const float* __restrict__ feats, const float* __restrict__ radius,
    float radius_all, int min_nn, const float* pk,
    const float* __restrict__ rgb, const float* __restrict__ save_c,
    const float* __restrict__ save_h, const float* __restrict__ save_y,
    const float* __restrict__ g_rgb, float* __restrict__ g_pts,
    float* __restrict__ g_feats, float* __restrict__ g_flat,
    float* __restrict__ ops_base, float* __restrict__ ws) {
  using K = PcPack;
  using R = WbRec;
  float* wl = reinterpret_cast<float*>(pc_smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  const int lp = wave * 16 + li;            // point of this lane in the block
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const WbOps ops(ops_base, n);
  using GlbRows = GlbRowsT<CLAMP, false>;    // rows the forward saved
  using OpsRows = GlbRowsT<CLAMP, true>;     // rows this launch writes
  float* part = ws + (int64_t)blockIdx.x * R::N * 256;
  float* tail = wl + WB_TAIL;
  const float* ow = tail;
  const float* bemb = tail + (K::BEMB - K::OW);
  {
    // one group of 128 points per block.  The last, partial group covers the
    // LAST 128 rows: its points below first are the previous group's and stay
    // out (valid = false: they publish zeros, write nothing)
    const int64_t first = (grp0 + blockIdx.x) * WB_PTS;
    const int64_t row0 = CLAMP || first + WB_PTS <= n ? first : n - WB_PTS;
    const int64_t pt = row0 + lp;
    const bool valid = pt >= first && pt < n;
    asm volatile("" : "+s"(pk));
    for (int i = threadIdx.x; i < kBwdTailLen; i += blockDim.x)
      tail[i] = pk[K::OW + i];
    __syncthreads();
    float p[3] = {0.f, 0.f, 0.f};
    if (valid) {
#pragma unroll
      for (int a = 0; a < 3; ++a) p[a] = pts[pt * 3 + a];
    }
    // ---- sigmoid, output layer ----------------------------------------------------
    float go[3] = {0.f, 0.f, 0.f};
    if (valid) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float o = rgb[pt * 3 + r];
        go[r] = g_rgb[pt * 3 + r] * o * (1.f - o);
      }
    }
    f32x4 g_h[8];
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
      g_h[jt] = z4;
#pragma unroll
      for (int r = 0; r < 3; ++r)
        g_h[jt] += *reinterpret_cast<const f32x4*>(ow + r * 128 + 16 * jt +
                                                   4 * q) * go[r];
    }
    f32x4 c[2];
    load_rows<2>(save_c, 32, pt, q, valid, c);
    // embedding of p (the B operand of two products); recomputed after the
    // trunk for d / d p: what the trunk does not need does not stay in registers
    auto embed = [&](float* e) {
      const float p2[3] = {kTwoPi * p[0], kTwoPi * p[1], kTwoPi * p[2]};
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bemb + (4 * s + q) * 4);
        float a = p2[0] * b[0];
        a = fmaf(p2[1], b[1], a);
        a = fmaf(p2[2], b[2], a);
        sincos_cw(a, e[s], e[s + 5]);
      }
    };
    f32x4 g_c[2] = {z4, z4}, g_e[3] = {z4, z4, z4};
    // ---- trunk, last layer first ---------------------------------------------------
#pragma unroll 1
    for (int i = 4; i >= 0; --i) {
      // (keeps the per-layer address arithmetic inside the loop: hoisted out
      // of it, it is spilled)
      int64_t pti = pt;
      asm volatile("" : "+v"(pti));
      // the layer's saved output (softplus' comes from it): issued in front
      // of the staging copy, it lands behind it
      f32x4 h[8];
      load_rows<8>(save_h + (int64_t)i * n * 128, 128, pti, q, valid, h);
      __syncthreads();      // the contraction of layer i + 1 has read its operands
      pc_copy(wl, pk + K::rw(i), K::rlen(i));
      pc_copy(wl + kBwdFcOff, pk + K::tfc(i), 8 * 8 * 64 + 128);
      __syncthreads();
      const float* WT = wl;
      const float* ET = wl + (K::ret(i) - K::rw(i));
      const float* FCT = wl + (K::rfc(i) - K::rw(i));
      const float* FCw = wl + kBwdFcOff;
      const float* FCb = FCw + 8 * 8 * 64;
      dense_hp<2, 32, 32>(FCT, lane, 0, g_h, g_c);
      f32x4 g_z[8];
      {
        f32x4 cc[8];
        if (i == 4) {
          // output layer (3 rows): d OW = sum_points go (x) h_4 as lane
          // products summed over the wave's 16 points on DPP (an MFMA product
          // with both operands read from global cost ~20 us of round trips)
          float* o = part + R::OUT * 256 + wave * 512;
#pragma unroll
          for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int jt = 0; jt < 8; ++jt)
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const float v = row16_sum_dpp(go[r] * h[jt][t]);
                if (li == 0) o[r * 128 + 16 * jt + 4 * q + t] = v;
              }
            const float v = row16_sum_dpp(go[r]);
            if (lane == 0) o[384 + r] = v;
          }
        }
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) cc[jt] = bias4(FCb, jt, q);
        dense_hp<8, 8, 8>(FCw, lane, 0, c, cc);
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            g_z[jt][t] = g_h[jt][t] *
                         softplus100_grad_of_value(h[jt][t] - cc[jt][t]);
      }
      if (i == 0 || i == 3) dense_hp<3, 32, 32>(ET, lane, 0, g_z, g_e);
      f32x4 gp[8];
      if (i >= 1) {
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) gp[jt] = z4;
        dense_hp<8, 32, 32>(WT, lane, 0, g_z, gp);
      }
      // ---- the layer's weight gradients ------------------------------------------
      // B fragments of the hidden-column product: issued here, they land
      // behind the barrier + publish below (an exposed round trip otherwise)
      float bP[32];
      if (i >= 1) {
        const GlbRows B(save_h + (int64_t)(i - 1) * n * 128, 128, row0, n,
                        first, q, li);
#pragma unroll
        for (int s = 0; s < 32; ++s) bP[s] = B.at(s, wave);
      }
      __syncthreads();      // every wave is done with the staged fragments
      publish_rows<8>(wl + WB_GS0, lp, q, g_h);
      publish_rows<8>(wl + WB_GS1, lp, q, g_z);
      // c rides in the 16 + 16 padding columns of the two rows
      *reinterpret_cast<f32x4*>(wl + WB_GS0 + lp * WB_S + 128 + 4 * q) = c[0];
      *reinterpret_cast<f32x4*>(wl + WB_GS1 + lp * WB_S + 128 + 4 * q) = c[1];
      __syncthreads();
      {
        const float* gh_s = wl + WB_GS0 + q * WB_S + li;
        const float* gz_s = wl + WB_GS1 + q * WB_S + li;
        if (i >= 1) {   // hidden columns: in tile `wave` x the 8 out tiles
          f32x4 acc[8], accb = z4;
#pragma unroll
          for (int ot = 0; ot < 8; ++ot) acc[ot] = z4;
          contract_8x1(gz_s, wave, bP, acc, accb);
#pragma unroll
          for (int ot = 0; ot < 8; ++ot)
            put_record(part, R::p(i) + ot * 8 + wave, lane, acc[ot]);
          put_record(part, R::pb(i) + wave, lane, accb);
        }
        {   // FC_i: out tile `wave` of gh x the 2 tiles of c, bias from gh
          f32x4 a0 = z4, a1 = z4, accb = z4, none = z4;
          contract_lds<1, true>(gh_s + 16 * wave, gh_s + 128, &a0, accb);
          contract_lds<1, false>(gh_s + 16 * wave, gz_s + 128, &a1, none);
          put_record(part, R::fc(i) + wave * 2, lane, a0);
          put_record(part, R::fc(i) + wave * 2 + 1, lane, a1);
          put_record(part, R::fcb(i) + wave, lane, accb);
        }
        if (i == 0 || i == 3) {
          // embedding columns: the embedding of p replaces gh (its products
          // are done) as the B operand: out tile `wave` of gz x 3 tiles
          __syncthreads();
          {
            float e[10];
            embed(e);
            float* row = wl + WB_GS0 + lp * WB_S;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
              row[4 * s + q] = e[s];
              row[20 + 4 * s + q] = e[s + 5];
            }
          }
          __syncthreads();
          f32x4 acc[3] = {z4, z4, z4}, accb = z4;
          if (i == 0) {
            contract_lds<3, true>(gz_s + 16 * wave, gh_s, acc, accb);
            put_record(part, R::pb(0) + wave, lane, accb);
          } else {
            contract_lds<3, false>(gz_s + 16 * wave, gh_s, acc, accb);
          }
#pragma unroll
          for (int it = 0; it < 3; ++it)
            put_record(part, R::e(i) + wave * 3 + it, lane, acc[it]);
        }
      }
      if (i >= 1) {
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) g_h[jt] = gp[jt];
      }
    }
    // ---- d / d p through the embedding -------------------------------------------
    float gp[3] = {0.f, 0.f, 0.f};
    float e[10];
    embed(e);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const float garg = g_e[s >> 2][s & 3] * e[s + 5] -
                         g_e[(s + 5) >> 2][(s + 5) & 3] * e[s];
      const f32x4 b = *reinterpret_cast<const f32x4*>(bemb + (4 * s + q) * 4);
#pragma unroll
      for (int a = 0; a < 3; ++a) gp[a] = fmaf(garg, b[a], gp[a]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) gp[a] = group4_sum(gp[a]) * kTwoPi;
    // ---- interpolation weights ----------------------------------------------------
    PointNb nb;
    point_neighbors(nbr, cloud, n_nb, radius, radius_all, min_nn, pt, valid, p,
                    nb);
    if (!nb.has) g_c[0] = g_c[1] = z4;
    if (valid) save_rows<2>(ops.gc, 32, pt, q, g_c);
    float gD[8], wsum = 0.f;
    {
      float gw[8], sum = 0.f;
      f32x4 y[8][2];   // (all eight rows in flight before the first product)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        load_rows<2>(save_y, 32, pt * 8 + k, q, valid, y[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float d = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          d += g_c[0][t] * y[k][0][t] + g_c[1][t] * y[k][1][t];
        gw[k] = group4_sum(d);
        sum += gw[k] * (nb.u[k] / nb.den);
        if (nb.has && nb.u[k] != 0.f) wsum += nb.u[k] / nb.den;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        gD[k] = (nb.has && nb.u[k] != 0.f)
                    ? -(nb.u[k] * nb.u[k]) * ((gw[k] - sum) / nb.den)
                    : 0.f;
    }
    // d loss / d b2 = sum over samples and neighbours of w_k g_c
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float v = row16_sum(g_c[jt][t] * wsum);
        float* o = part + R::B2 * 256 + wave * 32 + 16 * jt + 4 * q + t;
        if (li == 0) *o = v;
      }
    // ---- F_theta: u = W2^T g_c once ---------------------------------------------
    __syncthreads();        // layer 0's contraction has read its operands
    pc_copy(wl, pk + K::W2T, 8 * 8 * 64);
    __syncthreads();
    f32x4 u[8];
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) u[jt] = z4;
    dense_hp<8, 8, 8>(wl, lane, 0, g_c, u);
    __syncthreads();
    pc_copy(wl, pk + K::W1, 8 * 13 * 64 + 128);
    pc_copy(wl + WB_F_BREL, pk + K::BREL, 40);
    pc_copy(wl + WB_F_W1TF, pk + K::W1TF, 4 * 32 * 64);
    __syncthreads();
    const float* W1 = wl;
    const float* brel = wl + WB_F_BREL;
    f32x4 hbar[8], w1_acc[4] = {z4, z4, z4, z4}, b1_acc = z4;
    // gathers of a neighbour's position / colour feature row, one neighbour
    // ahead (an exposed L2 / HBM round trip per neighbour otherwise)
    float nbx[3] = {0.f, 0.f, 0.f};
    f32x4 nbf[2] = {z4, z4};
    auto gather_nb = [&](int k2) {
      const int id2 = pick8(nb.id, k2);
      if (nb.has && pick8(nb.u, k2) != 0.f) {
#pragma unroll
        for (int a = 0; a < 3; ++a) nbx[a] = cloud[(int64_t)id2 * 3 + a];
        nbf[0] = *reinterpret_cast<const f32x4*>(feats + (int64_t)id2 * 32 +
                                                 4 * q);
        nbf[1] = *reinterpret_cast<const f32x4*>(feats + (int64_t)id2 * 32 +
                                                 16 + 4 * q);
      }
    };
    gather_nb(0);
    // d loss / d B_rel: lane (q, .) owns columns fidx(4s + q), s < 5
    float brel_acc[5][3];
#pragma unroll
    for (int s = 0; s < 5; ++s)
      brel_acc[s][0] = brel_acc[s][1] = brel_acc[s][2] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) hbar[jt] = z4;
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      asm volatile("" ::: "memory");
      int64_t ptk = pt;
      asm volatile("" : "+v"(ptk));
      const float uk = pick8(nb.u, k);
      const bool live = nb.has && uk != 0.f;
      const int id = pick8(nb.id, k);
      // (this neighbour's position and feature row were gathered before the
      // previous neighbour's contraction: nbx / nbf)
      float raw[3], rel[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) raw[a] = live ? nbx[a] - p[a] : 0.f;
      const f32x4 f[2] = {live ? nbf[0] : z4, live ? nbf[1] : z4};
#pragma unroll
      for (int a = 0; a < 3; ++a) rel[a] = kTwoPi * raw[a];
      float e5[5], d5[5];
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int j = 4 * s + q, fi = j < 10 ? j : j - 10;
        const f32x4 b = *reinterpret_cast<const f32x4*>(brel + fi * 4);
        float a = rel[0] * b[0];
        a = fmaf(rel[1], b[1], a);
        a = fmaf(rel[2], b[2], a);
        float sn, cs;
        sincos_cw(a, sn, cs);
        e5[s] = j < 10 ? sn : cs;
        d5[s] = j < 10 ? cs : -sn;
      }
      if (valid) {   // F_theta's input row: the B operand of d W1
        float* x = ops.fx + ((int64_t)k * n + ptk) * 52;
#pragma unroll
        for (int s = 0; s < 5; ++s) x[4 * s + q] = e5[s];
        *reinterpret_cast<f32x4*>(x + 20 + 4 * q) = f[0];
        *reinterpret_cast<f32x4*>(x + 36 + 4 * q) = f[1];
      }
      const float w = live ? uk / nb.den : 0.f;
      f32x4 g_a[8];
      {
        f32x4 a[8];
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) a[jt] = bias4(wl + WB_F_B1, jt, q);
        dense_ep<8, 13, 5>(W1, lane, 0, e5, a);
        dense_hp<8, 13, 8>(W1, lane, 5, f, a);
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            g_a[jt][t] = (w * u[jt][t]) * softplus100_grad(a[jt][t]);
            hbar[jt][t] = fmaf(w, softplus100(a[jt][t]), hbar[jt][t]);
          }
      }
      if (g_feats != nullptr) {
        f32x4 g_f[2] = {z4, z4};
        dense_hp<2, 32, 32>(wl + WB_F_W1TF, lane, 0, g_a, g_f);
        float* T = wl + WB_TILE + wave * kTileLen;   // [16][33]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          T[li * 33 + 4 * q + t] = g_f[0][t];
          T[li * 33 + 16 + 4 * q + t] = g_f[1][t];
        }
        wave_lds_sync();
        const int idl = live ? id : -1;
        const int ff = lane & 31, half = lane >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int pt2 = 2 * i + half;
          const int id2 = __shfl(idl, pt2);
          if (id2 >= 0)
            atomicAdd(g_feats + (int64_t)id2 * 32 + ff, T[pt2 * 33 + ff]);
        }
        wave_lds_sync();
      }
      f32x4 g_er[2] = {z4, z4};
      dense_hp<2, 32, 32>(wl + WB_F_W1TE, lane, 0, g_a, g_er);
      float grel[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int j = 4 * s + q, fi = j < 10 ? j : j - 10;
        const float garg = g_er[s >> 2][s & 3] * d5[s];
        const f32x4 b = *reinterpret_cast<const f32x4*>(brel + fi * 4);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          grel[a] = fmaf(garg, b[a], grel[a]);
          brel_acc[s][a] = fmaf(garg, rel[a], brel_acc[s][a]);
        }
      }
      const float gDk = pick8(gD, k);
#pragma unroll
      for (int a = 0; a < 3; ++a)
        gp[a] -= group4_sum(grel[a]) * kTwoPi + 2.f * raw[a] * gDk;
      // ---- d W1 += g_a (x) x over the block's points -------------------------------
      __syncthreads();      // neighbour k - 1's contraction is done; every
                            // wave's x rows of this neighbour are written
      {
        const OpsRows B(ops.fx + (int64_t)k * n * 52, 52, row0, n, first, q, li);
        // the first pass's B fragments land behind the publish + barrier, the
        // next neighbour's gathers behind the contraction
        float b01[32][2];
        load_b<2, 32>(B, b01);
        if (k + 1 < 8) gather_nb(k + 1);
        publish_rows<8>(wl + WB_GSF, lp, q, g_a);
        __syncthreads();
        const float* ga_s = wl + WB_GSF + q * WB_S + li + 16 * wave;
        f32x4 none = z4;
        contract_1xn<2, true, 32, OpsRows, true>(ga_s, B, w1_acc, b1_acc, b01);
        contract_1xn<2, false, 32>(ga_s, B.tiles_from(2), w1_acc + 2, none);
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
      put_record(part, R::W1 + wave * 4 + it, lane, w1_acc[it]);
    put_record(part, R::B1 + wave, lane, b1_acc);
    // ---- d W2^T = hbar (x) g_c ------------------------------------------------------
    __syncthreads();
    {
      const OpsRows B(ops.gc, 32, row0, n, first, q, li);
      float bg[32][2];
      load_b<2, 32>(B, bg);
      publish_rows<8>(wl + WB_GSF, lp, q, hbar);
      __syncthreads();
      f32x4 acc[2] = {z4, z4}, none = z4;
      contract_1xn<2, false, 32, OpsRows, true>(
          wl + WB_GSF + q * WB_S + li + 16 * wave, B, acc, none, bg);
      put_record(part, R::W2 + wave * 2, lane, acc[0]);
      put_record(part, R::W2 + wave * 2 + 1, lane, acc[1]);
    }
    if (valid && q == 0 && g_pts != nullptr) {
#pragma unroll
      for (int a = 0; a < 3; ++a) g_pts[pt * 3 + a] = gp[a];
    }
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float v = row16_sum(brel_acc[s][a]);
        float* o = part + R::BREL * 256 + wave * 64 + (4 * s + q) * 3 + a;
        if (li == 0) *o = v;
      }
  }
}

// flat[dst] = sum over the blocks' partials of one record.  Block = record,
// 4 slices of the partials x 256 elements, 4 loads in flight a thread: the sum
// is a dependent chain of HBM round trips otherwise (192 partials).
__global__ __launch_bounds__(1024) void pc_dw_reduce_kernel(
    const WbProds prods, int n_blocks, int accumulate,
    const float* __restrict__ ws, float* __restrict__ flat) {
  __shared__ float red[3][256];
  const int rec = blockIdx.x, t = threadIdx.x & 255, slice = threadIdx.x >> 8;
  const float* __restrict__ src = ws + (int64_t)rec * 256 + t;
  constexpr int64_t STEP = (int64_t)WbRec::N * 256;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = slice;
  for (; b + 12 < n_blocks; b += 16) {
    s0 += src[b * STEP];
    s1 += src[(b + 4) * STEP];
    s2 += src[(b + 8) * STEP];
    s3 += src[(b + 12) * STEP];
  }
  for (; b < n_blocks; b += 4) s0 += src[b * STEP];
  float v = (s0 + s1) + (s2 + s3);
  if (slice > 0) red[slice - 1][t] = v;
  __syncthreads();
  if (slice > 0) return;
  v += (red[0][t] + red[1][t]) + red[2][t];
  int pi = 0;
#pragma unroll 1
  for (int j = 1; j < WB_NPROD; ++j)
    if (rec >= prods.p[j].rec0) pi = j;
  const WbProd& P = prods.p[pi];
  const int r = rec - P.rec0;
  if (r >= P.nrec) return;
  if (P.kind == 3) {        // 8 waves x 32 sums
    atomicAdd(flat + P.off + (t & 31), v);
    return;
  }
  if (P.kind == 5) {        // 8 waves x ([3][128] weights, [3] bias)
    const int e = (r * 256 + t) & 511;
    if (e < 384) atomicAdd(flat + P.off + e, v);
    else if (e < 387) atomicAdd(flat + P.ldo + (e - 384), v);
    return;
  }
  if (P.kind == 4) {        // 8 waves x (20 columns x 3 axes, 4 unused)
    const int e = (r * 256 + t) & 63, j = e / 3, ax = e - 3 * j;
    if (e < 60) atomicAdd(flat + P.off + ax * 10 + (j < 10 ? j : j - 10), v);
    return;
  }
  const int lane = t & 63, row = 4 * (lane >> 4) + (t >> 6), col = lane & 15;
  if (P.kind == 2) {
    const int o = 16 * r + row;
    if (col == 0 && o < P.M)
      flat[P.off + o] = accumulate ? flat[P.off + o] + v : v;
    return;
  }
  const int o = 16 * (r / P.nit) + row, c = 16 * (r % P.nit) + col;
  if (o >= P.M || c >= P.N) return;
  const int dst = P.kind == 1 ? P.off + c * P.ldo + o : P.off + o * P.ldo + c;
  flat[dst] = accumulate ? flat[dst] + v : v;
}

// host: packed <- flat index table (-1 = zero)
void build_pc_index(int32_t* idx) {
  using F = PcFlat;
  using K = PcPack;
  for (int i = 0; i < K::LEN; ++i) idx[i] = -1;
  for (int l = 0; l < 64; ++l) {
    const int m = l & 15, q = l >> 4;
    // ---- F_theta ------------------------------------------------------------------
    for (int jt = 0; jt < 8; ++jt)
      for (int s = 0; s < 13; ++s)
        idx[K::W1 + (jt * 13 + s) * 64 + l] =
            F::W1 + (16 * jt + m) * 52 + pc_ft_in(s, q);
    for (int jt = 0; jt < 2; ++jt)
      for (int s = 0; s < 32; ++s)
        idx[K::W2 + (jt * 32 + s) * 64 + l] =
            F::W2 + (16 * jt + m) * 128 + kmap(s, q);
    for (int kt = 0; kt < 8; ++kt)
      for (int s = 0; s < 8; ++s)   // d/d h = W2^T g_y
        idx[K::W2T + (kt * 8 + s) * 64 + l] =
            F::W2 + kmap(s, q) * 128 + 16 * kt + m;
    for (int kt = 0; kt < 2; ++kt)
      for (int s = 0; s < 32; ++s) {
        idx[K::W1TF + (kt * 32 + s) * 64 + l] =
            F::W1 + kmap(s, q) * 52 + 20 + 16 * kt + m;
        const int fe = emapT(kt, m);
        if (fe < 20)
          idx[K::W1TE + (kt * 32 + s) * 64 + l] = F::W1 + kmap(s, q) * 52 + fe;
      }
    // ---- trunk ---------------------------------------------------------------------
    for (int i = 0; i < 5; ++i) {
      const int S = K::ksteps(i), in = F::pin(i);
      for (int jt = 0; jt < 8; ++jt) {
        for (int s = 0; s < S; ++s)
          idx[K::tw(i) + (jt * S + s) * 64 + l] =
              F::pw(i) + (16 * jt + m) * in + pc_trunk_in(i, s, q);
        for (int s = 0; s < 8; ++s)
          idx[K::tfc(i) + (jt * 8 + s) * 64 + l] =
              F::fcw(i) + (16 * jt + m) * 32 + kmap(s, q);
      }
      const int hcol = i == 3 ? 40 : 0;
      if (i >= 1)
        for (int kt = 0; kt < 8; ++kt)
          for (int s = 0; s < 32; ++s)
            idx[K::rw(i) + (kt * 32 + s) * 64 + l] =
                F::pw(i) + kmap(s, q) * in + hcol + 16 * kt + m;
      if (i == 0 || i == 3)
        for (int kt = 0; kt < 3; ++kt)
          for (int s = 0; s < 32; ++s) {
            const int fe = emapT(kt, m);
            if (fe < 40)
              idx[K::ret(i) + (kt * 32 + s) * 64 + l] =
                  F::pw(i) + kmap(s, q) * in + fe;
          }
      for (int kt = 0; kt < 2; ++kt)
        for (int s = 0; s < 32; ++s)
          idx[K::rfc(i) + (kt * 32 + s) * 64 + l] =
              F::fcw(i) + kmap(s, q) * 32 + 16 * kt + m;
    }
  }
  for (int j = 0; j < 128; ++j) {
    idx[K::B1 + j] = F::B1 + j;
    for (int i = 0; i < 5; ++i) {
      idx[K::tb(i) + j] = F::pb(i) + j;
      idx[K::tfb(i) + j] = F::fcb(i) + j;
    }
  }
  for (int j = 0; j < 32; ++j) idx[K::B2 + j] = F::B2 + j;
  for (int f = 0; f < 10; ++f)
    for (int a = 0; a < 3; ++a) idx[K::BREL + f * 4 + a] = F::BREL + a * 10 + f;
  for (int f = 0; f < 20; ++f)
    for (int a = 0; a < 3; ++a) idx[K::BEMB + f * 4 + a] = F::BEMB + a * 20 + f;
  for (int r = 0; r < 3; ++r) {
    for (int j = 0; j < 128; ++j) idx[K::OW + r * 128 + j] = F::OW + r * 128 + j;
    idx[K::OB + r] = F::OB + r;
  }
}

}  // namespace
}  // namespace xrd

using namespace xrd;

template <class Kern>
static int pc_attr(Kern kern, int lds) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                          hipFuncAttributeMaxDynamicSharedMemorySize,
                          lds) != hipSuccess)
    return check_launch("hipFuncSetAttribute");
  return XRD_OK;
}

// waves per block: a block stages a whole layer per group of 16 PW points, so
// few points are spread over more, smaller blocks (one block per CU)
// waves per block (a block stages a whole layer per group of 16 PW points).
// Measured at 7 500 / 12 000 / 25 000 points (profiles/r02_pointslam_waves.txt):
// 4 waves beat 2 even when 2 would give twice the blocks; the forward gains
// from 8 (two waves per SIMD overlap) once 4 would need a second round of the
// 256 blocks, the register-heavier backward only at several rounds.
static int pc_waves_fwd(int64_t n) { return n > 16384 ? 8 : 4; }
static int pc_waves_bwd(int64_t n) { return n > 49152 ? 8 : 4; }

int64_t xrd_point_color_ops_floats(int64_t n_points) {
  // (+ 64: the tile reads of the narrow operands run past their last column)
  return n_points < 0 ? 0 : n_points * kWbOpsPerPoint + 64;
}

template <int PW>
static int pc_fwd(int64_t n, const float* points, const int64_t* neighbors,
                  const int32_t* n_neighbors, const float* cloud,
                  const float* col_feats, const float* radius,
                  float radius_all, int min_nn, const float* empty_feat,
                  const float* packed, float* rgb, float* save_c,
                  float* save_h, float* save_y, hipStream_t st) {
  static bool ready = false;
  if (!ready) {
    int rc = pc_attr(point_color_fwd_kernel<PW>, kPcLds);
    if (rc != XRD_OK) return rc;
    ready = true;
  }
  const int64_t groups = (n + 16 * PW - 1) / (16 * PW);
  const int nb = (int)(groups < kPcBlocks ? groups : kPcBlocks);
  hipLaunchKernelGGL(point_color_fwd_kernel<PW>, dim3(nb), dim3(PW * 64),
                     kPcLds, st, n, points, neighbors, n_neighbors, cloud,
                     col_feats, radius, radius_all, min_nn, empty_feat, packed,
                     rgb, save_c, save_h, save_y);
  return check_launch("xrd_point_color_fwd");
}

// the products of a backward in record order (pc_dw_reduce_kernel)
static WbProds wb_products() {
  using F = PcFlat;
  using R = WbRec;
  WbProds P;
  int k = 0;
  auto add = [&](int rec0, int nrec, int nit, int kind, int off, int ldo, int M,
                 int N) { P.p[k++] = WbProd{rec0, nrec, nit, kind, off, ldo, M, N}; };
  for (int i = 1; i < 5; ++i)      // hidden columns of the trunk layers
    add(R::p(i), 64, 8, 0, F::pw(i) + (i == 3 ? 40 : 0), F::pin(i), 128, 128);
  for (int i = 0; i < 5; ++i) add(R::pb(i), 8, 1, 2, F::pb(i), 0, 128, 0);
  add(R::e(0), 24, 3, 0, F::pw(0), 40, 128, 40);    // embedding columns
  add(R::e(3), 24, 3, 0, F::pw(3), 168, 128, 40);
  for (int i = 0; i < 5; ++i) add(R::fc(i), 16, 2, 0, F::fcw(i), 32, 128, 32);
  for (int i = 0; i < 5; ++i) add(R::fcb(i), 8, 1, 2, F::fcb(i), 0, 128, 0);
  add(R::W2, 16, 2, 1, F::W2, 128, 128, 32);        // [h][y] -> W2[y][h]
  add(R::W1, 32, 4, 0, F::W1, 52, 128, 52);
  add(R::B1, 8, 1, 2, F::B1, 0, 128, 0);
  add(R::B2, 1, 1, 3, F::B2, 0, 0, 0);
  add(R::BREL, 2, 1, 4, F::BREL, 0, 0, 0);
  add(R::OUT, 16, 1, 5, F::OW, F::OB, 0, 0);     // (ldo carries the bias offset)
  return P;
}

template <int PW>
static int pc_bwd(int64_t n, const float* points, const int64_t* neighbors,
                  const int32_t* n_neighbors, const float* cloud,
                  const float* col_feats, const float* radius,
                  float radius_all, int min_nn, const float* packed,
                  const float* rgb, const float* save_c, const float* save_h,
                  const float* save_y, const float* g_rgb, float* g_points,
                  float* g_feats, hipStream_t st) {
  static bool ready = false;
  if (!ready) {
    int rc = pc_attr(point_color_bwd_kernel<PW>, kBwdLds);
    if (rc != XRD_OK) return rc;
    ready = true;
  }
  const int64_t groups = (n + 16 * PW - 1) / (16 * PW);
  const int nb = (int)(groups < kPcBlocks ? groups : kPcBlocks);
  hipLaunchKernelGGL(point_color_bwd_kernel<PW>, dim3(nb), dim3(PW * 64),
                     kBwdLds, st, n, points, neighbors, n_neighbors, cloud,
                     col_feats, radius, radius_all, min_nn, packed, rgb, save_c,
                     save_h, save_y, g_rgb, g_points, g_feats);
  return check_launch("xrd_point_color_bwd");
}

extern "C" {

int xrd_point_color_flat_len(void) { return PcFlat::LEN; }
int xrd_point_color_grad_len(void) { return PcFlat::N_GRAD; }
int xrd_point_color_pack_len(void) { return PcPack::LEN; }

int xrd_point_color_pack_index(int32_t* idx) {
  if (idx == nullptr) return XRD_ERR_ARG;
  build_pc_index(idx);
  return XRD_OK;
}


int64_t xrd_point_color_ws_floats(void) {
  return (int64_t)kWbBlocks * WbRec::N * 256;   // one partial per block
}


int xrd_point_color_fwd(int64_t n_points, const float* points,
                        const int64_t* neighbors, const int32_t* n_neighbors,
                        const float* cloud, const float* col_feats,
                        const float* radius, float radius_all, int min_nn,
                        const float* empty_feat, const float* packed,
                        float* rgb, float* save_c, float* save_h,
                        float* save_y, xrd_stream_t stream) {
  if (n_points < 0 || min_nn < 0) return XRD_ERR_ARG;
  if (n_points == 0) return XRD_OK;
  if (!points || !neighbors || !n_neighbors || !cloud || !col_feats ||
      !empty_feat || !packed || !rgb)
    return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define XRD_PC_FWD(PW)                                                        \
  return pc_fwd<PW>(n_points, points, neighbors, n_neighbors, cloud,          \
                    col_feats, radius, radius_all, min_nn, empty_feat, packed, \
                    rgb, save_c, save_h, save_y, st)
  if (pc_waves_fwd(n_points) == 8) XRD_PC_FWD(8);
  XRD_PC_FWD(4);
#undef XRD_PC_FWD
}



int xrd_point_color_bwd(int64_t n_points, const float* points,
                        const int64_t* neighbors, const int32_t* n_neighbors,
                        const float* cloud, const float* col_feats,
                        const float* radius, float radius_all, int min_nn,
                        const float* packed, const float* rgb,
                        const float* save_c, const float* save_h,
                        const float* save_y, const float* g_rgb,
                        float* g_points, float* g_col_feats, float* g_flat,
                        float* ops, float* workspace, xrd_stream_t stream) {
  using F = PcFlat;
  if (n_points < 0 || min_nn < 0) return XRD_ERR_ARG;
  if (g_flat != nullptr && (!ops || !workspace)) return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (g_flat != nullptr) {
    int rc = zero_floats(g_flat, F::N_GRAD, st);
    if (rc != XRD_OK) return rc;
  }
  if (n_points == 0) return XRD_OK;
  if (!points || !neighbors || !n_neighbors || !cloud || !col_feats ||
      !packed || !rgb || !save_c || !save_h || !save_y || !g_rgb)
    return XRD_ERR_ARG;
  const int64_t n = n_points;
  int rc;
  if (g_flat == nullptr) {      // tracking: pose (and feature) gradients only
#define XRD_PC_BWD(PW)                                                         \
  rc = pc_bwd<PW>(n, points, neighbors, n_neighbors, cloud, col_feats, radius, \
                  radius_all, min_nn, packed, rgb, save_c, save_h, save_y,     \
                  g_rgb, g_points, g_col_feats, st)
    if (pc_waves_bwd(n) == 8)
      XRD_PC_BWD(8);
    else
      XRD_PC_BWD(4);
#undef XRD_PC_BWD
    return rc;
  }
  static bool ready = false;
  if (!ready) {
    rc = pc_attr(point_color_bwd_w_kernel<false>, kWbLds);
    if (rc == XRD_OK) rc = pc_attr(point_color_bwd_w_kernel<true>, kWbLds);
    if (rc != XRD_OK) return rc;
    ready = true;
  }
  // one group of 128 points per block; more than kWbBlocks groups (one
  // partial each in the workspace) run as several launch pairs, the later
  // reductions adding to the flat gradient
  static const WbProds prods = wb_products();
  const int64_t groups = (n + WB_PTS - 1) / WB_PTS;
  for (int64_t g0 = 0; g0 < groups; g0 += kWbBlocks) {
    const int nb = (int)(groups - g0 < kWbBlocks ? groups - g0 : kWbBlocks);
#define XRD_PC_BWD_W(CLAMP)                                                     \
  hipLaunchKernelGGL(point_color_bwd_w_kernel<CLAMP>, dim3(nb),                \
                     dim3(WB_PW * 64), kWbLds, st, n, g0, points, neighbors,   \
                     n_neighbors, cloud, col_feats, radius, radius_all, min_nn, \
                     packed, rgb, save_c, save_h, save_y, g_rgb, g_points,     \
                     g_col_feats, g_flat, ops, workspace)
    if (n < WB_PTS)
      XRD_PC_BWD_W(true);
    else
      XRD_PC_BWD_W(false);
#undef XRD_PC_BWD_W
    hipLaunchKernelGGL(pc_dw_reduce_kernel, dim3(WbRec::N), dim3(1024), 0, st,
                       prods, nb, g0 > 0 ? 1 : 0, workspace, g_flat);
  }
  return check_launch("xrd_point_color_bwd (weights)");
}

}  // extern "C"
