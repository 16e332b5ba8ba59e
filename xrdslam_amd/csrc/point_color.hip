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
// relative-position features, neighbour distances), scatters the colour
// feature gradients with atomics (tiles transposed through LDS: 128-byte rows
// per instruction), and leaves the operands of the weight gradients in HBM for
// pc_dw_kernel (all 14 products in one launch over a job table).
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
// operands of the weight gradients, one workspace (floats per point: 4044)
struct PcOps {
  float *gz, *gh, *e40, *go, *fx, *fga, *fh, *fgy;
  __host__ __device__ PcOps(float* base, int64_t n) {
    gz = base;                  // [5][n][128] d loss / d pre-activation
    gh = gz + 5 * n * 128;      // [5][n][128] d loss / d layer output
    e40 = gh + 5 * n * 128;     // [n][40]     embedding of p
    go = e40 + n * 40;          // [n][4]      d loss / d output logit (3 + 0)
    fx = go + n * 4;            // [8n][52]    F_theta input [e_rel | f]
    fga = fx + 8 * n * 52;      // [8n][128]   d loss / d F_theta pre-activation
    fh = fga + 8 * n * 128;     // [8n][128]   F_theta hidden
    fgy = fh + 8 * n * 128;     // [8n][32]    d loss / d F_theta output
  }
};
constexpr int64_t kOpsPerPoint = 1280 + 44 + 8 * 340;

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
    float* __restrict__ g_feats, float* __restrict__ g_flat,
    float* __restrict__ ops_base) {
  using K = PcPack;
  float* wl = reinterpret_cast<float*>(pc_smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const bool want_w = ops_base != nullptr;
  const PcOps ops(ops_base, n);
  float* tail = wl + kBwdTail;
  for (int i = threadIdx.x; i < kBwdTailLen; i += blockDim.x)
    tail[i] = pk[K::OW + i];
  const float* ow = tail;
  const float* bemb = tail + (K::BEMB - K::OW);
  // d loss / d B_rel: lane (q, .) owns columns fidx(4s + q), s < 5
  float brel_acc[5][3];
#pragma unroll
  for (int s = 0; s < 5; ++s)
    brel_acc[s][0] = brel_acc[s][1] = brel_acc[s][2] = 0.f;
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
      if (want_w && q == 0)
        *reinterpret_cast<f32x4*>(ops.go + pt * 4) =
            f32x4{go[0], go[1], go[2], 0.f};
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
      if (want_w && valid) {
#pragma unroll
        for (int s = 0; s < 10; ++s) ops.e40[pt * 40 + 4 * s + q] = e[s];
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
      if (want_w && valid)
        save_rows<8>(ops.gh + (int64_t)i * n * 128, 128, pt, q, g_h);
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
      if (want_w && valid)
        save_rows<8>(ops.gz + (int64_t)i * n * 128, 128, pt, q, g_h);
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
      const int64_t row = pt * 8 + k;
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
          for (int t = 0; t < 4; ++t) {
            sp[jt][t] = softplus100_grad(a[jt][t]);
            a[jt][t] = softplus100(a[jt][t]);
          }
        if (want_w && valid) save_rows<8>(ops.fh, 128, row, q, a);
      }
      const float w = live ? uk / nb.den : 0.f;
      f32x4 g_y[2] = {g_c[0] * w, g_c[1] * w};
      if (want_w && valid) {
        save_rows<2>(ops.fgy, 32, row, q, g_y);
#pragma unroll
        for (int s = 0; s < 5; ++s) ops.fx[row * 52 + 4 * s + q] = e5[s];
        *reinterpret_cast<f32x4*>(ops.fx + row * 52 + 20 + 4 * q) = f[0];
        *reinterpret_cast<f32x4*>(ops.fx + row * 52 + 36 + 4 * q) = f[1];
      }
      f32x4 g_a[8];
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) g_a[jt] = z4;
      dense_h<8, 8, 8>(wl + kBwdRfOff + (K::W2T - K::RF), lane, 0, g_y, g_a);
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) g_a[jt] *= sp[jt];
      if (want_w && valid) save_rows<8>(ops.fga, 128, row, q, g_a);
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
        for (int a = 0; a < 3; ++a) {
          grel[a] = fmaf(garg, b[a], grel[a]);
          brel_acc[s][a] = fmaf(garg, rel[a], brel_acc[s][a]);
        }
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
  if (g_flat != nullptr) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int j = 4 * s + q, fi = j < 10 ? j : j - 10;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float v = row16_sum(brel_acc[s][a]);
        if (li == 0) atomicAdd(g_flat + PcFlat::BREL + a * 10 + fi, v);
      }
    }
  }
}

// ---- weight gradients: out[128][N] = G^T A over the rows ---------------------------
// G [rows][128], A [rows][N] (N <= 128).  A block of 4 waves owns chunks of 64
// rows and keeps its share of the product in MFMA accumulators (wave w: output
// rows 32w .. 32w+31 = two G tiles x NT column tiles, so that a fragment of A
// feeds two MFMAs); the chunk's rows are staged in LDS (row stride 144 floats:
// the four row groups of a fragment read fall on distinct banks).  The rows
// travel global -> registers -> LDS with the NEXT chunk's loads issued before
// this chunk's contraction, and inside the contraction the fragments of
// K-step s+1 are read before the MFMAs of step s (pinned asm reads,
// common.h); 74 KB of LDS and <= 256 registers: two blocks share a CU.
// Column sums of G and of A ride along (bias gradients).  All 14 products of a
// backward run as ONE launch over a job table (blocks of different jobs share
// the CUs), their per-block partials [128 * 16 NT + 128 + 16 NT] are summed
// into the flat gradient by one launch of pc_dw_reduce_kernel.
constexpr int DW_WAVES = 4, DW_THREADS = DW_WAVES * 64, DW_CHUNK = 64,
              DW_GS = 144, DW_AS = 144;
constexpr int DW_JOBS = 14;
constexpr int kDwLds = DW_CHUNK * (DW_GS + DW_AS) * (int)sizeof(float);
__host__ __device__ constexpr int dw_plen(int nt) {
  return 128 * 16 * nt + 128 + 16 * nt;
}

struct DwJob {
  const float *G, *A1, *A2;
  int64_t rows, ws_off;      // partials of this job: ws + ws_off
  int w1, w2, N, nt;
  int blk0, nblk;            // blocks [blk0, blk0 + nblk) of the contraction
  int red0;                  // first block of the reduction
  // destination: product element (o, c < N) -> w_off + o * ldo + c, or
  // (transposed) w_off + c * ldo + o; bias = column sums of G (128) or of A (N)
  int w_off, ldo, transposed, b_off, b_from_a;
};
struct DwJobs {
  DwJob j[DW_JOBS];
};

// fragments of one K-step: the wave's two G tiles and the NT column tiles
template <int NT>
struct DwFrag {
  float g[2], a[NT];
  template <int KOFF, int I>
  __device__ __forceinline__ void load_a(uint32_t aaddr) {
    if constexpr (I < NT) {
      a[I] = lds_async<KOFF + 64 * I>(aaddr);
      load_a<KOFF, I + 1>(aaddr);
    }
  }
  template <int KOFF>
  __device__ __forceinline__ void load(uint32_t gaddr, uint32_t aaddr) {
    g[0] = lds_async<KOFF>(gaddr);
    g[1] = lds_async<KOFF + 64>(gaddr);
    load_a<KOFF, 0>(aaddr);
  }
  template <int PENDING>
  __device__ __forceinline__ void landed() {
    lds_landed<PENDING>();
    lds_tie(g[0]);
    lds_tie(g[1]);
#pragma unroll
    for (int it = 0; it < NT; ++it) lds_tie(a[it]);
  }
  __device__ __forceinline__ void mma(f32x4 (*acc)[NT]) const {
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      acc[0][it] = XRD_MFMA4(g[0], a[it], acc[0][it]);
      acc[1][it] = XRD_MFMA4(g[1], a[it], acc[1][it]);
    }
  }
};

template <int NT>
__device__ __forceinline__ void dw_block(const DwJob& job, int blk,
                                         float* __restrict__ ws, float* Gs,
                                         float* As) {
  static_assert(DW_GS == DW_AS, "one K-step stride for both operands");
  constexpr int KSTEP = 4 * DW_GS * 4;          // bytes between K-steps
  const float* __restrict__ G = job.G;
  const float* __restrict__ A1 = job.A1;
  const float* __restrict__ A2 = job.A2;
  const int64_t rows = job.rows;
  const int w1 = job.w1, w2 = job.w2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = threadIdx.x & 127, grp = threadIdx.x >> 7;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[2][NT];
#pragma unroll
  for (int it = 0; it < NT; ++it) acc[0][it] = acc[1][it] = z4;
  float gsum = 0.f, asum = 0.f;
  const int64_t nchunks = (rows + DW_CHUNK - 1) / DW_CHUNK;
  const int k = lane >> 4, j = lane & 15;
  const int q1 = w1 >> 2, qq = q1 + (w2 >> 2);
  // (unconditional loads on clamped rows — no branch per load; rows beyond
  // the job's count become zeros when they are stored to LDS)
  constexpr int GQ = 8;               // G quads per thread: 64 x 32 / 256
  constexpr int AQ = NT;              // A quads per thread: 64 x 4 NT / 256
  f32x4 vg[GQ], va[AQ];
  // thread -> A quad: row i / (4 NT), quad i % (4 NT) of the 16 NT staged
  // columns; quads beyond the job's qq are staged as zeros (padding columns of
  // the last tile), so the tile needs no separate clearing pass
  constexpr int RQ = 4 * NT;
  const int64_t last = rows - 1;
  auto fetch = [&](int64_t ch) {
    const int64_t r0 = ch * DW_CHUNK;
#pragma unroll
    for (int u = 0; u < GQ; ++u) {
      const int i = threadIdx.x + u * DW_THREADS;
      int64_t row = r0 + (i >> 5);
      row = row < last ? row : last;
      vg[u] = *reinterpret_cast<const f32x4*>(G + row * 128 + ((i & 31) << 2));
    }
#pragma unroll
    for (int u = 0; u < AQ; ++u) {
      const int i = threadIdx.x + u * DW_THREADS;
      const int r = i / RQ;
      int c = i - r * RQ;
      c = c < qq ? c : qq - 1;
      int64_t row = r0 + r;
      row = row < last ? row : last;
      const float* __restrict__ src =
          c < q1 ? A1 + row * w1 + 4 * c : A2 + row * w2 + 4 * (c - q1);
      va[u] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  if (blk < nchunks) fetch(blk);
  for (int64_t ch = blk; ch < nchunks; ch += job.nblk) {
    const int64_t r0 = ch * DW_CHUNK;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GQ; ++u) {
      const int i = threadIdx.x + u * DW_THREADS;
      *reinterpret_cast<f32x4*>(Gs + (i >> 5) * DW_GS + ((i & 31) << 2)) =
          r0 + (i >> 5) < rows ? vg[u] : z4;
    }
#pragma unroll
    for (int u = 0; u < AQ; ++u) {
      const int i = threadIdx.x + u * DW_THREADS;
      const int r = i / RQ, c = i - r * RQ;
      *reinterpret_cast<f32x4*>(As + r * DW_AS + 4 * c) =
          c < qq && r0 + r < rows ? va[u] : z4;
    }
    __syncthreads();
    if (ch + job.nblk < nchunks) fetch(ch + job.nblk);
    {
      uint32_t gaddr = lds_addr(Gs + k * DW_GS + 32 * wave + j);
      uint32_t aaddr = lds_addr(As + k * DW_AS + j);
      DwFrag<NT> f0, f1;
      f0.template load<0>(gaddr, aaddr);
#pragma unroll 1
      for (int ks = 0; ks < DW_CHUNK / 4 - 2; ks += 2) {
        f1.template load<KSTEP>(gaddr, aaddr);
        f0.template landed<NT + 2>();     // f1's reads stay in flight
        f0.mma(acc);
        f0.template load<2 * KSTEP>(gaddr, aaddr);
        f1.template landed<NT + 2>();
        f1.mma(acc);
        gaddr += 2 * KSTEP;
        aaddr += 2 * KSTEP;
      }
      f1.template load<KSTEP>(gaddr, aaddr);
      f0.template landed<NT + 2>();
      f0.mma(acc);
      f1.template landed<0>();
      f1.mma(acc);
    }
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      gsum += Gs[(grp * 32 + r) * DW_GS + col];
      asum += As[(grp * 32 + r) * DW_AS + col];
    }
  }
  float* out = ws + job.ws_off + (int64_t)blk * dw_plen(NT);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 32 * wave + 16 * t + 4 * k + r;
#pragma unroll
      for (int it = 0; it < NT; ++it)
        out[o * (16 * NT) + 16 * it + j] = acc[t][it][r];
    }
  __syncthreads();
  float* R = Gs;   // [2][2][128]
  R[(0 * 2 + grp) * 128 + col] = gsum;
  R[(1 * 2 + grp) * 128 + col] = asum;
  __syncthreads();
  if (threadIdx.x < 128) {
    out[128 * 16 * NT + col] = R[0 * 128 + col] + R[1 * 128 + col];
    if (col < 16 * NT)
      out[128 * 16 * NT + 128 + col] = R[2 * 128 + col] + R[3 * 128 + col];
  }
}

__global__ __launch_bounds__(DW_THREADS, 2) void pc_dw_kernel(
    const DwJobs jobs, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Gs = reinterpret_cast<float*>(smem_raw);   // [64][144]
  float* As = Gs + DW_CHUNK * DW_GS;                 // [64][144]
  // (the 13 thresholds as independent scalar loads: one latency, not 13)
  int ji = 0;
#pragma unroll
  for (int t = 1; t < DW_JOBS; ++t) ji += (int)blockIdx.x >= jobs.j[t].blk0;
  const DwJob& job = jobs.j[ji];
  const int blk = blockIdx.x - job.blk0;
  switch (job.nt) {
    case 1: dw_block<1>(job, blk, ws, Gs, As); break;
    case 2: dw_block<2>(job, blk, ws, Gs, As); break;
    case 3: dw_block<3>(job, blk, ws, Gs, As); break;
    case 4: dw_block<4>(job, blk, ws, Gs, As); break;
    default: dw_block<8>(job, blk, ws, Gs, As); break;
  }
}

// flat[dst(i)] = sum over the job's blocks of partial[b][i]
__global__ __launch_bounds__(256) void pc_dw_reduce_kernel(
    const DwJobs jobs, const float* __restrict__ ws, float* __restrict__ flat) {
  __shared__ float red[4][64];
  int ji = 0;
#pragma unroll 1
  for (int t = 1; t < DW_JOBS; ++t)
    if ((int)blockIdx.x >= jobs.j[t].red0) ji = t;
  const DwJob& job = jobs.j[ji];
  const int nt = job.nt, N = job.N, n_blocks = job.nblk;
  const float* __restrict__ partial = ws + job.ws_off;
  const int plen = dw_plen(nt);
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = (blockIdx.x - job.red0) * 64 + c;
  float s0 = 0.f, s1 = 0.f;
  if (i < plen) {
    int b = grp;
    for (; b + 4 < n_blocks; b += 8) {
      s0 += partial[(int64_t)b * plen + i];
      s1 += partial[(int64_t)(b + 4) * plen + i];
    }
    for (; b < n_blocks; b += 4) s0 += partial[(int64_t)b * plen + i];
  }
  red[grp][c] = s0 + s1;
  __syncthreads();
  if (grp != 0 || i >= plen) return;
  const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  const int wl = 128 * 16 * nt;
  if (i < wl) {
    const int o = i / (16 * nt), cc = i - o * (16 * nt);
    if (cc < N)
      flat[job.transposed ? job.w_off + cc * job.ldo + o
                          : job.w_off + o * job.ldo + cc] = v;
  } else if (i < wl + 128) {
    if (!job.b_from_a && job.b_off >= 0) flat[job.b_off + (i - wl)] = v;
  } else {
    if (job.b_from_a && i - wl - 128 < N) flat[job.b_off + (i - wl - 128)] = v;
  }
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
  return n_points < 0 ? 0 : n_points * kOpsPerPoint;
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

// Blocks a job may use.  A block pays ~9 us before its first MFMA (dispatch,
// job record, first rows from HBM) and a partial of 128 x 16 NT floats at the
// end, so jobs get FEW, long blocks: the 12 products over n rows share one
// round of the 512 resident blocks (2 a CU), F_theta's two over 8 n rows the
// next, each in proportion to its cost per chunk (16 x 2 NT MFMAs + ~1.5 us of
// staging ~ NT + 4).
static int dw_cap(int nt, int64_t rows_per_point) {
  if (rows_per_point > 1) return nt == 2 ? 219 : 293;    // weights 6 : 8
  return nt == 8 ? 62 : nt == 3 ? 36 : nt == 2 ? 31 : 26;  // 12 : 7 : 6 : 5
}

struct DwPlan {
  DwJobs jobs;
  int n = 0, blocks = 0, red_blocks = 0;
  int64_t ws = 0;
  void add(int64_t rows, int cap, const float* G, const float* A1, int w1,
           const float* A2, int w2, int N, int nt, int w_off, int ldo,
           int transposed, int b_off, int b_from_a) {
    DwJob& j = jobs.j[n++];
    const int64_t nchunks = (rows + DW_CHUNK - 1) / DW_CHUNK;
    j.G = G; j.A1 = A1; j.A2 = A2; j.rows = rows; j.ws_off = ws;
    j.w1 = w1; j.w2 = w2; j.N = N; j.nt = nt;
    j.blk0 = blocks; j.nblk = (int)(nchunks < cap ? nchunks : cap);
    j.red0 = red_blocks;
    j.w_off = w_off; j.ldo = ldo; j.transposed = transposed;
    j.b_off = b_off; j.b_from_a = b_from_a;
    blocks += j.nblk;
    red_blocks += (dw_plen(nt) + 63) / 64;
    ws += (int64_t)j.nblk * dw_plen(nt);
  }
};

static void dw_plan(int64_t n, const float* save_c, const float* save_h,
                    float* ops, DwPlan& p) {
  using F = PcFlat;
  const PcOps op(ops, n);
  for (int i = 0; i < 5; ++i) {
    const float* gz = op.gz + (int64_t)i * n * 128;
    const float* hp = save_h + (int64_t)(i - 1) * n * 128;
    if (i == 0) {
      p.add(n, dw_cap(3, 1), gz, op.e40, 40, nullptr, 0, 40, 3, F::pw(0), 40, 0,
            F::pb(0), 0);
    } else if (i == 3) {
      // [e40 | h] -> 168 columns as two products (columns 0..39 with the bias,
      // columns 40..167 without: b_off -1)
      p.add(n, dw_cap(3, 1), gz, op.e40, 40, nullptr, 0, 40, 3, F::pw(3), 168,
            0, F::pb(3), 0);
      p.add(n, dw_cap(8, 1), gz, hp, 128, nullptr, 0, 128, 8, F::pw(3) + 40,
            168, 0, -1, 0);
    } else {
      p.add(n, dw_cap(8, 1), gz, hp, 128, nullptr, 0, 128, 8, F::pw(i), 128, 0,
            F::pb(i), 0);
    }
    p.add(n, dw_cap(2, 1), op.gh + (int64_t)i * n * 128, save_c, 32, nullptr, 0,
          32, 2, F::fcw(i), 32, 0, F::fcb(i), 0);
  }
  // output layer and F_theta's second layer: the 128-wide operand is G, the
  // product comes out transposed
  p.add(n, dw_cap(1, 1), save_h + 4 * n * 128, op.go, 4, nullptr, 0, 3, 1,
        F::OW, 128, 1, F::OB, 1);
  p.add(8 * n, dw_cap(2, 8), op.fh, op.fgy, 32, nullptr, 0, 32, 2, F::W2, 128,
        1, F::B2, 1);
  p.add(8 * n, dw_cap(4, 8), op.fga, op.fx, 52, nullptr, 0, 52, 4, F::W1, 52, 0,
        F::B1, 0);
}

template <int PW>
static int pc_bwd(int64_t n, const float* points, const int64_t* neighbors,
                  const int32_t* n_neighbors, const float* cloud,
                  const float* col_feats, const float* radius,
                  float radius_all, int min_nn, const float* packed,
                  const float* rgb, const float* save_c, const float* save_h,
                  const float* save_y, const float* g_rgb, float* g_points,
                  float* g_feats, float* g_flat, float* ops, hipStream_t st) {
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
                     save_h, save_y, g_rgb, g_points, g_feats, g_flat, ops);
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
  DwPlan p;   // the bound: every job at its block cap
  dw_plan((int64_t)1 << 20, nullptr, nullptr, nullptr, p);
  return p.ws;
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
  float* o = g_flat != nullptr ? ops : nullptr;
  int rc;
#define XRD_PC_BWD(PW)                                                         \
  rc = pc_bwd<PW>(n, points, neighbors, n_neighbors, cloud, col_feats, radius, \
                  radius_all, min_nn, packed, rgb, save_c, save_h, save_y,     \
                  g_rgb, g_points, g_col_feats, g_flat, o, st)
  if (pc_waves_bwd(n) == 8)
    XRD_PC_BWD(8);
  else
    XRD_PC_BWD(4);
#undef XRD_PC_BWD
  if (rc != XRD_OK || g_flat == nullptr) return rc;
  static bool ready = false;
  if (!ready) {
    rc = pc_attr(pc_dw_kernel, kDwLds);
    if (rc != XRD_OK) return rc;
    ready = true;
  }
  DwPlan plan;
  dw_plan(n, save_c, save_h, ops, plan);
  hipLaunchKernelGGL(pc_dw_kernel, dim3(plan.blocks), dim3(DW_THREADS),
                     kDwLds, st, plan.jobs, workspace);
  hipLaunchKernelGGL(pc_dw_reduce_kernel, dim3(plan.red_blocks), dim3(256), 0,
                     st, plan.jobs, workspace, g_flat);
  return check_launch("xrd_point_color_bwd (weights)");
}

}  // extern "C"
