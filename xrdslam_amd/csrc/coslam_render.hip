// Fused Co-SLAM ray renderer for gfx950: depth-guided z sampling, float64
// bound normalisation, multi-resolution hash grid + OneBlob encodings, the two
// bias-free 2x32 MLPs on v_mfma_f32_16x16x4_f32, SDF-to-weight compositing,
// forward and backward, replacing JointEncoding.render_rays and everything
// below it (slam/models/joint_encoding.py:250-344, 346-407, 463-507;
// slam/model_components/decoder_coslam.py; tiny-cuda-nn encodings per
// SURVEY.md App. C.1/C.2).
//
// One wave = one tile of 16 samples of one ray: lane (j = l&15, q = l>>4)
// works on sample j and owns hash levels q, q+4, q+8, q+12 and OneBlob bins
// 4q..4q+3 — all 64 lanes gather distinct table entries, and the features
// land directly in the MFMA B-operand layout of the first layer
// (coslam_layout.h).  The kernels are gather/atomic bound (16 levels x 8
// corners x 8 B per sample); the MLPs are ~5 kFLOP/sample.
//
// Forward: block = one ray (ceil(S/16) waves); samples are sorted/jittered in
// LDS and composited by wave 0.  Backward: persistent waves loop over tiles,
// every wave re-derives the per-ray compositing gradient from the saved
// raw/z_vals (no block-level sync), weight gradients accumulate in MFMA
// accumulators across tiles (operands transposed through wave-private LDS) and
// are reduced once per block.
#include "common.h"
#include "coslam_layout.h"

namespace xrd {
namespace {

using namespace cs;
typedef xrd_coslam_scene Scene;

constexpr int kMaxS = 48;
constexpr int kRow = 20;              // staged row: 16 samples + pad (16-B aligned)
constexpr int kStage = 7 * 16 * kRow; // floats of wave-private staging (7 tiles)
constexpr int kBwdWaves = 4;
constexpr int kBwdMaxBlocks = 512;

#define WF(base, idx) pack[((base) + (idx)) * 64 + lane]
#define CS_SB __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ float sigmoidf_(float x) {
  return 1.f / (1.f + expf(-x));
}

// ---- hash grid (tcnn grid_index: dense strides while they fit, else the
// coherent prime hash; modulo the level size) ------------------------------
__device__ __forceinline__ uint32_t level_index(bool dense, uint32_t res,
                                                uint32_t size, uint32_t cx,
                                                uint32_t cy, uint32_t cz) {
  if (dense) {
    // samples beyond the bound have coordinates outside [0,1] (negative
    // floors wrap in uint32 exactly as in tiny-cuda-nn): a true modulo
    return (cx + cy * res + cz * res * res) % size;
  }
  // hashed levels have size = 2^log2_hashmap_size
  return ((cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u)) & (size - 1);
}

// per-level table staged in LDS: lanes index it by their own level (a
// lane-varying index into the by-value kernel argument would force the
// argument struct into scratch)
struct LevelTab {
  float scale[XRD_COSLAM_LEVELS];
  uint32_t res[XRD_COSLAM_LEVELS];
  uint32_t size[XRD_COSLAM_LEVELS];   // top bit: dense level
  uint32_t offset[XRD_COSLAM_LEVELS];
};
__device__ __forceinline__ void load_level_tab(const Scene& sc, LevelTab& lt,
                                               int tid) {
  if (tid < XRD_COSLAM_LEVELS) {
    const uint32_t res = sc.lv_res[tid], size = sc.lv_size[tid];
    const bool dense = (uint64_t)res * res * res <= (uint64_t)size;
    lt.scale[tid] = sc.lv_scale[tid];
    lt.res[tid] = res;
    lt.size[tid] = size | (dense ? 0x80000000u : 0u);
    lt.offset[tid] = sc.lv_offset[tid];
  }
}

struct LevelPos {
  uint32_t cx, cy, cz;
  float wx, wy, wz;
};
__device__ __forceinline__ LevelPos level_pos(float scale, float x, float y,
                                              float z) {
  const float fx = fmaf(scale, x, 0.5f), fy = fmaf(scale, y, 0.5f),
              fz = fmaf(scale, z, 0.5f);
  const float ffx = floorf(fx), ffy = floorf(fy), ffz = floorf(fz);
  LevelPos p;
  p.cx = (uint32_t)(int)ffx;
  p.cy = (uint32_t)(int)ffy;
  p.cz = (uint32_t)(int)ffz;
  p.wx = fx - ffx;
  p.wy = fy - ffy;
  p.wz = fz - ffz;
  return p;
}

__device__ __forceinline__ void hash_level_fwd(const LevelTab& lt,
                                               const float* table, int lvl,
                                               float x, float y, float z,
                                               float& f0, float& f1) {
  const uint32_t res = lt.res[lvl], szf = lt.size[lvl];
  const uint32_t size = szf & 0x7fffffffu;
  const bool dense = (szf >> 31) != 0;
  const float2* tab = reinterpret_cast<const float2*>(table) + lt.offset[lvl];
  const LevelPos p = level_pos(lt.scale[lvl], x, y, z);
  float2 v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
    v[c] = tab[level_index(dense, res, size, p.cx + (c & 1),
                           p.cy + ((c >> 1) & 1), p.cz + (c >> 2))];
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float ax = (c & 1) ? p.wx : 1.f - p.wx,
                ay = (c & 2) ? p.wy : 1.f - p.wy,
                az = (c & 4) ? p.wz : 1.f - p.wz;
    const float w = ax * ay * az;
    a0 = fmaf(w, v[c].x, a0);
    a1 = fmaf(w, v[c].y, a1);
  }
  f0 = a0;
  f1 = a1;
}

// d(loss)/d(position) through one level: sum over corners of the trilinear
// weight derivatives times <entry, feature gradient>
__device__ __forceinline__ void hash_level_dx(const LevelTab& lt,
                                              const float* table, int lvl,
                                              float x, float y, float z,
                                              float g0, float g1, float& dx,
                                              float& dy, float& dz) {
  const uint32_t res = lt.res[lvl], szf = lt.size[lvl];
  const uint32_t size = szf & 0x7fffffffu;
  const bool dense = (szf >> 31) != 0;
  const uint32_t off = lt.offset[lvl];
  const float scale = lt.scale[lvl];
  const LevelPos p = level_pos(scale, x, y, z);
  float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const uint32_t idx = off + level_index(dense, res, size, p.cx + (c & 1),
                                           p.cy + ((c >> 1) & 1),
                                           p.cz + (c >> 2));
    const float ax = (c & 1) ? p.wx : 1.f - p.wx,
                ay = (c & 2) ? p.wy : 1.f - p.wy,
                az = (c & 4) ? p.wz : 1.f - p.wz;
    const float2 v = reinterpret_cast<const float2*>(table)[idx];
    const float dv = v.x * g0 + v.y * g1;
    sx += ((c & 1) ? 1.f : -1.f) * ay * az * dv;
    sy += ((c & 2) ? 1.f : -1.f) * ax * az * dv;
    sz += ((c & 4) ? 1.f : -1.f) * ax * ay * dv;
  }
  dx = fmaf(scale, sx, dx);
  dy = fmaf(scale, sy, dy);
  dz = fmaf(scale, sz, dz);
}

// ---- OneBlob (quartic kernel, 16 bins) ------------------------------------
constexpr float kBins = 16.f;
__device__ __forceinline__ float quartic_cdf(float x) {
  const float u = x * kBins, u2 = u * u, u4 = u2 * u2;
  return fminf(fmaxf((15.f / 16.f) * u * (1.f - (2.f / 3.f) * u2 +
                                          (1.f / 5.f) * u4) + 0.5f, 0.f), 1.f);
}
__device__ __forceinline__ float quartic_pdf(float x) {
  const float u = x * kBins;
  if (fabsf(u) >= 1.f) return 0.f;
  const float t = 1.f - u * u;
  return (15.f / 16.f) * t * t * kBins;
}
__device__ __forceinline__ float cdf3(float x) {
  return quartic_cdf(x) + quartic_cdf(x - 1.f) + quartic_cdf(x + 1.f);
}
__device__ __forceinline__ float pdf3(float x) {
  return quartic_pdf(x) + quartic_pdf(x - 1.f) + quartic_pdf(x + 1.f);
}
// bins 4q..4q+3 of one coordinate
__device__ __forceinline__ void oneblob_fwd(float v, int q, float* f) {
  float left = cdf3((float)(4 * q) / kBins - v);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float right = cdf3((float)(4 * q + r + 1) / kBins - v);
    f[r] = right - left;
    left = right;
  }
}
__device__ __forceinline__ float oneblob_bwd(float v, int q, const float* g) {
  float left = pdf3((float)(4 * q) / kBins - v);
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float right = pdf3((float)(4 * q + r + 1) / kBins - v);
    acc += g[r] * (-(right - left));
    left = right;
  }
  return acc;
}

// ---- encodings of one sample for lane group q -> X[20] (slot order) ----------
// SEQ: levels in two batches (bounds the gathers in flight, i.e. the register
// peak, in the backward kernel whose weight-gradient accumulators stay live)
template <bool SEQ>
__device__ __forceinline__ void encode(const LevelTab& lt, const float* table,
                                       int q, float x, float y, float z,
                                       float* X) {
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    hash_level_fwd(lt, table, q + 4 * a, x, y, z, X[2 * a], X[2 * a + 1]);
    if (SEQ && (a & 1)) CS_SB;
  }
  oneblob_fwd(x, q, X + 8);
  oneblob_fwd(y, q, X + 12);
  oneblob_fwd(z, q, X + 16);
}

// float64 normalisation of run_network (joint_encoding.py:483-507): the
// bounding box is a float64 tensor, so (p - lo) / (hi - lo) is evaluated in
// double before tiny-cuda-nn casts its input to float
__device__ __forceinline__ float normalise(const Scene& sc, int d, float p) {
  return (float)(((double)p - sc.bound[2 * d]) /
                 (sc.bound[2 * d + 1] - sc.bound[2 * d]));
}

struct Act {
  float X[20];
  f32x4 h1[2];
  f32x4 h2;
  f32x4 hc[2];
  f32x4 out;
};

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = fmaxf(v[i], 0.f);
  return r;
}

__device__ __forceinline__ void mlp_forward(const float* __restrict__ pack,
                                            int lane, Act& A) {
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
  for (int s = 0; s < 20; ++s) {
    a0 = XRD_MFMA4(WF(kF0, s), A.X[s], a0);
    a1 = XRD_MFMA4(WF(kF0, 20 + s), A.X[s], a1);
  }
  A.h1[0] = relu4(a0);
  A.h1[1] = relu4(a1);
  CS_SB;
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) b = XRD_MFMA4(WF(kF1, s), A.h1[s >> 2][s & 3], b);
  A.h2 = b;
  CS_SB;
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float xin = s < 12 ? A.X[8 + s] : A.h2[s - 12];
    c0 = XRD_MFMA4(WF(kF2, s), xin, c0);
    c1 = XRD_MFMA4(WF(kF2, 16 + s), xin, c1);
  }
  A.hc[0] = relu4(c0);
  A.hc[1] = relu4(c1);
  CS_SB;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) o = XRD_MFMA4(WF(kF3, s), A.hc[s >> 2][s & 3], o);
  A.out = o;
  CS_SB;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(192) void coslam_fwd_kernel(
    Scene sc, int n_rays, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ target_d,
    const float* __restrict__ rnd, float* __restrict__ z_out,
    float* __restrict__ raw_out, float* __restrict__ maps) {
  __shared__ float zc[kMaxS], zs[kMaxS], zj[kMaxS];
  __shared__ float rawS[kMaxS][4];
  __shared__ LevelTab lt;
  const int ray = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = sc.n_range_d + sc.n_sample_d;
  load_level_tab(sc, lt, tid);
  // -- candidates: cat([uniform, near-depth]) (joint_encoding.py:262-281)
  if (tid < S) {
    const float d = target_d[ray];
    float z;
    if (tid < sc.n_sample_d) {
      z = sc.t_uniform[tid];
    } else {
      const int k = tid - sc.n_sample_d;
      z = d <= 0.f ? sc.t_far[k] : __fadd_rn(sc.t_near[k], d);
    }
    zc[tid] = z;
  }
  __syncthreads();
  if (tid < S) {  // rank sort (ties by index: values equal anyway)
    const float z = zc[tid];
    int rank = 0;
    for (int k = 0; k < S; ++k) {
      const float o = zc[k];
      rank += (o < z || (o == z && k < tid)) ? 1 : 0;
    }
    zs[rank] = z;
  }
  __syncthreads();
  if (tid < S) {  // stratified jitter (:289-298)
    float z = zs[tid];
    if (sc.perturb) {
      const float lo = tid == 0 ? z : __fmul_rn(.5f, __fadd_rn(z, zs[tid - 1]));
      const float hi = tid == S - 1 ? z
                                    : __fmul_rn(.5f, __fadd_rn(zs[tid + 1], z));
      z = __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), rnd[(size_t)ray * S + tid]));
    }
    zj[tid] = z;
    z_out[(size_t)ray * S + tid] = z;
  }
  __syncthreads();
  // -- this wave's tile
  const int j = lane & 15, q = lane >> 4;
  const int smp = wave * 16 + j;
  const bool live = smp < S;
  const float z = zj[live ? smp : S - 1];
  float xn[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float p = __fadd_rn(rays_o[ray * 3 + d],
                              __fmul_rn(rays_d[ray * 3 + d], z));
    xn[d] = normalise(sc, d, p);
  }
  Act A;
  encode<false>(lt, sc.table, q, xn[0], xn[1], xn[2], A.X);
  mlp_forward(sc.pack, lane, A);
  if (q == 0 && live) {
    const float4 r = make_float4(A.out[0], A.out[1], A.out[2], A.h2[0]);
    *reinterpret_cast<float4*>(&rawS[smp][0]) = r;
    *reinterpret_cast<float4*>(raw_out + ((size_t)ray * S + smp) * 4) = r;
  }
  __syncthreads();
  if (wave != 0) return;
  // -- sdf2weights + raw2outputs (:346-407)
  const bool v = lane < S;
  const int li = v ? lane : S - 1;
  const float s = rawS[li][3], zz = zj[li];
  const float u = s / sc.trunc;
  const float a = sigmoidf_(u) * sigmoidf_(-u);
  const float sn = __shfl_down(s, 1);
  const bool cr = v && lane < S - 1 && sn * s < 0.f;
  const unsigned long long bal = __ballot(cr);
  const int ind = bal ? __ffsll((long long)bal) - 1 : 0;
  const float zmin = zj[ind];
  const float band = (float)((double)sc.sc_factor * (double)sc.trunc);
  const float wu = (v && zz < zmin + band) ? a : 0.f;
  const float W = wave_sum(wu) + 1e-8f;
  const float w = wu / W;
  const float r0 = wave_sum(w * sigmoidf_(rawS[li][0]));
  const float r1 = wave_sum(w * sigmoidf_(rawS[li][1]));
  const float r2 = wave_sum(w * sigmoidf_(rawS[li][2]));
  const float depth = wave_sum(w * zz);
  const float dz = zz - depth;
  const float var = wave_sum(w * (dz * dz));
  const float acc = wave_sum(w);
  if (lane == 0) {
    const float bg = sc.white_bkgd ? 1.f - acc : 0.f;
    float* m = maps + (size_t)ray * 8;
    *reinterpret_cast<float4*>(m) = make_float4(r0 + bg, r1 + bg, r2 + bg, depth);
    *reinterpret_cast<float4*>(m + 4) =
        make_float4(var, acc, 1.f / fmaxf(1e-10f, depth / acc), 0.f);
  }
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
// gradient wrt raw[lane] (rgb logits, sdf) of one ray; every wave that works
// on a tile of the ray recomputes it from the saved forward outputs
__device__ __forceinline__ void composite_bwd(const Scene& sc, int S, int lane,
                                              const float* __restrict__ raw_ray,
                                              const float* __restrict__ z_ray,
                                              const float* __restrict__ gm,
                                              const float* __restrict__ graw_ray,
                                              float& z_lane, float* dr) {
  const bool v = lane < S;
  const int li = v ? lane : S - 1;
  const float4 rw = *reinterpret_cast<const float4*>(raw_ray + li * 4);
  const float zz = z_ray[li];
  z_lane = zz;
  const float s = rw.w;
  const float u = s / sc.trunc;
  const float sg = sigmoidf_(u), sm = sigmoidf_(-u);
  const float a = sg * sm;
  const float sn = __shfl_down(s, 1);
  const bool cr = v && lane < S - 1 && sn * s < 0.f;
  const unsigned long long bal = __ballot(cr);
  const int ind = bal ? __ffsll((long long)bal) - 1 : 0;
  const float zmin = __shfl(zz, ind);
  const float band = (float)((double)sc.sc_factor * (double)sc.trunc);
  const bool msk = v && zz < zmin + band;
  const float wu = msk ? a : 0.f;
  const float W = wave_sum(wu) + 1e-8f;
  const float w = wu / W;
  const float depth = wave_sum(w * zz);
  const float dzv = zz - depth;
  const float4 g0 = *reinterpret_cast<const float4*>(gm);      // rgb, depth
  const float4 g1 = *reinterpret_cast<const float4*>(gm + 4);  // var, acc
  float g_acc = g1.y;
  if (sc.white_bkgd) g_acc -= g0.x + g0.y + g0.z;
  const float gd = g0.w + g1.x * (-2.f) * wave_sum(w * dzv);
  const float c0 = sigmoidf_(rw.x), c1 = sigmoidf_(rw.y), c2 = sigmoidf_(rw.z);
  const float dLdw = g0.x * c0 + g0.y * c1 + g0.z * c2 + gd * zz +
                     g1.x * dzv * dzv + g_acc;
  const float T = wave_sum(w * dLdw);
  const float dwu = (dLdw - T) / W;
  float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
  if (graw_ray != nullptr && v)
    gr = *reinterpret_cast<const float4*>(graw_ray + li * 4);
  dr[0] = v ? g0.x * w * c0 * (1.f - c0) + gr.x : 0.f;
  dr[1] = v ? g0.y * w * c1 * (1.f - c1) + gr.y : 0.f;
  dr[2] = v ? g0.z * w * c2 * (1.f - c2) + gr.z : 0.f;
  dr[3] = (msk ? dwu * a * (sm - sg) / sc.trunc : 0.f) + gr.w;
}

// staging of one D-layout tile: T[slot 4q+r][sample j]
__device__ __forceinline__ void stage4(float* T, int tile, int j, int q,
                                       float v0, float v1, float v2, float v3) {
  float* p = T + (tile * 16 + 4 * q) * kRow + j;
  p[0] = v0;
  p[kRow] = v1;
  p[2 * kRow] = v2;
  p[3 * kRow] = v3;
}
// acc[m][n] += G_m^T X_n over the 16 samples: A[i][k_t] = G[slot i][sample
// 4k+t], B[k_t][j] = X[slot j][sample 4k+t]
template <int MT, int NTL>
__device__ __forceinline__ void dw_accum(const float* T, int i, int k,
                                         f32x4 (*acc)[NTL]) {
  f32x4 a[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
    a[m] = *reinterpret_cast<const f32x4*>(T + (m * 16 + i) * kRow + 4 * k);
#pragma unroll
  for (int n = 0; n < NTL; ++n) {
    const f32x4 b =
        *reinterpret_cast<const f32x4*>(T + ((MT + n) * 16 + i) * kRow + 4 * k);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][n] = XRD_MFMA4(a[m][t], b[t], acc[m][n]);
  }
}

template <int MT, int NTL>
__device__ __forceinline__ void dw_flush(float* red, int base, int in_dim, int j,
                                         int q, const f32x4 (*acc)[NTL]) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTL; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(red + base + (16 * m + 4 * q + r) * in_dim + 16 * n + j,
                  acc[m][n][r]);
}

// (the table-gradient variant keeps 30 weight-gradient accumulators a wave:
// 250 VGPRs + 184 AGPRs, one wave a SIMD)
template <bool DP, bool DG>
__global__ __launch_bounds__(kBwdWaves * 64, DG ? 1 : 2) void coslam_bwd_kernel(
    Scene sc, int n_rays, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ z_vals,
    const float* __restrict__ raw, const float* __restrict__ g_maps,
    const float* __restrict__ g_raw, float* __restrict__ g_o,
    float* __restrict__ g_d, float* __restrict__ partials,
    float* __restrict__ xs, float* __restrict__ dfeat, int64_t n_extra) {
  __shared__ __attribute__((aligned(16))) float lds[kBwdWaves * kStage];
  __shared__ LevelTab lt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  load_level_tab(sc, lt, tid);
  __syncthreads();
  const int j = lane & 15, q = lane >> 4;
  const int S = sc.n_range_d + sc.n_sample_d;
  const int NT = (S + 15) >> 4;
  const int tiles = n_rays * NT;
  // (level stride of dfeat: the samples + the extra points appended behind
  // them for the merged table scatter)
  const size_t n_samples = (size_t)n_rays * S + (size_t)n_extra;
  constexpr bool DW = DG;
  float* T = lds + wave * kStage;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 dw0[2][5], dw1[1][2], dw2[2][4], dw3[1][2];
  if (DG) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int n = 0; n < 5; ++n) dw0[m][n] = z4;
#pragma unroll
      for (int n = 0; n < 4; ++n) dw2[m][n] = z4;
    }
    dw1[0][0] = dw1[0][1] = dw3[0][0] = dw3[0][1] = z4;
  }
  for (int tile = blockIdx.x * kBwdWaves + wave; tile < tiles;
       tile += gridDim.x * kBwdWaves) {
    const int ray = tile / NT, t = tile - ray * NT;
    // opaque per iteration: keeps the 176 fragment addresses from being
    // hoisted out of the tile loop as 64-bit per-lane pointers (spills)
    const float* pack = sc.pack;
    asm volatile("" : "+s"(pack));
    float z_lane, dr[4];
    composite_bwd(sc, S, lane, raw + (size_t)ray * S * 4,
                  z_vals + (size_t)ray * S, g_maps + (size_t)ray * 8,
                  g_raw ? g_raw + (size_t)ray * S * 4 : nullptr, z_lane, dr);
    const int smp = t * 16 + j;
    const bool live = smp < S;
    const int src = live ? smp : S - 1;
    const float z = __shfl(z_lane, src);
    float gc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float g = __shfl(dr[c], src);
      gc[c] = live ? g : 0.f;
    }
    // tiles whose samples all lie behind the truncation band of their ray
    // carry exactly zero gradient: nothing to do
    if (!__any(gc[0] != 0.f || gc[1] != 0.f || gc[2] != 0.f || gc[3] != 0.f)) {
      if (DG && live) {
        const size_t smp_g = (size_t)ray * S + smp;
#pragma unroll
        for (int a = 0; a < 4; ++a)
          *reinterpret_cast<float2*>(dfeat + ((size_t)(q + 4 * a) * n_samples +
                                              smp_g) * 2) = make_float2(0.f, 0.f);
      }
      continue;
    }
    float xn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float p = __fadd_rn(rays_o[ray * 3 + d],
                                __fmul_rn(rays_d[ray * 3 + d], z));
      xn[d] = normalise(sc, d, p);
    }
    Act A;
    encode<DG>(lt, sc.table, q, xn[0], xn[1], xn[2], A.X);
    CS_SB;
    mlp_forward(pack, lane, A);

    // Each layer's weight gradient is accumulated as soon as its output
    // gradient exists, so that activations die early (register pressure).
    // ---- colour layer 2: slots 0..2 (lane group 0) carry d rgb-logits
    f32x4 g3;
#pragma unroll
    for (int r = 0; r < 4; ++r) g3[r] = (q == 0 && r < 3) ? gc[r] : 0.f;
    if (DW) {  // G = g3 (1 tile), X = hc (2 tiles)
      stage4(T, 0, j, q, g3[0], g3[1], g3[2], g3[3]);
      stage4(T, 1, j, q, A.hc[0][0], A.hc[0][1], A.hc[0][2], A.hc[0][3]);
      stage4(T, 2, j, q, A.hc[1][0], A.hc[1][1], A.hc[1][2], A.hc[1][3]);
      wave_lds_sync();
      dw_accum<1, 2>(T, j, q, dw3);
      wave_lds_sync();
    }
    f32x4 dhc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x4 acc = z4;
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = XRD_MFMA4(WF(kB3, m * 4 + s), g3[s], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = A.hc[m][r] > 0.f ? acc[r] : 0.f;
      dhc[m] = acc;
    }
    CS_SB;
    if (DW) {  // colour layer 1: G = dhc (2), X = OneBlob (3) + h2 (1)
      stage4(T, 0, j, q, dhc[0][0], dhc[0][1], dhc[0][2], dhc[0][3]);
      stage4(T, 1, j, q, dhc[1][0], dhc[1][1], dhc[1][2], dhc[1][3]);
#pragma unroll
      for (int n = 0; n < 3; ++n)
        stage4(T, 2 + n, j, q, A.X[8 + 4 * n], A.X[9 + 4 * n], A.X[10 + 4 * n],
               A.X[11 + 4 * n]);
      stage4(T, 5, j, q, A.h2[0], A.h2[1], A.h2[2], A.h2[3]);
      wave_lds_sync();
      dw_accum<2, 4>(T, j, q, dw2);
      wave_lds_sync();
    }
    // ---- colour layer 1: -> d OneBlob (tiles 0..2), d (sdf|geo) (tile 3)
    f32x4 dxc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f32x4 acc = z4;
#pragma unroll
      for (int s = 0; s < 8; ++s)
        acc = XRD_MFMA4(WF(kB2, m * 8 + s), dhc[s >> 2][s & 3], acc);
      dxc[m] = acc;
      CS_SB;
    }
    f32x4 gh2 = dxc[3];
    if (q == 0) gh2[0] += gc[3];  // slot 0 = sdf
    if (DW) {  // sdf layer 2: G = gh2 (1), X = h1 (2)
      stage4(T, 0, j, q, gh2[0], gh2[1], gh2[2], gh2[3]);
      stage4(T, 1, j, q, A.h1[0][0], A.h1[0][1], A.h1[0][2], A.h1[0][3]);
      stage4(T, 2, j, q, A.h1[1][0], A.h1[1][1], A.h1[1][2], A.h1[1][3]);
      wave_lds_sync();
      dw_accum<1, 2>(T, j, q, dw1);
      wave_lds_sync();
    }
    // ---- sdf layer 2
    f32x4 dh1[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x4 acc = z4;
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = XRD_MFMA4(WF(kB1, m * 4 + s), gh2[s], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = A.h1[m][r] > 0.f ? acc[r] : 0.f;
      dh1[m] = acc;
    }
    CS_SB;
    if (DW) {  // sdf layer 1: G = dh1 (2 tiles), X = X0 (5 tiles)
      stage4(T, 0, j, q, dh1[0][0], dh1[0][1], dh1[0][2], dh1[0][3]);
      stage4(T, 1, j, q, dh1[1][0], dh1[1][1], dh1[1][2], dh1[1][3]);
#pragma unroll
      for (int n = 0; n < 5; ++n)
        stage4(T, 2 + n, j, q, A.X[4 * n], A.X[4 * n + 1], A.X[4 * n + 2],
               A.X[4 * n + 3]);
      wave_lds_sync();
      dw_accum<2, 5>(T, j, q, dw0);
      wave_lds_sync();
    }
    // ---- sdf layer 1: -> d hash (tiles 0,1), d OneBlob (tiles 2..4)
    f32x4 dx0[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      f32x4 acc = z4;
#pragma unroll
      for (int s = 0; s < 8; ++s)
        acc = XRD_MFMA4(WF(kB0, m * 8 + s), dh1[s >> 2][s & 3], acc);
      dx0[m] = acc;
      CS_SB;
    }
    // ---- encodings backward
    float dpx = 0.f, dpy = 0.f, dpz = 0.f;
    if (DP) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        hash_level_dx(lt, sc.table, q + 4 * a, xn[0], xn[1], xn[2],
                      dx0[a >> 1][2 * (a & 1)], dx0[a >> 1][2 * (a & 1) + 1],
                      dpx, dpy, dpz);
        CS_SB;
      }
    }
    if (DG && live) {
      // hash-feature gradients, level-major [level][sample] float2, for the
      // chunked LDS scatter that follows this kernel (no global atomics)
      const size_t smp_g = (size_t)ray * S + smp;
#pragma unroll
      for (int a = 0; a < 4; ++a)
        *reinterpret_cast<float2*>(dfeat + ((size_t)(q + 4 * a) * n_samples +
                                            smp_g) * 2) =
            make_float2(dx0[a >> 1][2 * (a & 1)], dx0[a >> 1][2 * (a & 1) + 1]);
      if (q == 0) {
        xs[smp_g * 3 + 0] = xn[0];
        xs[smp_g * 3 + 1] = xn[1];
        xs[smp_g * 3 + 2] = xn[2];
      }
    }
    if (DP) {
      float gb[4];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int r = 0; r < 4; ++r) gb[r] = dx0[2 + d][r] + dxc[d][r];
        const float g = oneblob_bwd(xn[d], q, gb);
        if (d == 0) dpx += g;
        if (d == 1) dpy += g;
        if (d == 2) dpz += g;
      }
      float dp[3] = {dpx, dpy, dpz};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float g = group4_sum(dp[d]);
        g = (float)((double)g / (sc.bound[2 * d + 1] - sc.bound[2 * d]));
        g = live ? g : 0.f;
        const float so = row16_sum(g), sd = row16_sum(g * z);
        if (lane == 0) {
          atomicAdd(g_o + ray * 3 + d, so);
          atomicAdd(g_d + ray * 3 + d, sd);
        }
      }
    }
  }
  if (DG) {
    // block reduction of the weight-gradient accumulators, then one partial
    // row per block
    __syncthreads();
    for (int i = tid; i < kDwLen; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    dw_flush<2, 5>(lds, kD0, 80, j, q, dw0);
    dw_flush<1, 2>(lds, kD1, 32, j, q, dw1);
    dw_flush<2, 4>(lds, kD2, 64, j, q, dw2);
    dw_flush<1, 2>(lds, kD3, 32, j, q, dw3);
    __syncthreads();
    float* out = partials + (size_t)blockIdx.x * kDwLen;
    for (int i = tid; i < kDwLen; i += blockDim.x) out[i] = lds[i];
  }
}

// g_dw[i] += sum over a chunk of block partials (g_dw zeroed beforehand)
__global__ __launch_bounds__(256) void coslam_reduce_kernel(
    const float* __restrict__ partials, int n_blocks, int chunk,
    float* __restrict__ g_dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kDwLen) return;
  const int b0 = blockIdx.y * chunk, b1 = min(n_blocks, b0 + chunk);
  float s = 0.f;
  for (int b = b0; b < b1; ++b) s += partials[(size_t)b * kDwLen + i];
  if (b1 > b0) atomicAdd(g_dw + i, s);
}

// ---------------------------------------------------------------------------
// loss of JointEncoding.get_loss_dict without the smoothness term
// (joint_encoding.py:94-147, model_components/utils.py get_sdf_loss/get_masks)
// ---------------------------------------------------------------------------
struct LossCfg {
  float w_rgb, w_depth, w_sdf, w_fs;
  float trunc;        // training_trunc * data_sc_factor
  float depth_trunc;  // cam_depth_trunc
  float rgb_missing;  // weight of colour on invalid-depth pixels
  int S;
};
// per-ray statistics: n_fs, n_sdf, S_fs, S_sdf, valid, depth err^2, rgb err^2
__global__ __launch_bounds__(256) void coslam_loss_stats_kernel(
    LossCfg L, int n, const float* __restrict__ maps,
    const float* __restrict__ z_vals, const float* __restrict__ raw,
    const float* __restrict__ tgt_d, const float* __restrict__ tgt_rgb,
    const int* __restrict__ n_live, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n) return;
  if (n_live != nullptr && ray >= *n_live) {
    // capacity batch (persistent mapping graph): a ray behind the live
    // count is not part of the batch
    if (lane < 8) stats[(size_t)ray * 8 + lane] = 0.f;
    return;
  }
  const float d = tgt_d[ray];
  float nfs = 0.f, nsdf = 0.f, sfs = 0.f, ssdf = 0.f;
  if (lane < L.S) {
    const float z = z_vals[(size_t)ray * L.S + lane];
    const float sdf = raw[((size_t)ray * L.S + lane) * 4 + 3];
    const bool front = z < d - L.trunc, back = z > d + L.trunc;
    const bool m = !front && !back && d > 0.f;
    nfs = front ? 1.f : 0.f;
    nsdf = m ? 1.f : 0.f;
    const float e0 = sdf - 1.f, e1 = z + sdf * L.trunc - d;
    sfs = front ? e0 * e0 : 0.f;
    ssdf = m ? e1 * e1 : 0.f;
  }
  nfs = wave_sum(nfs);
  nsdf = wave_sum(nsdf);
  sfs = wave_sum(sfs);
  ssdf = wave_sum(ssdf);
  if (lane == 0) {
    const float* m = maps + (size_t)ray * 8;
    const bool valid = d > 0.f && d < L.depth_trunc;
    // reference quirk (joint_encoding.py:106-107): the colour weight tensor
    // is boolean, so rgb_missing != 0 stores True
    const float w = valid ? 1.f : (L.rgb_missing != 0.f ? 1.f : 0.f);
    float sr = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float e = m[c] * w - tgt_rgb[ray * 3 + c] * w;
      sr += e * e;
    }
    const float ed = m[3] - d;
    float* o = stats + (size_t)ray * 8;
    *reinterpret_cast<float4*>(o) = make_float4(nfs, nsdf, sfs, ssdf);
    *reinterpret_cast<float4*>(o + 4) =
        make_float4(valid ? 1.f : 0.f, valid ? ed * ed : 0.f, sr, w);
  }
}

// every block reduces the per-ray statistics (n x 8 floats, L2 resident), then
// writes the gradients of its 16 rays; block 0 writes the loss terms
__global__ __launch_bounds__(1024) void coslam_loss_grad_kernel(
    LossCfg L, int n, const float* __restrict__ maps,
    const float* __restrict__ z_vals, const float* __restrict__ raw,
    const float* __restrict__ tgt_d, const float* __restrict__ tgt_rgb,
    const float* __restrict__ stats, const double* __restrict__ totals,
    int64_t n_total, const int* __restrict__ n_live,
    float* __restrict__ loss_out, float* __restrict__ g_maps,
    float* __restrict__ g_raw) {
  __shared__ double red[16][7];
  const int live = n_live != nullptr ? min(n, *n_live) : n;
  if (n_live != nullptr && totals == nullptr) n_total = live;
  __shared__ double tot[7];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  // totals != NULL: the batch is sharded over ranks and the caller has
  // all-reduced the seven sums; normalisers use the GLOBAL ray count
  for (int r = tid; totals == nullptr && r < n; r += 1024) {
    const float4 a = *reinterpret_cast<const float4*>(stats + (size_t)r * 8);
    const float4 b = *reinterpret_cast<const float4*>(stats + (size_t)r * 8 + 4);
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
    acc[4] += b.x; acc[5] += b.y; acc[6] += b.z;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (tid < 7) {
    double v = 0;
    for (int w = 0; w < 16; ++w) v += red[w][tid];
    tot[tid] = totals != nullptr ? totals[tid] : v;
  }
  __syncthreads();
  const float n_fs = (float)tot[0], n_sdf = (float)tot[1];
  const float n_all = n_fs + n_sdf;
  const float fs_w = 1.f - n_fs / n_all, sdf_w = 1.f - n_sdf / n_all;
  const float n_glob = (float)n_total;
  const float nS = n_glob * (float)L.S;
  const float nv = fmaxf((float)tot[4], 1.f);
  if (blockIdx.x == 0 && tid == 0) {
    const float l_rgb = (float)(tot[6] / (3.0 * n_total)) * L.w_rgb;
    const float l_d = (float)(tot[5] / nv) * L.w_depth;
    const float l_sdf = (float)(tot[3] / nS) * sdf_w * L.w_sdf;
    const float l_fs = (float)(tot[2] / nS) * fs_w * L.w_fs;
    loss_out[0] = l_rgb + l_d + l_sdf + l_fs;
    loss_out[1] = l_rgb;
    loss_out[2] = l_d;
    loss_out[3] = l_sdf;
    loss_out[4] = l_fs;
  }
  const int ray = blockIdx.x * 16 + wave;
  if (ray >= n) return;
  if (ray >= live) {
    if (lane < L.S)
      *reinterpret_cast<float4*>(g_raw + ((size_t)ray * L.S + lane) * 4) =
          make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < 2)
      *reinterpret_cast<float4*>(g_maps + (size_t)ray * 8 + lane * 4) =
          make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float d = tgt_d[ray];
  if (lane < L.S) {
    const size_t o = (size_t)ray * L.S + lane;
    const float z = z_vals[o], sdf = raw[o * 4 + 3];
    const bool front = z < d - L.trunc, back = z > d + L.trunc;
    const bool m = !front && !back && d > 0.f;
    float g = 0.f;
    if (front) g += L.w_fs * fs_w * 2.f * (sdf - 1.f) / nS;
    if (m) g += L.w_sdf * sdf_w * 2.f * (z + sdf * L.trunc - d) * L.trunc / nS;
    *reinterpret_cast<float4*>(g_raw + o * 4) = make_float4(0.f, 0.f, 0.f, g);
  }
  if (lane == 0) {
    const float* mp = maps + (size_t)ray * 8;
    const bool valid = d > 0.f && d < L.depth_trunc;
    const float w = valid ? 1.f : (L.rgb_missing != 0.f ? 1.f : 0.f);
    const float k = L.w_rgb * 2.f * w * w / (3.f * n_glob);
    float* gm = g_maps + (size_t)ray * 8;
    *reinterpret_cast<float4*>(gm) = make_float4(
        k * (mp[0] - tgt_rgb[ray * 3]), k * (mp[1] - tgt_rgb[ray * 3 + 1]),
        k * (mp[2] - tgt_rgb[ray * 3 + 2]),
        valid ? L.w_depth * 2.f * (mp[3] - d) / nv : 0.f);
    *reinterpret_cast<float4*>(gm + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

int check_scene(const Scene* sc, int n_rays) {
  if (sc == nullptr || n_rays < 0) return XRD_ERR_ARG;
  if (sc->table == nullptr || sc->pack == nullptr || sc->t_near == nullptr ||
      sc->t_far == nullptr)
    return XRD_ERR_ARG;
  if (sc->n_range_d < 1 || sc->n_sample_d < 0 ||
      (sc->n_sample_d > 0 && sc->t_uniform == nullptr))
    return XRD_ERR_ARG;
  const int S = sc->n_range_d + sc->n_sample_d;
  if (S > kMaxS) return XRD_ERR_UNSUPPORTED;
  for (int l = 0; l < XRD_COSLAM_LEVELS; ++l) {
    const uint64_t r = sc->lv_res[l], sz = sc->lv_size[l];
    if (r < 2 || sz == 0) return XRD_ERR_ARG;
    if (r * r * r > sz && (sz & (sz - 1)) != 0) return XRD_ERR_UNSUPPORTED;
  }
  return XRD_OK;
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int xrd_coslam_flat_len(void) { return cs::kFlatLen; }
int xrd_coslam_pack_len(void) { return cs::kPackLen; }
int xrd_coslam_dw_len(void) { return cs::kDwLen; }

int xrd_coslam_index(int32_t* pack_idx, int32_t* dw_idx) {
  using namespace cs;
  if (pack_idx != nullptr) {
    const int fbase[4] = {kF0, kF1, kF2, kF3}, bbase[4] = {kB0, kB1, kB2, kB3};
    for (int L = 0; L < 4; ++L) {
      const int ks_f = kIn[L] / 4, ks_b = kOut[L] / 4;
      // forward: frag (M, s): lane (i,k) -> W[out 16M+i][in 16(s>>2)+4k+(s&3)]
      for (int M = 0; M < kOut[L] / 16; ++M)
        for (int s = 0; s < ks_f; ++s)
          for (int l = 0; l < 64; ++l) {
            const int i = l & 15, k = l >> 4;
            pack_idx[(fbase[L] + M * ks_f + s) * 64 + l] =
                flat_of(L, 16 * M + i, 16 * (s >> 2) + 4 * k + (s & 3));
          }
      // backward: frag (M', s): lane (i,k) -> W[out 16(s>>2)+4k+(s&3)][in 16M'+i]
      for (int M = 0; M < kIn[L] / 16; ++M)
        for (int s = 0; s < ks_b; ++s)
          for (int l = 0; l < 64; ++l) {
            const int i = l & 15, k = l >> 4;
            pack_idx[(bbase[L] + M * ks_b + s) * 64 + l] =
                flat_of(L, 16 * (s >> 2) + 4 * k + (s & 3), 16 * M + i);
          }
    }
  }
  if (dw_idx != nullptr) {
    const int dbase[4] = {kD0, kD1, kD2, kD3};
    for (int i = 0; i < kFlatLen; ++i) dw_idx[i] = -1;
    for (int L = 0; L < 4; ++L)
      for (int o = 0; o < kOut[L]; ++o)
        for (int i = 0; i < kIn[L]; ++i) {
          const int f = flat_of(L, o, i);
          if (f >= 0) dw_idx[f] = dbase[L] + o * kIn[L] + i;
        }
    for (int i = 0; i < kFlatLen; ++i)
      if (dw_idx[i] < 0) return XRD_ERR_LAUNCH;  // layout bug
  }
  return XRD_OK;
}

int xrd_coslam_render_fwd(const xrd_coslam_scene* scene, int n_rays,
                          const float* rays_o, const float* rays_d,
                          const float* target_d, const float* rnd,
                          float* z_vals, float* raw, float* maps,
                          xrd_stream_t stream) {
  int rc = check_scene(scene, n_rays);
  if (rc != XRD_OK) return rc;
  if (!rays_o || !rays_d || !target_d || !z_vals || !raw || !maps ||
      (scene->perturb && !rnd))
    return XRD_ERR_ARG;
  if (n_rays == 0) return XRD_OK;
  const int S = scene->n_range_d + scene->n_sample_d;
  const int nt = (S + 15) / 16;
  hipLaunchKernelGGL(coslam_fwd_kernel, dim3(n_rays), dim3(nt * 64), 0,
                     (hipStream_t)stream, *scene, n_rays, rays_o, rays_d,
                     target_d, rnd, z_vals, raw, maps);
  return check_launch("coslam_fwd_kernel");
}

// workspace: [block partials of dW][normalised sample positions N*3]
// [hash-feature gradients 16*N*2], N = n_rays * kMaxS
int64_t xrd_coslam_bwd_ws_floats(int n_rays) {
  return xrd_coslam_bwd_ws_floats_extra(n_rays, 0);
}
int64_t xrd_coslam_bwd_ws_floats_extra(int n_rays, int64_t n_extra) {
  if (n_rays < 0 || n_extra < 0) return -1;
  return (int64_t)kBwdMaxBlocks * cs::kDwLen +
         ((int64_t)n_rays * kMaxS + n_extra) * (3 + 2 * XRD_COSLAM_LEVELS);
}

int xrd_coslam_render_bwd(const xrd_coslam_scene* scene, int n_rays,
                          const float* rays_o, const float* rays_d,
                          const float* z_vals, const float* raw,
                          const float* g_maps, const float* g_raw,
                          float* g_rays_o, float* g_rays_d, float* g_table,
                          float* g_dw, float* workspace, xrd_stream_t stream) {
  return xrd_coslam_render_bwd_extra(scene, n_rays, rays_o, rays_d, z_vals,
                                     raw, g_maps, g_raw, g_rays_o, g_rays_d,
                                     g_table, g_dw, 0, nullptr, nullptr,
                                     workspace, stream);
}

// extra points [n_extra,3] (normalised like the samples) with feature
// gradients [n_extra, 32] (point-major) -> behind the samples in the
// level-major staging of the table scatter
namespace xrd {
namespace {
__global__ __launch_bounds__(256) void coslam_extra_stage_kernel(
    int64_t n_extra, int64_t n_own, const float* __restrict__ ex,
    const float* __restrict__ edf, float* __restrict__ xs,
    float* __restrict__ dfeat) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int l = threadIdx.x & 15;
  const int64_t p = gid >> 4;
  if (p >= n_extra) return;
  const int64_t stride = n_own + n_extra;
  *reinterpret_cast<float2*>(dfeat + ((size_t)l * stride + n_own + p) * 2) =
      *reinterpret_cast<const float2*>(edf + p * 2 * XRD_COSLAM_LEVELS +
                                       2 * l);
  if (l < 3) xs[(n_own + p) * 3 + l] = ex[p * 3 + l];
}
}  // namespace
}  // namespace xrd

int xrd_coslam_render_bwd_extra(
    const xrd_coslam_scene* scene, int n_rays, const float* rays_o,
    const float* rays_d, const float* z_vals, const float* raw,
    const float* g_maps, const float* g_raw, float* g_rays_o, float* g_rays_d,
    float* g_table, float* g_dw, int64_t n_extra, const float* extra_x,
    const float* extra_dfeat, float* workspace, xrd_stream_t stream) {
  int rc = check_scene(scene, n_rays);
  if (rc != XRD_OK) return rc;
  if (n_extra < 0 || (n_extra > 0 && (!extra_x || !extra_dfeat || !g_table)))
    return XRD_ERR_ARG;
  if (!rays_o || !rays_d || !z_vals || !raw || !g_maps) return XRD_ERR_ARG;
  const bool dp = g_rays_o != nullptr || g_rays_d != nullptr;
  const bool dg = g_table != nullptr || g_dw != nullptr;
  if (dp && (!g_rays_o || !g_rays_d)) return XRD_ERR_ARG;
  if (dg && (!g_table || !g_dw || !workspace)) return XRD_ERR_ARG;
  if (!dp && !dg) return XRD_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dp) {
    // (the two ray gradients back to back in one buffer: one fill launch)
    if (g_rays_d == g_rays_o + (size_t)3 * n_rays) {
      if ((rc = zero_floats(g_rays_o, (size_t)6 * n_rays, stream)) != XRD_OK)
        return rc;
    } else if ((rc = zero_floats(g_rays_o, (size_t)3 * n_rays, stream)) !=
                   XRD_OK ||
               (rc = zero_floats(g_rays_d, (size_t)3 * n_rays, stream)) !=
                   XRD_OK) {
      return rc;
    }
  }
  if (dg && (rc = zero_floats(g_dw, cs::kDwLen, stream)) != XRD_OK) return rc;
  if (n_rays == 0) return XRD_OK;
  const int S = scene->n_range_d + scene->n_sample_d;
  const int tiles = n_rays * ((S + 15) / 16);
  int blocks = (tiles + kBwdWaves - 1) / kBwdWaves;
  if (dg && blocks > kBwdMaxBlocks) blocks = kBwdMaxBlocks;
  const int64_t n_samples = (int64_t)n_rays * S;
  float* xs = dg ? workspace + (size_t)kBwdMaxBlocks * cs::kDwLen : nullptr;
  float* dfeat = dg ? xs + (n_samples + n_extra) * 3 : nullptr;
#define BWD_CASE(DPV, DGV)                                                    \
  hipLaunchKernelGGL((coslam_bwd_kernel<DPV, DGV>), dim3(blocks),             \
                     dim3(kBwdWaves * 64), 0, st, *scene, n_rays, rays_o,     \
                     rays_d, z_vals, raw, g_maps, g_raw, g_rays_o, g_rays_d,  \
                     workspace, xs, dfeat, n_extra)
  // map + pose gradients: two launches (each recomputes the tile's forward)
  // beat the combined variant, which needs all 512 registers and runs at one
  // wave per SIMD: 2.8 + 3.3 ms vs 9.1 ms at 1e5 rays, 0.23 vs 0.36 ms at
  // 2560 rays (profiles/r01_large_batch.txt)
  if (dg) BWD_CASE(false, true);
  if (dp) {
    const int all = (tiles + kBwdWaves - 1) / kBwdWaves;
    const int saved = blocks;
    blocks = all;
    BWD_CASE(true, false);
    blocks = saved;
  }
#undef BWD_CASE
  rc = check_launch("coslam_bwd_kernel");
  if (rc != XRD_OK) return rc;
  if (dg) {
    const int chunk = 32;
    hipLaunchKernelGGL(coslam_reduce_kernel,
                       dim3((cs::kDwLen + 255) / 256, (blocks + chunk - 1) / chunk),
                       dim3(256), 0, st, workspace, blocks, chunk, g_dw);
    rc = check_launch("coslam_reduce_kernel");
    if (rc != XRD_OK) return rc;
    if (n_extra > 0) {
      hipLaunchKernelGGL(coslam_extra_stage_kernel,
                         dim3((unsigned)((n_extra * 16 + 255) / 256)), dim3(256),
                         0, st, n_extra, n_samples, extra_x, extra_dfeat, xs,
                         dfeat);
      rc = check_launch("coslam_extra_stage_kernel");
      if (rc != XRD_OK) return rc;
    }
    // ONE scatter for the samples and the extra points (the smoothness
    // lattice): a second launch costs as much as the first (~140 us)
    rc = launch_hash_chunk_scatter(XRD_COSLAM_LEVELS, scene->lv_scale,
                                   scene->lv_res, scene->lv_size,
                                   scene->lv_offset, n_samples + n_extra, xs,
                                   dfeat, 2, 2 * (n_samples + n_extra),
                                   g_table, /*accumulate=*/false, stream);
  }
  return rc;
}

static int coslam_loss_check(int n_rays, int n_samples, const void* a,
                             const void* b, const void* c, const void* d,
                             const void* e, const void* ws) {
  if (n_rays < 1 || n_samples < 1 || n_samples > 64 || !a || !b || !c || !d ||
      !e || !ws)
    return XRD_ERR_ARG;
  return XRD_OK;
}

int xrd_coslam_loss_stats(int n_rays, int n_samples, float trunc,
                          float depth_trunc, float rgb_missing,
                          const float* maps, const float* z_vals,
                          const float* raw, const float* target_d,
                          const float* target_rgb, float* stats,
                          xrd_stream_t stream) {
  int rc = coslam_loss_check(n_rays, n_samples, maps, z_vals, raw, target_d,
                             target_rgb, stats);
  if (rc != XRD_OK) return rc;
  const LossCfg L{0.f, 0.f, 0.f, 0.f, trunc, depth_trunc, rgb_missing,
                  n_samples};
  hipLaunchKernelGGL(coslam_loss_stats_kernel, dim3((n_rays + 3) / 4), dim3(256),
                     0, (hipStream_t)stream, L, n_rays, maps, z_vals, raw,
                     target_d, target_rgb, (const int*)nullptr, stats);
  return check_launch("xrd_coslam_loss_stats");
}

int xrd_coslam_loss_grads(int n_rays, int n_samples, float w_rgb, float w_depth,
                          float w_sdf, float w_fs, float trunc,
                          float depth_trunc, float rgb_missing,
                          const float* maps, const float* z_vals,
                          const float* raw, const float* target_d,
                          const float* target_rgb, const float* stats,
                          const double* totals7, int64_t n_rays_total,
                          float* loss5, float* g_maps, float* g_raw,
                          xrd_stream_t stream) {
  int rc = coslam_loss_check(n_rays, n_samples, maps, z_vals, raw, target_d,
                             target_rgb, stats);
  if (rc != XRD_OK) return rc;
  if (!loss5 || !g_maps || !g_raw) return XRD_ERR_ARG;
  if (totals7 == nullptr) n_rays_total = n_rays;
  if (n_rays_total < n_rays) return XRD_ERR_ARG;
  const LossCfg L{w_rgb, w_depth, w_sdf, w_fs, trunc, depth_trunc, rgb_missing,
                  n_samples};
  hipLaunchKernelGGL(coslam_loss_grad_kernel, dim3((n_rays + 15) / 16),
                     dim3(1024), 0, (hipStream_t)stream, L, n_rays, maps, z_vals,
                     raw, target_d, target_rgb, stats, totals7, n_rays_total,
                     (const int*)nullptr, loss5, g_maps, g_raw);
  return check_launch("xrd_coslam_loss_grads");
}

int xrd_coslam_loss_live(int n_rays, int n_samples, float w_rgb, float w_depth,
                         float w_sdf, float w_fs, float trunc,
                         float depth_trunc, float rgb_missing,
                         const float* maps, const float* z_vals,
                         const float* raw, const float* target_d,
                         const float* target_rgb, const int32_t* n_live,
                         float* loss5, float* g_maps, float* g_raw,
                         float* workspace, xrd_stream_t stream) {
  int rc = coslam_loss_check(n_rays, n_samples, maps, z_vals, raw, target_d,
                             target_rgb, workspace);
  if (rc != XRD_OK) return rc;
  if (!n_live || !loss5 || !g_maps || !g_raw) return XRD_ERR_ARG;
  const LossCfg L{w_rgb, w_depth, w_sdf, w_fs, trunc, depth_trunc, rgb_missing,
                  n_samples};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(coslam_loss_stats_kernel, dim3((n_rays + 3) / 4), dim3(256),
                     0, st, L, n_rays, maps, z_vals, raw, target_d, target_rgb,
                     n_live, workspace);
  hipLaunchKernelGGL(coslam_loss_grad_kernel, dim3((n_rays + 15) / 16),
                     dim3(1024), 0, st, L, n_rays, maps, z_vals, raw, target_d,
                     target_rgb, workspace, (const double*)nullptr,
                     (int64_t)n_rays, n_live, loss5, g_maps, g_raw);
  return check_launch("xrd_coslam_loss_live");
}

int xrd_coslam_loss(int n_rays, int n_samples, float w_rgb, float w_depth,
                    float w_sdf, float w_fs, float trunc, float depth_trunc,
                    float rgb_missing, const float* maps, const float* z_vals,
                    const float* raw, const float* target_d,
                    const float* target_rgb, float* loss5, float* g_maps,
                    float* g_raw, float* workspace, xrd_stream_t stream) {
  int rc = xrd_coslam_loss_stats(n_rays, n_samples, trunc, depth_trunc,
                                 rgb_missing, maps, z_vals, raw, target_d,
                                 target_rgb, workspace, stream);
  if (rc != XRD_OK) return rc;
  return xrd_coslam_loss_grads(n_rays, n_samples, w_rgb, w_depth, w_sdf, w_fs,
                               trunc, depth_trunc, rgb_missing, maps, z_vals,
                               raw, target_d, target_rgb, workspace, nullptr,
                               n_rays, loss5, g_maps, g_raw, stream);
}

}  // extern "C"
