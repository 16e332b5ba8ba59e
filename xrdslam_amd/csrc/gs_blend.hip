// Tile blend of the 3-D Gaussian rasteriser, third formulation (gfx950):
// one WAVE per 8x8 SUB-TILE with exact sub-tile culling, both directions
// front-to-back, per-Gaussian gradients reduced in-wave and summed in LDS.
//
// Replaces the blend part of the unvendored CUDA dependency
// diff-gaussian-rasterization-w-depth @ cb65e4b (reference call sites
// slam/model_components/gaussian_cloud_splatam.py:63-69,267-268; algorithm per
// SURVEY.md App. C.3, oracle: oracle/gs_oracle.py, parity unpinned).
//
// What bounds this work on MI355X.  Both blends are fp32-VALU bound on
// (Gaussian, pixel) evaluations.  The published kernels (and the two earlier
// formulations here) evaluate every Gaussian of a 16x16 tile's list at all 256
// pixels; SplaTAM's Gaussians are ~1 px sigma, a tile-list entry reaches
// ~30 of the 256 pixels and the rest fail the alpha >= 1/255 test after the
// full evaluation (measured: 185 M evaluations per 640x480 pass, 98 M up to
// the last contributor, < 20 M contributing).  Here the 16x16 tile's block is
// four waves, one per 8x8 sub-tile.  Per bucket of 64 list entries (staged in
// LDS once per block) each wave tests entry l against its sub-tile in lane l
// — the axis-aligned bound of { alpha >= 1/255 } = { power >= -ln(255 op) },
// widened by a margin — and walks only the set bits of the ballot.  The test
// can only drop evaluations whose alpha test fails: the set of contributing
// (Gaussian, pixel) pairs, their order and therefore the image and the
// gradients are those of the 16x16 formulation.
//
// Backward, front to back.  With
//   dL/dalpha_i = T_i q_i - (R - A_i - q_i alpha_i T_i) / (1 - alpha_i),
//   q_i = c_i . dL/dC,  A_i = sum_{j<i} q_j alpha_j T_j,
//   R = sum_j q_j alpha_j T_j + T_final (bg . dL/dC)
// (the published back-to-front recurrence rearranged; R comes from the
// forward's image) a pixel needs nothing from the Gaussians behind the current
// one: the backward walks the list in the forward's order with T and A in
// registers — no per-pixel history, no checkpoints.  Per entry a wave reduces
// twelve per-pixel values over its 64 pixels with a transposed DPP merge
// (12 -> 6 -> 3 registers, then two row shifts; 33 VALU instead of 72) and
// adds them to the entry's LDS row; after the bucket the block writes one
// finished gradient row per (Gaussian, tile) key.  gs_key_reduce sums a
// Gaussian's rows through the inverse map of the binning sort (gs_bin.hip).
// No global atomics anywhere.
#include <hip/hip_runtime.h>

#include "common.h"

namespace xrd {
namespace {

constexpr int TILE = 16;
constexpr int BLOCK = TILE * TILE;
#ifndef XRD_GS_BUCKET
#define XRD_GS_BUCKET 64
#endif
constexpr int BUCKET = XRD_GS_BUCKET;  // list entries staged per barrier
constexpr int CHUNKS = BUCKET / 64;     // 64-entry chunks, one ballot each
constexpr int KEYROW = 12;  // col a 3, col b 3, mean2D 2, conic 3, opacity 1
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct BCam {
  int H, W;
  float bg[3];
};

// one bucket of a tile's list in LDS
template <bool DUAL, bool BWD>
struct Stage {
  float2 xy[BUCKET];
  float2 ext[BUCKET];  // half extents of { alpha >= 1/255 } + margin
  f32x4 q[BUCKET];     // -0.5 log2e A, -log2e B, -0.5 log2e C, opacity
  f32x4 ca[BUCKET];    // r, g, b, depth
  f32x4 cb[DUAL ? BUCKET : 1];
  f32x4 con[BWD ? BUCKET : 1];           // A, B, C, opacity
  float acc[BWD ? BUCKET * KEYROW : 1];  // per entry: the twelve sums
};

// wave w of the block fills its share of bucket `b` of the list [r0, r1)
template <bool DUAL, bool BWD>
__device__ __forceinline__ void stage_bucket(
    Stage<DUAL, BWD>& S, int wave, int lane, int r0, int r1, int b,
    const int* __restrict__ plist, const float* __restrict__ xy,
    const float* __restrict__ conic_o, const float* __restrict__ colors,
    const float* __restrict__ colors_b, const float* __restrict__ depths) {
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
  const int e = c * 64 + lane;
  const int k = r0 + b * BUCKET + e;
  const bool have = k < r1;
  const int g = have ? plist[k] : 0;
  if (wave == 0) {
    float2 p = make_float2(0.f, 0.f);
    float2 ex = make_float2(-INFINITY, -INFINITY);  // reaches no pixel
    if (have) {
      p = make_float2(xy[g * 2], xy[g * 2 + 1]);
      const f32x4 co = *reinterpret_cast<const f32x4*>(conic_o + g * 4);
      const float det = co[0] * co[2] - co[1] * co[1];
      // alpha = min(.99, op G) >= 1/255 needs -power <= ln(255 op) = tau:
      // inside the ellipse (1/2) d^T conic d <= tau, whose axis-aligned
      // half extents are sqrt(2 tau C / det), sqrt(2 tau A / det).  Widened
      // (1 % + 0.05 in tau, half a pixel) so that rounding in the kernels'
      // own evaluation can never disagree; anything not provably outside
      // (NaN, degenerate conic) is evaluated.
      const float o255 = 255.f * co[3];
      if (o255 < 0.999f) {
        // alpha <= op < 1/255 everywhere
      } else if (det > 0.f && co[0] > 0.f && co[2] > 0.f && o255 < 1e30f) {
        const float tau = 1.01f * kLn2 * __builtin_amdgcn_logf(o255) + 0.05f;
        const float s = 2.f * tau / det;
        ex = make_float2(sqrtf(s * co[2]) + 0.5f, sqrtf(s * co[0]) + 0.5f);
        if (!(ex.x == ex.x) || !(ex.y == ex.y))
          ex = make_float2(INFINITY, INFINITY);
      } else {
        ex = make_float2(INFINITY, INFINITY);
      }
    }
    S.xy[e] = p;
    S.ext[e] = ex;
  } else if (wave == 1) {
    f32x4 co = {0.f, 0.f, 0.f, 0.f};
    if (have) co = *reinterpret_cast<const f32x4*>(conic_o + g * 4);
    S.q[e] = f32x4{-0.5f * kLog2e * co[0], -kLog2e * co[1],
                      -0.5f * kLog2e * co[2], co[3]};
    if (BWD) S.con[e] = co;
  } else if (wave == 2) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (have)
      c = f32x4{colors[g * 3], colors[g * 3 + 1], colors[g * 3 + 2],
                depths ? depths[g] : 0.f};
    S.ca[e] = c;
  } else if (DUAL) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (have)
      c = f32x4{colors_b[g * 3], colors_b[g * 3 + 1], colors_b[g * 3 + 2],
                0.f};
    S.cb[e] = c;
  }
  }
}

// entries of the bucket that can reach the 8x8 sub-tile whose first pixel
// centre is (sx0, sy0): bit l = entry l
template <bool DUAL, bool BWD>
__device__ __forceinline__ uint64_t reach_mask(const Stage<DUAL, BWD>& S,
                                               int lane, int chunk,
                                               float sx0, float sy0) {
  const float2 p = S.xy[chunk * 64 + lane];
  const float2 e = S.ext[chunk * 64 + lane];
  const bool hit = p.x + e.x >= sx0 && p.x - e.x <= sx0 + 7.f &&
                   p.y + e.y >= sy0 && p.y - e.y <= sy0 + 7.f;
  return __ballot(hit);
}

// G = exp(power) of entry (gxy, q) at the pixel, as 2^(log2e power); the
// forward and the backward share it: same skip decisions
__device__ __forceinline__ float gauss_weight(float2 gxy, const f32x4& q,
                                              float pfx, float pfy, float& dx,
                                              float& dy, float& p2) {
  dx = gxy.x - pfx;
  dy = gxy.y - pfy;
  p2 = dx * (q[0] * dx + q[1] * dy) + q[2] * (dy * dy);
  return __builtin_amdgcn_exp2f(p2);
}

template <int CTRL>
__device__ __forceinline__ int dpp_get_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float,
                            dpp_get_i<CTRL>(__builtin_bit_cast(int, v)));
}

// max over the wave of a non-negative int, in every lane
__device__ __forceinline__ int wave_max_nonneg(int v) {
  v = max(v, dpp_get_i<0xB1>(v));    // quad_perm [1,0,3,2]
  v = max(v, dpp_get_i<0x4E>(v));    // quad_perm [2,3,0,1]
  v = max(v, dpp_get_i<0x141>(v));   // row_half_mirror
  v = max(v, dpp_get_i<0x140>(v));   // row_mirror
  v = max(v, __shfl_xor(v, 16));
  v = max(v, __shfl_xor(v, 32));
  return v;
}

// DUAL: a second colour set blended with the same weights (SplaTAM renders rgb
// and (z, 1, z^2) with identical geometry)
template <bool DUAL>
__global__ __launch_bounds__(BLOCK) void gs_blend_fwd_kernel(
    BCam cam, const int* __restrict__ ranges, const int* __restrict__ plist,
    const float* __restrict__ xy, const float* __restrict__ colors,
    const float* __restrict__ colors_b, const float* __restrict__ conic_o,
    const float* __restrict__ depths, float* __restrict__ out_color,
    float* __restrict__ out_color_b, float* __restrict__ out_depth,
    float* __restrict__ final_T, int* __restrict__ n_contrib) {
  __shared__ Stage<DUAL, false> S;
  const int gx = (cam.W + TILE - 1) / TILE;
  const int tile = blockIdx.y * gx + blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int sx0 = blockIdx.x * TILE + (wave & 1) * 8,
            sy0 = blockIdx.y * TILE + (wave >> 1) * 8;
  const int px = sx0 + (lane & 7), py = sy0 + (lane >> 3);
  const bool inside = px < cam.W && py < cam.H;
  const float pfx = (float)px, pfy = (float)py;
  const int r0 = ranges[tile * 2], r1 = ranges[tile * 2 + 1];
  const int n_buckets = (r1 - r0 + BUCKET - 1) / BUCKET;
  bool done = !inside;
  float T = 1.f, C[3] = {0.f, 0.f, 0.f}, Cb[3] = {0.f, 0.f, 0.f}, D = 0.f;
  int last = 0;
  for (int b = 0; b < n_buckets; ++b) {
    if (__syncthreads_count(done) == BLOCK) break;  // also: S is free
    stage_bucket<DUAL, false>(S, wave, lane, r0, r1, b, plist, xy, conic_o,
                              colors, colors_b, depths);
    __syncthreads();
    for (int c = 0; c < CHUNKS; ++c) {
      if (__ballot(!done) == 0) break;
      uint64_t m = reach_mask(S, lane, c, (float)sx0, (float)sy0);
      if (m == 0) continue;
      // the next entry's data is read while the current one is blended
      int j = c * 64 + __builtin_ctzll(m);
      f32x4 q = S.q[j], cd = S.ca[j], cb = DUAL ? S.cb[j] : f32x4{};
      float2 gxy = S.xy[j];
      while (true) {
        m &= m - 1;
        const int jn = m ? c * 64 + __builtin_ctzll(m) : j;
        const f32x4 qn = S.q[jn], cdn = S.ca[jn],
                    cbn = DUAL ? S.cb[jn] : f32x4{};
        const float2 gxyn = S.xy[jn];
        float dx, dy, p2;
        const float G = gauss_weight(gxy, q, pfx, pfy, dx, dy, p2);
        const float alpha = fminf(0.99f, q[3] * G);
        const float test_T = T * (1.f - alpha);
        const bool use = !done && p2 <= 0.f && alpha >= 1.f / 255.f;
        if (use && test_T < 0.0001f) done = true;
        if (use && !done) {
          const float w = alpha * T;
          C[0] += cd[0] * w;
          C[1] += cd[1] * w;
          C[2] += cd[2] * w;
          D += cd[3] * w;
          if (DUAL) {
            Cb[0] += cb[0] * w;
            Cb[1] += cb[1] * w;
            Cb[2] += cb[2] * w;
          }
          T = test_T;
          last = b * BUCKET + j + 1;
        }
        if (m == 0 || __ballot(!done) == 0) break;
        j = jn;
        q = qn;
        cd = cdn;
        cb = cbn;
        gxy = gxyn;
      }
    }
  }
  if (inside) {
    const int pix = py * cam.W + px;
    final_T[pix] = T;
    n_contrib[pix] = last;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      out_color[ch * cam.H * cam.W + pix] = C[ch] + T * cam.bg[ch];
      if (DUAL)
        out_color_b[ch * cam.H * cam.W + pix] = Cb[ch] + T * cam.bg[ch];
    }
    out_depth[pix] = D;
  }
}

// v[0..11] per lane -> t[0..2]: lanes 12..15 of every 16-lane row hold the
// row's sum of v[4k + (lane & 3)] in t[k]
__device__ __forceinline__ void merge_reduce12(const float (&v)[12], int lane,
                                               float (&t)[3]) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float u[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float keep = b0 ? v[2 * k + 1] : v[2 * k];
    const float give = b0 ? v[2 * k] : v[2 * k + 1];
    u[k] = keep + dpp_get<0xB1>(give);  // quad_perm [1,0,3,2]
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float keep = b1 ? u[2 * k + 1] : u[2 * k];
    const float give = b1 ? u[2 * k] : u[2 * k + 1];
    float s = keep + dpp_get<0x4E>(give);  // quad_perm [2,3,0,1]
    s += dpp_get<0x114>(s);                // row_shr:4 (zeros shifted in)
    s += dpp_get<0x118>(s);                // row_shr:8
    t[k] = s;
  }
}

template <bool DUAL>
__global__ __launch_bounds__(BLOCK) void gs_blend_bwd_kernel(
    BCam cam, const int* __restrict__ ranges, const int* __restrict__ plist,
    const float* __restrict__ xy, const float* __restrict__ conic_o,
    const float* __restrict__ colors, const float* __restrict__ colors_b,
    const float* __restrict__ final_T, const int* __restrict__ n_contrib,
    const float* __restrict__ out_color, const float* __restrict__ out_color_b,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_b,
    float* __restrict__ key_grad) {
  __shared__ Stage<DUAL, true> S;
  __shared__ int s_max;
  const int gx = (cam.W + TILE - 1) / TILE;
  const int tile = blockIdx.y * gx + blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = ranges[tile * 2], r1 = ranges[tile * 2 + 1];
  if (r1 <= r0) return;
  const int sx0 = blockIdx.x * TILE + (wave & 1) * 8,
            sy0 = blockIdx.y * TILE + (wave >> 1) * 8;
  const int px = sx0 + (lane & 7), py = sy0 + (lane >> 3);
  const bool inside = px < cam.W && py < cam.H;
  const float pfx = (float)px, pfy = (float)py;
  const int HW = cam.H * cam.W;
  if (tid == 0) s_max = 0;
  __syncthreads();
  // per-pixel constants: dL/dC of both colour sets, R, last contributor
  float dLa[3] = {0.f, 0.f, 0.f}, dLb[3] = {0.f, 0.f, 0.f}, R = 0.f;
  int nc = 0;
  if (inside) {
    const int pix = py * cam.W + px;
    const float Tf = final_T[pix];
    float bgdot = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      dLa[ch] = dL_dpix[ch * HW + pix];
      bgdot += cam.bg[ch] * dLa[ch];
      R += (out_color[ch * HW + pix] - Tf * cam.bg[ch]) * dLa[ch];
      if (DUAL) {
        dLb[ch] = dL_dpix_b[ch * HW + pix];
        bgdot += cam.bg[ch] * dLb[ch];
        R += (out_color_b[ch * HW + pix] - Tf * cam.bg[ch]) * dLb[ch];
      }
    }
    R += Tf * bgdot;
    nc = n_contrib[pix];
  }
  const int wave_nc = __builtin_amdgcn_readfirstlane(wave_max_nonneg(nc));
  if (lane == 0) atomicMax(&s_max, wave_nc);
  __syncthreads();
  const int max_nc = s_max;  // Gaussians behind it contribute to no pixel
  const int n_buckets = (min(r1 - r0, max_nc) + BUCKET - 1) / BUCKET;
  const float ddx = 0.5f * cam.W, ddy = 0.5f * cam.H;
  float T = 1.f, A = 0.f;
  for (int b = 0; b < n_buckets; ++b) {
    stage_bucket<DUAL, true>(S, wave, lane, r0, r1, b, plist, xy, conic_o,
                             colors, colors_b, nullptr);
#pragma unroll
    for (int k = 0; k < BUCKET * KEYROW / BLOCK; ++k)
      S.acc[k * BLOCK + tid] = 0.f;
    __syncthreads();
    for (int c = 0; c < CHUNKS; ++c) {
      if (b * BUCKET + c * 64 >= wave_nc) break;
      uint64_t m = reach_mask(S, lane, c, (float)sx0, (float)sy0);
      // entries behind the wave's last contributor reach nothing
      const int left = wave_nc - (b * BUCKET + c * 64);
      if (left < 64) m &= (1ull << left) - 1ull;
      if (m == 0) continue;
      int j = c * 64 + __builtin_ctzll(m);
      f32x4 q = S.q[j], ca = S.ca[j], cb = DUAL ? S.cb[j] : f32x4{};
      float2 gxy = S.xy[j];
      while (true) {
        m &= m - 1;
        const int jn = m ? c * 64 + __builtin_ctzll(m) : j;
        const f32x4 qn = S.q[jn], can = S.ca[jn],
                    cbn = DUAL ? S.cb[jn] : f32x4{};
        const float2 gxyn = S.xy[jn];
        const int pos = b * BUCKET + j;
        float dx, dy, p2;
        const float G = gauss_weight(gxy, q, pfx, pfy, dx, dy, p2);
        const float alpha = fminf(0.99f, q[3] * G);
        const bool act = pos < nc && p2 <= 0.f && alpha >= 1.f / 255.f;
        if (__ballot(act) != 0) {
          float qd = ca[0] * dLa[0] + ca[1] * dLa[1] + ca[2] * dLa[2];
          if (DUAL) qd += cb[0] * dLb[0] + cb[1] * dLb[1] + cb[2] * dLb[2];
          const float w = act ? alpha * T : 0.f;
          const float qw = qd * w;
          const float om = 1.f - alpha;
          const float dL_dalpha =
              T * qd - (R - A - qw) * __builtin_amdgcn_rcpf(om);
          const float h = act ? G * dL_dalpha : 0.f;
          const float hx = h * dx, hy = h * dy;
          const float v[12] = {w * dLa[0], w * dLa[1], w * dLa[2], w * dLb[0],
                               w * dLb[1], w * dLb[2], h,          hx,
                               hy,         hx * dx,    hx * dy,    hy * dy};
          float t[3];
          merge_reduce12(v, lane, t);
          if ((lane & 12) == 12) {
            float* row = S.acc + j * KEYROW + (lane & 3);
            atomicAdd(row, t[0]);
            atomicAdd(row + 4, t[1]);
            atomicAdd(row + 8, t[2]);
          }
          T = act ? T * om : T;
          A += qw;
        }
        if (m == 0) break;
        j = jn;
        q = qn;
        ca = can;
        cb = cbn;
        gxy = gxyn;
      }
    }
    __syncthreads();
    // finished rows of the bucket: sums -> gradients w.r.t. colours, mean2D
    // (ndc), conic (true partials), opacity
    for (int e = tid; e < BUCKET; e += BLOCK) {
      const int k = b * BUCKET + e;
      if (r0 + k < r1) {
        const float* s = S.acc + e * KEYROW;
        const f32x4 co = S.con[e];
        const float o = co[3];
        float* row = key_grad + (int64_t)(r0 + k) * KEYROW;
        *reinterpret_cast<f32x4*>(row) = f32x4{s[0], s[1], s[2], s[3]};
        *reinterpret_cast<f32x4*>(row + 4) =
            f32x4{s[4], s[5], -o * (co[0] * s[7] + co[1] * s[8]) * ddx,
                  -o * (co[2] * s[8] + co[1] * s[7]) * ddy};
        *reinterpret_cast<f32x4*>(row + 8) =
            f32x4{-0.5f * o * s[9], -o * s[10], -0.5f * o * s[11], s[6]};
      }
    }
    __syncthreads();
  }
  // keys of this tile behind the last contributor: zero rows
  for (int k = n_buckets * BUCKET + tid; k < r1 - r0; k += BLOCK) {
    float* row = key_grad + (int64_t)(r0 + k) * KEYROW;
#pragma unroll
    for (int c = 0; c < KEYROW; c += 4)
      *reinterpret_cast<f32x4*>(row + c) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// gradients of Gaussian i = sum of the rows of its keys: pre-sort keys
// [offsets[i-1], offsets[i]) through key_pos (sorted position)
template <bool DUAL>
__global__ __launch_bounds__(256) void gs_key_reduce_kernel(
    int n, int64_t cap, const int64_t* __restrict__ offsets,
    const int* __restrict__ key_pos, const float* __restrict__ key_grad,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dopac, float* __restrict__ dL_dcolors,
    float* __restrict__ dL_dcolors_b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc[KEYROW];
#pragma unroll
  for (int c = 0; c < KEYROW; ++c) acc[c] = 0.f;
  int64_t k0 = i == 0 ? 0 : offsets[i - 1], k1 = offsets[i];
  if (k1 > cap) k1 = cap;  // keys beyond the capacity were dropped
  for (int64_t k = k0; k < k1; ++k) {
    const float* row = key_grad + (int64_t)key_pos[k] * KEYROW;
    const f32x4 a = *reinterpret_cast<const f32x4*>(row);
    const f32x4 b = *reinterpret_cast<const f32x4*>(row + 4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(row + 8);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[r] += a[r];
      acc[4 + r] += b[r];
      acc[8 + r] += c[r];
    }
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    dL_dcolors[i * 3 + ch] = acc[ch];
    if (DUAL) dL_dcolors_b[i * 3 + ch] = acc[3 + ch];
    dL_dconic[i * 3 + ch] = acc[8 + ch];
  }
  dL_dmean2D[i * 2] = acc[6];
  dL_dmean2D[i * 2 + 1] = acc[7];
  dL_dopac[i] = acc[11];
}

int to_bcam(const xrd_gs_camera* c, BCam& cam) {
  if (!c || c->image_height < 1 || c->image_width < 1) return XRD_ERR_ARG;
  cam.H = c->image_height;
  cam.W = c->image_width;
  for (int i = 0; i < 3; ++i) cam.bg[i] = c->bg[i];
  return XRD_OK;
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int64_t xrd_gs_blend_ckpt_floats(int64_t key_capacity, int image_width,
                                 int image_height) {
  if (key_capacity < 0 || image_width < 1 || image_height < 1) return -1;
  return 64;  // the front-to-back backward keeps no checkpoints
}

int xrd_gs_blend_fwd(const xrd_gs_camera* c, const int32_t* ranges,
                     const int32_t* point_list, const float* xy,
                     const float* colors_a, const float* colors_b,
                     const float* conic_opacity, const float* depths,
                     float* out_color_a, float* out_color_b, float* out_depth,
                     float* final_T, int32_t* n_contrib, float* ckpt,
                     xrd_stream_t stream) {
  (void)ckpt;
  BCam cam;
  int rc = to_bcam(c, cam);
  if (rc) return rc;
  if (!ranges || !out_color_a || !out_depth || !final_T || !n_contrib)
    return XRD_ERR_ARG;
  if ((colors_b == nullptr) != (out_color_b == nullptr)) return XRD_ERR_ARG;
  const dim3 grid((cam.W + TILE - 1) / TILE, (cam.H + TILE - 1) / TILE);
  if (colors_b)
    hipLaunchKernelGGL(gs_blend_fwd_kernel<true>, grid, dim3(BLOCK), 0,
                       (hipStream_t)stream, cam, ranges, point_list, xy,
                       colors_a, colors_b, conic_opacity, depths, out_color_a,
                       out_color_b, out_depth, final_T, n_contrib);
  else
    hipLaunchKernelGGL(gs_blend_fwd_kernel<false>, grid, dim3(BLOCK), 0,
                       (hipStream_t)stream, cam, ranges, point_list, xy,
                       colors_a, nullptr, conic_opacity, depths, out_color_a,
                       nullptr, out_depth, final_T, n_contrib);
  return check_launch("xrd_gs_blend_fwd");
}

int xrd_gs_blend_bwd(const xrd_gs_camera* c, int n, int64_t key_capacity,
                     const int32_t* ranges, const int32_t* point_list,
                     const int32_t* key_pos, const int64_t* offsets,
                     const float* xy, const float* conic_opacity,
                     const float* colors_a, const float* colors_b,
                     const float* final_T, const int32_t* n_contrib,
                     const float* out_color_a, const float* out_color_b,
                     const float* dL_dcolor_a, const float* dL_dcolor_b,
                     const float* ckpt, float* key_grad, float* dL_dmean2D,
                     float* dL_dconic, float* dL_dopacity, float* dL_dcolors_a,
                     float* dL_dcolors_b, xrd_stream_t stream) {
  (void)ckpt;
  BCam cam;
  int rc = to_bcam(c, cam);
  if (rc) return rc;
  if (n < 0 || key_capacity < 1) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!ranges || !point_list || !key_pos || !offsets || !xy ||
      !conic_opacity || !colors_a || !final_T || !n_contrib || !out_color_a ||
      !dL_dcolor_a || !key_grad || !dL_dmean2D || !dL_dconic ||
      !dL_dopacity || !dL_dcolors_a)
    return XRD_ERR_ARG;
  const bool dual = colors_b != nullptr;
  if (dual && (!out_color_b || !dL_dcolor_b || !dL_dcolors_b))
    return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((cam.W + TILE - 1) / TILE, (cam.H + TILE - 1) / TILE);
  if (dual) {
    hipLaunchKernelGGL(gs_blend_bwd_kernel<true>, grid, dim3(BLOCK), 0, st,
                       cam, ranges, point_list, xy, conic_opacity, colors_a,
                       colors_b, final_T, n_contrib, out_color_a, out_color_b,
                       dL_dcolor_a, dL_dcolor_b, key_grad);
    hipLaunchKernelGGL(gs_key_reduce_kernel<true>, dim3((n + 255) / 256),
                       dim3(256), 0, st, n, key_capacity, offsets, key_pos,
                       key_grad, dL_dmean2D, dL_dconic, dL_dopacity,
                       dL_dcolors_a, dL_dcolors_b);
  } else {
    hipLaunchKernelGGL(gs_blend_bwd_kernel<false>, grid, dim3(BLOCK), 0, st,
                       cam, ranges, point_list, xy, conic_opacity, colors_a,
                       nullptr, final_T, n_contrib, out_color_a, nullptr,
                       dL_dcolor_a, nullptr, key_grad);
    hipLaunchKernelGGL(gs_key_reduce_kernel<false>, dim3((n + 255) / 256),
                       dim3(256), 0, st, n, key_capacity, offsets, key_pos,
                       key_grad, dL_dmean2D, dL_dconic, dL_dopacity,
                       dL_dcolors_a, nullptr);
  }
  return check_launch("xrd_gs_blend_bwd");
}

}  // extern "C"
