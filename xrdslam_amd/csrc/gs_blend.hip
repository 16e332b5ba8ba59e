// Tile blend of the 3-D Gaussian rasteriser, fourth formulation (gfx950):
// one WAVE per 8x8 SUB-TILE with exact sub-tile culling, the list entries read
// as packed records through the SCALAR path, both directions front-to-back,
// per-Gaussian gradients reduced in-wave and summed in LDS.
//
// Replaces the blend part of the unvendored CUDA dependency
// diff-gaussian-rasterization-w-depth @ cb65e4b (reference call sites
// slam/model_components/gaussian_cloud_splatam.py:63-69,267-268; algorithm per
// SURVEY.md App. C.3, oracle: oracle/gs_oracle.py, parity unpinned).
//
// What bounds this work on MI355X.  Both blends are fp32-VALU bound on
// (Gaussian, pixel) evaluations.  The published kernels (and the first two
// formulations here) evaluate every Gaussian of a 16x16 tile's list at all 256
// pixels; SplaTAM's Gaussians are ~1 px sigma, a tile-list entry reaches
// ~30 of the 256 pixels and the rest fail the alpha >= 1/255 test after the
// full evaluation (measured at 640x480 / 320 k Gaussians: 185 M evaluations a
// pass, 98 M up to the last contributor, < 20 M contributing).  Here the
// 16x16 tile's block is four waves, one per 8x8 sub-tile.  Per chunk of 64
// list entries a wave tests entry l against its sub-tile in lane l — the
// axis-aligned bound of { alpha >= 1/255 } = { power >= -ln(255 op) }, widened
// by a margin — and walks only the set bits of the ballot (measured: 1.75 of
// the 4 sub-tiles per entry).  The test can only drop evaluations whose alpha
// test fails: the set of contributing (Gaussian, pixel) pairs, their order and
// therefore the image and the gradients are those of the 16x16 formulation.
//
// Records.  gs_pack writes one 64-byte record per sorted (Gaussian, tile) key:
// centre, reach, conic pre-scaled by log2(e) (G = v_exp_f32 of one fma
// chain), opacity, both colour sets, depth.  The entry a wave works on is the
// same for its 64 lanes: its record is fetched with ONE s_load_dwordx16 into
// SGPRs (the third formulation staged it in LDS and broadcast-read 52 B per
// lane and entry — the LDS pipe, not the VALU, was the bound), the next
// entry's record is in flight while the current one is blended.  The forward
// has no LDS and no barrier at all.
//
// Backward, front to back.  With
//   dL/dalpha_i = T_i q_i - (R - A_i - q_i alpha_i T_i) / (1 - alpha_i),
//   q_i = c_i . dL/dC,  A_i = sum_{j<i} q_j alpha_j T_j,
//   R = sum_j q_j alpha_j T_j + T_final (bg . dL/dC)
// (the published back-to-front recurrence rearranged; R comes from the
// forward's image) a pixel needs nothing from the Gaussians behind the current
// one: the backward walks the list in the forward's order with T and A in
// registers — no per-pixel history, no checkpoints.  Per entry a wave reduces
// twelve per-pixel values over its 64 pixels with a transposed merge
// (v_permlane32_swap / v_permlane16_swap / DPP: 27 VALU instead of 72 for
// twelve butterflies) that leaves each sum in its own lane, and adds them to
// the entry's LDS row with one conflict-free ds_add; after 256 entries the
// block writes one finished gradient row per (Gaussian, tile) key.
// gs_key_reduce sums a Gaussian's rows — filed under the pairs' pre-sort
// index (gs_bin.hip), hence contiguous.  No global atomics anywhere.
#include <hip/hip_runtime.h>

#include "common.h"

namespace xrd {
namespace {

constexpr int TILE = 16;
constexpr int BLOCK = TILE * TILE;
constexpr int BUCKET = 256;  // entries per LDS row block of the backward
constexpr int KEYROW = 12;   // col a 3, col b 3, mean2D 2, conic 3, opacity 1
constexpr int REC = 16;      // floats per record
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct BCam {
  int H, W;
  float bg[3];
};

// record of one sorted key:
//  0 x   1 y   2 ex  3 ey     centre, half extents of { alpha >= 1/255 }
//  4 q0  5 q1  6 q2  7 op     -0.5 log2e A, -log2e B, -0.5 log2e C, opacity
//  8 r   9 g  10 b  11 depth
// 12..14 second colour set, 15 unused
__global__ __launch_bounds__(256) void gs_pack_kernel(
    const int* __restrict__ ranges, const int* __restrict__ plist,
    const float* __restrict__ xy, const float* __restrict__ conic_o,
    const float* __restrict__ colors, const float* __restrict__ colors_b,
    const float* __restrict__ depths, float* __restrict__ rec) {
  const int tile = blockIdx.x;
  const int r0 = ranges[tile * 2], r1 = ranges[tile * 2 + 1];
  for (int k = r0 + threadIdx.x; k < r1; k += 256) {
    const int g = plist[k];
    const f32x4 co = *reinterpret_cast<const f32x4*>(conic_o + g * 4);
    float ex = -INFINITY, ey = -INFINITY;   // reaches no pixel
    const float det = co[0] * co[2] - co[1] * co[1];
    // alpha = min(.99, op G) >= 1/255 needs -power <= ln(255 op) = tau:
    // inside the ellipse (1/2) d^T conic d <= tau, whose axis-aligned half
    // extents are sqrt(2 tau C / det), sqrt(2 tau A / det).  Widened (1 % +
    // 0.05 in tau, half a pixel) so that rounding in the kernels' own
    // evaluation can never disagree; anything not provably outside (NaN,
    // degenerate conic) is evaluated.
    const float o255 = 255.f * co[3];
    if (o255 < 0.999f) {
      // alpha <= op < 1/255 everywhere
    } else if (det > 0.f && co[0] > 0.f && co[2] > 0.f && o255 < 1e30f) {
      const float tau = 1.01f * kLn2 * __builtin_amdgcn_logf(o255) + 0.05f;
      const float s = 2.f * tau / det;
      ex = sqrtf(s * co[2]) + 0.5f;
      ey = sqrtf(s * co[0]) + 0.5f;
      if (!(ex == ex) || !(ey == ey)) ex = ey = INFINITY;
    } else {
      ex = ey = INFINITY;
    }
    f32x4* out = reinterpret_cast<f32x4*>(rec + (int64_t)k * REC);
    out[0] = f32x4{xy[g * 2], xy[g * 2 + 1], ex, ey};
    out[1] = f32x4{-0.5f * kLog2e * co[0], -kLog2e * co[1],
                   -0.5f * kLog2e * co[2], co[3]};
    out[2] = f32x4{colors[g * 3], colors[g * 3 + 1], colors[g * 3 + 2],
                   depths[g]};
    out[3] = colors_b ? f32x4{colors_b[g * 3], colors_b[g * 3 + 1],
                              colors_b[g * 3 + 2], 0.f}
                      : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

struct Rec {
  f32x4 a, b, c, d;
};

// record j of the tile's list: j is wave-uniform, the loads are scalar
__device__ __forceinline__ Rec load_rec(const float* __restrict__ base,
                                        int j) {
  const f32x4* p = reinterpret_cast<const f32x4*>(
      base + (int64_t)__builtin_amdgcn_readfirstlane(j) * REC);
  return Rec{p[0], p[1], p[2], p[3]};
}

// lane l: centre and reach of entry `e` of the list (nothing beyond `len`)
__device__ __forceinline__ f32x4 load_head(const float* __restrict__ base,
                                           int e, int len) {
  if (e >= len) return f32x4{0.f, 0.f, -INFINITY, -INFINITY};
  return *reinterpret_cast<const f32x4*>(base + (int64_t)e * REC);
}

// entries of the chunk that can reach the 8x8 sub-tile whose first pixel
// centre is (sx0, sy0): bit l = lane l's entry
__device__ __forceinline__ uint64_t reach_mask(const f32x4& h, float sx0,
                                               float sy0) {
  const bool hit = h[0] + h[2] >= sx0 && h[0] - h[2] <= sx0 + 7.f &&
                   h[1] + h[3] >= sy0 && h[1] - h[3] <= sy0 + 7.f;
  return __ballot(hit);
}

// G = exp(power) of the record at the pixel, as 2^(log2e power); the forward
// and the backward share it: same skip decisions
__device__ __forceinline__ float gauss_weight(const Rec& r, float pfx,
                                              float pfy, float& dx, float& dy,
                                              float& p2) {
  dx = r.a[0] - pfx;
  dy = r.a[1] - pfy;
  p2 = dx * (r.b[0] * dx + r.b[1] * dy) + r.b[2] * (dy * dy);
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
    BCam cam, const int* __restrict__ ranges, const float* __restrict__ rec,
    float* __restrict__ out_color, float* __restrict__ out_color_b,
    float* __restrict__ out_depth, float* __restrict__ final_T,
    int* __restrict__ n_contrib) {
  const int gx = (cam.W + TILE - 1) / TILE;
  const int tile = blockIdx.y * gx + blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int sx0 = blockIdx.x * TILE + (wave & 1) * 8,
            sy0 = blockIdx.y * TILE + (wave >> 1) * 8;
  const int px = sx0 + (lane & 7), py = sy0 + (lane >> 3);
  const bool inside = px < cam.W && py < cam.H;
  const float pfx = (float)px, pfy = (float)py;
  const int r0 = ranges[tile * 2], r1 = ranges[tile * 2 + 1];
  const int len = r1 - r0;
  const float* __restrict__ base = rec + (int64_t)r0 * REC;
  const int n_chunks = (len + 63) / 64;
  bool done = !inside;
  float T = 1.f, C[3] = {0.f, 0.f, 0.f}, Cb[3] = {0.f, 0.f, 0.f}, D = 0.f;
  int last = 0;
  f32x4 head = load_head(base, lane, len);
  for (int c = 0; c < n_chunks; ++c) {
    if (__ballot(!done) == 0) break;
    uint64_t m = reach_mask(head, (float)sx0, (float)sy0);
    if (c + 1 < n_chunks) head = load_head(base, (c + 1) * 64 + lane, len);
    if (m == 0) continue;
    // the next entry's record is fetched while the current one is blended
    // (a two-entry-deep variant measured the same: the loop is not bound by
    // the scalar loads)
    int j = c * 64 + __builtin_ctzll(m);
    Rec r = load_rec(base, j);
    while (true) {
      m &= m - 1;
      const int jn = m ? c * 64 + __builtin_ctzll(m) : j;
      const Rec rn = load_rec(base, jn);
      float dx, dy, p2;
      const float G = gauss_weight(r, pfx, pfy, dx, dy, p2);
      const float alpha = fminf(0.99f, r.b[3] * G);
      const float test_T = T * (1.f - alpha);
      const bool use = !done && p2 <= 0.f && alpha >= 1.f / 255.f;
      if (use && test_T < 0.0001f) done = true;
      if (use && !done) {
        const float w = alpha * T;
        C[0] += r.c[0] * w;
        C[1] += r.c[1] * w;
        C[2] += r.c[2] * w;
        D += r.c[3] * w;
        if (DUAL) {
          Cb[0] += r.d[0] * w;
          Cb[1] += r.d[1] * w;
          Cb[2] += r.d[2] * w;
        }
        T = test_T;
        last = j + 1;
      }
      if (m == 0 || __ballot(!done) == 0) break;
      j = jn;
      r = rn;
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

// ---- transposed wave reduction ------------------------------------------
// Transposed merge steps over the wave halves / the 16-lane row pairs:
// v_permlane32_swap(a, b) leaves {a.lo, b.lo} in a and {a.hi, b.hi} in b, so
// a + b is "a summed over {l, l+32}" in lanes [0,32) and the same of b in
// lanes [32,64); v_permlane16_swap does it for the row pairs.  Inline asm:
// this toolchain's __builtin_amdgcn_permlane{16,32}_swap drops the second
// result (both elements of the returned pair read the first register).
// Independent swaps are issued back to back; the s_nop cover the VALU ->
// swap and swap -> VALU wait states (inline asm is opaque to the hazard
// recogniser).
__device__ __forceinline__ void swap32x6(float (&v)[12]) {
  asm volatile(
      "s_nop 1\n"
      "v_permlane32_swap_b32 %0, %1\n"
      "v_permlane32_swap_b32 %2, %3\n"
      "v_permlane32_swap_b32 %4, %5\n"
      "v_permlane32_swap_b32 %6, %7\n"
      "v_permlane32_swap_b32 %8, %9\n"
      "v_permlane32_swap_b32 %10, %11\n"
      "s_nop 1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]),
        "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
        "+v"(v[10]), "+v"(v[11]));
}
__device__ __forceinline__ void swap16x3(float (&u)[6]) {
  asm volatile(
      "s_nop 1\n"
      "v_permlane16_swap_b32 %0, %1\n"
      "v_permlane16_swap_b32 %2, %3\n"
      "v_permlane16_swap_b32 %4, %5\n"
      "s_nop 1"
      : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]),
        "+v"(u[5]));
}

// which of the twelve sums lane l holds after merge_reduce12 (every lane of
// a quad the same; lanes with bits 2 and 3 both set repeat 8..11)
__device__ __forceinline__ int merge_slot(int lane) {
  const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1;
  return (lane & 4) ? 8 + 2 * b4 + b5 : 4 * b3 + 2 * b4 + b5;
}

// v[0..11] per lane -> the sum over the wave of v[merge_slot(lane)]
__device__ __forceinline__ float merge_reduce12(float (&v)[12], int lane) {
  float u[6], t[3];
  swap32x6(v);
#pragma unroll
  for (int k = 0; k < 6; ++k) u[k] = v[2 * k] + v[2 * k + 1];
  swap16x3(u);
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = u[2 * k] + u[2 * k + 1];
  // t[k]: slot 4k + 2 b4 + b5, summed over lane bits 4, 5
  const bool b3 = lane & 8, b2 = lane & 4;
  const float w0 = (b3 ? t[1] : t[0]) +
                   dpp_get<0x128>(b3 ? t[0] : t[1]);   // row_ror:8
  const float w1 = t[2] + dpp_get<0x128>(t[2]);
  // lanes l <-> l ^ 7 (row_half_mirror): opposite bit 2, same bits 3..5
  float z = (b2 ? w1 : w0) + dpp_get<0x141>(b2 ? w0 : w1);
  z += dpp_get<0xB1>(z);   // quad_perm [1,0,3,2]
  z += dpp_get<0x4E>(z);   // quad_perm [2,3,0,1]
  return z;
}

template <bool DUAL>
__global__ __launch_bounds__(BLOCK) void gs_blend_bwd_kernel(
    BCam cam, const int* __restrict__ ranges, const float* __restrict__ rec,
    const float* __restrict__ final_T, const int* __restrict__ n_contrib,
    const float* __restrict__ out_color, const float* __restrict__ out_color_b,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_b,
    const int* __restrict__ key_pos, float* __restrict__ key_grad) {
  __shared__ float s_acc[BUCKET * KEYROW];
  __shared__ int s_max;
  const int gx = (cam.W + TILE - 1) / TILE;
  const int tile = blockIdx.y * gx + blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = ranges[tile * 2], r1 = ranges[tile * 2 + 1];
  if (r1 <= r0) return;
  const int len = r1 - r0;
  const float* __restrict__ base = rec + (int64_t)r0 * REC;
  const int sx0 = blockIdx.x * TILE + (wave & 1) * 8,
            sy0 = blockIdx.y * TILE + (wave >> 1) * 8;
  const int px = sx0 + (lane & 7), py = sy0 + (lane >> 3);
  const bool inside = px < cam.W && py < cam.H;
  const float pfx = (float)px, pfy = (float)py;
  const int HW = cam.H * cam.W;
  if (tid == 0) s_max = 0;
#pragma unroll
  for (int k = 0; k < BUCKET * KEYROW / BLOCK; ++k)
    s_acc[k * BLOCK + tid] = 0.f;
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
  const int n_live = min(len, max_nc);
  const int n_buckets = (n_live + BUCKET - 1) / BUCKET;
  const float ddx = 0.5f * cam.W, ddy = 0.5f * cam.H;
  const int slot = merge_slot(lane);
  const bool owner = (lane & 3) == 0 && (lane & 12) != 12;
  float T = 1.f, A = 0.f;
  f32x4 head = load_head(base, lane, len);
  for (int b = 0; b < n_buckets; ++b) {
    for (int cc = 0; cc < BUCKET / 64; ++cc) {
      const int e0 = b * BUCKET + cc * 64;   // first entry of the chunk
      if (e0 >= wave_nc) break;
      uint64_t m = reach_mask(head, (float)sx0, (float)sy0);
      if (e0 + 64 < wave_nc) head = load_head(base, e0 + 64 + lane, len);
      // entries behind the wave's last contributor reach nothing
      if (wave_nc - e0 < 64) m &= (1ull << (wave_nc - e0)) - 1ull;
      if (m == 0) continue;
      int j = e0 + __builtin_ctzll(m);
      Rec r = load_rec(base, j);
      while (true) {
        m &= m - 1;
        const int jn = m ? e0 + __builtin_ctzll(m) : j;
        const Rec rn = load_rec(base, jn);
        float dx, dy, p2;
        const float G = gauss_weight(r, pfx, pfy, dx, dy, p2);
        const float alpha = fminf(0.99f, r.b[3] * G);
        const bool act = j < nc && p2 <= 0.f && alpha >= 1.f / 255.f;
        if (__ballot(act) != 0) {
          float qd = r.c[0] * dLa[0] + r.c[1] * dLa[1] + r.c[2] * dLa[2];
          if (DUAL)
            qd += r.d[0] * dLb[0] + r.d[1] * dLb[1] + r.d[2] * dLb[2];
          const float w = act ? alpha * T : 0.f;
          const float qw = qd * w;
          const float om = 1.f - alpha;
          const float dL_dalpha =
              T * qd - (R - A - qw) * __builtin_amdgcn_rcpf(om);
          const float h = act ? G * dL_dalpha : 0.f;
          const float hx = h * dx, hy = h * dy;
          float v[12] = {w * dLa[0], w * dLa[1], w * dLa[2], w * dLb[0],
                               w * dLb[1], w * dLb[2], h,          hx,
                               hy,         hx * dx,    hx * dy,    hy * dy};
          const float z = merge_reduce12(v, lane);
          if (owner) atomicAdd(s_acc + (j - b * BUCKET) * KEYROW + slot, z);
          T = act ? T * om : T;
          A += qw;
        }
        if (m == 0) break;
        j = jn;
        r = rn;
      }
    }
    __syncthreads();
    // finished rows of the bucket: sums -> gradients w.r.t. colours, mean2D
    // (ndc), conic (true partials), opacity; the row is cleared for the next
    // bucket
    {
      const int k = b * BUCKET + tid;
      if (k < len) {
        float s[KEYROW];
#pragma unroll
        for (int c = 0; c < KEYROW; ++c) {
          s[c] = s_acc[tid * KEYROW + c];
          s_acc[tid * KEYROW + c] = 0.f;
        }
        const f32x4 q = *reinterpret_cast<const f32x4*>(
            base + (int64_t)k * REC + 4);
        const float o = q[3];
        const float cA = q[0] * (-2.f / kLog2e), cB = q[1] * (-1.f / kLog2e),
                    cC = q[2] * (-2.f / kLog2e);
        float* row = key_grad + (int64_t)key_pos[r0 + k] * KEYROW;
        *reinterpret_cast<f32x4*>(row) = f32x4{s[0], s[1], s[2], s[3]};
        *reinterpret_cast<f32x4*>(row + 4) =
            f32x4{s[4], s[5], -o * (cA * s[7] + cB * s[8]) * ddx,
                  -o * (cC * s[8] + cB * s[7]) * ddy};
        *reinterpret_cast<f32x4*>(row + 8) =
            f32x4{-0.5f * o * s[9], -o * s[10], -0.5f * o * s[11], s[6]};
      }
    }
    __syncthreads();
  }
  // keys of this tile behind the last contributor: zero rows
  for (int k = n_buckets * BUCKET + tid; k < len; k += BLOCK) {
    float* row = key_grad + (int64_t)key_pos[r0 + k] * KEYROW;
#pragma unroll
    for (int c = 0; c < KEYROW; c += 4)
      *reinterpret_cast<f32x4*>(row + c) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// gradients of Gaussian i = sum of the rows of its pairs: rows are filed
// under the pre-sort pair index, a Gaussian's are [offsets[i-1], offsets[i])
// (pairs numbered behind the capacity were dropped by gs_bin.hip)
template <bool DUAL>
__global__ __launch_bounds__(256) void gs_key_reduce_kernel(
    int n, int64_t cap, const int64_t* __restrict__ offsets,
    const int* __restrict__ live_pre, const float* __restrict__ key_grad,
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
    const float* row = key_grad + k * KEYROW;
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
  return (key_capacity + 1) * REC;  // one packed record per key
}

int xrd_gs_blend_fwd(const xrd_gs_camera* c, const int32_t* ranges,
                     const int32_t* point_list, const float* xy,
                     const float* colors_a, const float* colors_b,
                     const float* conic_opacity, const float* depths,
                     float* out_color_a, float* out_color_b, float* out_depth,
                     float* final_T, int32_t* n_contrib, float* ckpt,
                     xrd_stream_t stream) {
  BCam cam;
  int rc = to_bcam(c, cam);
  if (rc) return rc;
  if (!ranges || !out_color_a || !out_depth || !final_T || !n_contrib ||
      !ckpt)
    return XRD_ERR_ARG;
  if ((colors_b == nullptr) != (out_color_b == nullptr)) return XRD_ERR_ARG;
  const dim3 grid((cam.W + TILE - 1) / TILE, (cam.H + TILE - 1) / TILE);
  hipStream_t st = (hipStream_t)stream;
  if (point_list) {
    if (!xy || !colors_a || !conic_opacity || !depths) return XRD_ERR_ARG;
    hipLaunchKernelGGL(gs_pack_kernel, dim3(grid.x * grid.y), dim3(256), 0, st,
                       ranges, point_list, xy, conic_opacity, colors_a,
                       colors_b, depths, ckpt);
  }
  if (colors_b)
    hipLaunchKernelGGL(gs_blend_fwd_kernel<true>, grid, dim3(BLOCK), 0, st,
                       cam, ranges, ckpt, out_color_a, out_color_b, out_depth,
                       final_T, n_contrib);
  else
    hipLaunchKernelGGL(gs_blend_fwd_kernel<false>, grid, dim3(BLOCK), 0, st,
                       cam, ranges, ckpt, out_color_a, nullptr, out_depth,
                       final_T, n_contrib);
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
  BCam cam;
  int rc = to_bcam(c, cam);
  if (rc) return rc;
  if (n < 0 || key_capacity < 1) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!ranges || !point_list || !key_pos || !offsets || !xy ||
      !conic_opacity || !colors_a || !final_T || !n_contrib || !out_color_a ||
      !dL_dcolor_a || !ckpt || !key_grad || !dL_dmean2D || !dL_dconic ||
      !dL_dopacity || !dL_dcolors_a)
    return XRD_ERR_ARG;
  const bool dual = colors_b != nullptr;
  if (dual && (!out_color_b || !dL_dcolor_b || !dL_dcolors_b))
    return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((cam.W + TILE - 1) / TILE, (cam.H + TILE - 1) / TILE);
  if (dual) {
    hipLaunchKernelGGL(gs_blend_bwd_kernel<true>, grid, dim3(BLOCK), 0, st,
                       cam, ranges, ckpt, final_T, n_contrib, out_color_a,
                       out_color_b, dL_dcolor_a, dL_dcolor_b, key_pos,
                       key_grad);
    hipLaunchKernelGGL(gs_key_reduce_kernel<true>, dim3((n + 255) / 256),
                       dim3(256), 0, st, n, key_capacity, offsets,
                       key_pos + key_capacity, key_grad, dL_dmean2D, dL_dconic,
                       dL_dopacity, dL_dcolors_a, dL_dcolors_b);
  } else {
    hipLaunchKernelGGL(gs_blend_bwd_kernel<false>, grid, dim3(BLOCK), 0, st,
                       cam, ranges, ckpt, final_T, n_contrib, out_color_a,
                       nullptr, dL_dcolor_a, nullptr, key_pos, key_grad);
    hipLaunchKernelGGL(gs_key_reduce_kernel<false>, dim3((n + 255) / 256),
                       dim3(256), 0, st, n, key_capacity, offsets,
                       key_pos + key_capacity, key_grad, dL_dmean2D, dL_dconic,
                       dL_dopacity, dL_dcolors_a, nullptr);
  }
  return check_launch("xrd_gs_blend_bwd");
}

}  // extern "C"
