// Tile blend of the 3-D Gaussian rasteriser, second formulation (gfx950):
// forward with per-bucket checkpoints, backward with one LANE PER GAUSSIAN.
//
// Replaces the blend part of the unvendored CUDA dependency
// diff-gaussian-rasterization-w-depth @ cb65e4b (reference call sites
// slam/model_components/gaussian_cloud_splatam.py:63-69,267-268; algorithm per
// SURVEY.md App. C.3, oracle: oracle/gs_oracle.py, parity unpinned).
//
// Why.  The published backward (and round 2's gs_render_bwd) runs one thread
// per PIXEL: every Gaussian of a tile needs its nine to twelve gradient values
// reduced over the pixels — 64-lane reductions + atomics per Gaussian and wave
// (measured: 72 of ~150 instructions per Gaussian and wave, 3.3 % of the fp32
// peak).  Here a wave owns a BUCKET of 64 consecutive Gaussians of a tile's
// depth-sorted list, one per lane, and the tile's 256 pixels stream through
// the lanes like through a systolic array: at step s lane l blends pixel
// s - l, takes the pixel's running state (transmittance T and the prefix A of
// sum_j (c_j . dL/dC) alpha_j T_j) from lane l - 1 with one DPP wave shift and
// hands its own to lane l + 1.  Every lane accumulates the gradient of ITS
// Gaussian in registers over the 256 pixels: no reduction, no atomic.  The
// state a bucket starts from (T and the colour prefix sums of every pixel in
// front of the bucket's first Gaussian) is a checkpoint the forward leaves
// every 64 Gaussians (7 floats a pixel and bucket).  With
//   dL/dalpha_i = T_i q_i - (Q - A_i - q_i alpha_i T_i + T_final (bg . dL/dC))
//                 / (1 - alpha_i),  q_i = c_i . dL/dC,  Q = sum_j q_j alpha_j T_j
// (the published back-to-front recurrence, rearranged front-to-back: Q is the
// pixel's rendered colour without background dotted with dL/dC), a lane needs
// nothing from the Gaussians behind it.  One gradient row per (Gaussian, tile)
// key is written; gs_key_reduce sums a Gaussian's rows through the inverse
// map of the binning sort (gs_bin.hip) — no atomics anywhere.
#include <hip/hip_runtime.h>

#include "common.h"

namespace xrd {
namespace {

constexpr int TILE = 16;
constexpr int BLOCK = TILE * TILE;
constexpr int BUCKET = 64;
constexpr int KEYROW = 12;  // col a 3, col b 3, mean2D 2, conic 3, opacity 1

struct BCam {
  int H, W;
  float bg[3];
};

// checkpoint slot of bucket b of a tile whose list starts at r0: the tiles'
// lists are consecutive ranges of the sorted key list, so floor(r0 / 64) +
// tile + b never collides (every tile adds at most one partial bucket)
__device__ __forceinline__ int64_t ckpt_slot(int r0, int tile, int b) {
  return (int64_t)(r0 / BUCKET) + tile + b;
}

// DUAL: a second colour set blended with the same weights (SplaTAM renders rgb
// and (z, 1, z^2) with identical geometry).  CK floats per pixel and bucket:
// T, Ca[3] (, Cb[3]) — component-major [CK][256] per slot (coalesced).
template <bool DUAL>
__global__ __launch_bounds__(BLOCK) void gs_blend_fwd_kernel(
    BCam cam, const int* __restrict__ ranges, const int* __restrict__ plist,
    const float* __restrict__ xy, const float* __restrict__ colors,
    const float* __restrict__ colors_b, const float* __restrict__ conic_o,
    const float* __restrict__ depths, float* __restrict__ out_color,
    float* __restrict__ out_color_b, float* __restrict__ out_depth,
    float* __restrict__ final_T, int* __restrict__ n_contrib,
    float* __restrict__ ckpt) {
  constexpr int CK = DUAL ? 7 : 4;
  __shared__ float2 s_xy[BLOCK];
  __shared__ f32x4 s_co[BLOCK];
  __shared__ f32x4 s_cd[BLOCK];  // r,g,b,depth
  __shared__ f32x4 s_cb[DUAL ? BLOCK : 1];
  const int gx = (cam.W + TILE - 1) / TILE;
  const int tile = blockIdx.y * gx + blockIdx.x;
  const int tid = threadIdx.y * TILE + threadIdx.x;
  const int px = blockIdx.x * TILE + threadIdx.x,
            py = blockIdx.y * TILE + threadIdx.y;
  const bool inside = px < cam.W && py < cam.H;
  const float pfx = (float)px, pfy = (float)py;
  const int r0 = ranges[tile * 2], r1 = ranges[tile * 2 + 1];
  const int rounds = (r1 - r0 + BLOCK - 1) / BLOCK;
  int todo = r1 - r0;
  bool done = !inside;
  float T = 1.f, C[3] = {0.f, 0.f, 0.f}, Cb[3] = {0.f, 0.f, 0.f}, D = 0.f;
  int contributor = 0, last = 0;
  for (int rd = 0; rd < rounds; ++rd, todo -= BLOCK) {
    if (__syncthreads_count(done) == BLOCK) break;
    const int prog = rd * BLOCK + tid;
    if (r0 + prog < r1) {
      const int g = plist[r0 + prog];
      s_xy[tid] = make_float2(xy[g * 2], xy[g * 2 + 1]);
      s_co[tid] = *reinterpret_cast<const f32x4*>(conic_o + g * 4);
      s_cd[tid] = f32x4{colors[g * 3], colors[g * 3 + 1], colors[g * 3 + 2],
                        depths[g]};
      if (DUAL)
        s_cb[tid] = f32x4{colors_b[g * 3], colors_b[g * 3 + 1],
                          colors_b[g * 3 + 2], 0.f};
    }
    __syncthreads();
    const int nj = min(BLOCK, todo);
    for (int j = 0; j < nj; ++j) {
      if (ckpt != nullptr && (j & (BUCKET - 1)) == 0) {
        // state in front of Gaussian rd*256 + j (every thread of the tile,
        // finished pixels included: their value is never used)
        float* ck = ckpt + ckpt_slot(r0, tile, (rd * BLOCK + j) / BUCKET) *
                               (CK * BLOCK);
        ck[tid] = T;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          ck[(1 + ch) * BLOCK + tid] = C[ch];
          if (DUAL) ck[(4 + ch) * BLOCK + tid] = Cb[ch];
        }
      }
      if (done) continue;
      ++contributor;
      const float dx = s_xy[j].x - pfx, dy = s_xy[j].y - pfy;
      const f32x4 co = s_co[j];
      const float power =
          -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
      if (power > 0.f) continue;
      const float alpha = fminf(0.99f, co[3] * expf(power));
      if (alpha < 1.f / 255.f) continue;
      const float test_T = T * (1.f - alpha);
      if (test_T < 0.0001f) {
        done = true;
        continue;
      }
      const f32x4 cd = s_cd[j];
      const float w = alpha * T;
      C[0] += cd[0] * w;
      C[1] += cd[1] * w;
      C[2] += cd[2] * w;
      D += cd[3] * w;
      if (DUAL) {
        const f32x4 cb = s_cb[j];
        Cb[0] += cb[0] * w;
        Cb[1] += cb[1] * w;
        Cb[2] += cb[2] * w;
      }
      T = test_T;
      last = contributor;
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

// value of lane l - 1 (lane 0 keeps `first`): one DPP wave shift
__device__ __forceinline__ float from_prev_lane(float v, float first) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first),
                                         __builtin_bit_cast(int, v),
                                         0x138 /* wave_shr:1 */, 0xf, 0xf,
                                         false));
}

constexpr int BW = 8;  // waves (buckets in flight) per block

// per-pixel constants of the tile in LDS: [256][8] =
//   dLa[3], dLb[3] (0 unless DUAL), R = Q + T_final (bg . dL), n_contrib
template <bool DUAL>
__global__ __launch_bounds__(BW * 64) void gs_blend_bwd_kernel(
    BCam cam, const int* __restrict__ ranges, const int* __restrict__ plist,
    const float* __restrict__ xy, const float* __restrict__ conic_o,
    const float* __restrict__ colors, const float* __restrict__ colors_b,
    const float* __restrict__ final_T, const int* __restrict__ n_contrib,
    const float* __restrict__ out_color, const float* __restrict__ out_color_b,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_b,
    const float* __restrict__ ckpt, float* __restrict__ key_grad) {
  constexpr int CK = DUAL ? 7 : 4;
  __shared__ __attribute__((aligned(16))) float s_pix[BLOCK * 8];
  __shared__ float s_T0[BW][BLOCK], s_A0[BW][BLOCK];
  __shared__ int s_max;
  const int gx = (cam.W + TILE - 1) / TILE;
  const int tile = blockIdx.y * gx + blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = ranges[tile * 2], r1 = ranges[tile * 2 + 1];
  if (r1 <= r0) return;
  const int HW = cam.H * cam.W;
  if (tid == 0) s_max = 0;
  __syncthreads();
  if (tid < BLOCK) {
    const int px = blockIdx.x * TILE + (tid & 15),
              py = blockIdx.y * TILE + (tid >> 4);
    const bool inside = px < cam.W && py < cam.H;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int nc = 0;
    if (inside) {
      const int pix = py * cam.W + px;
      const float Tf = final_T[pix];
      float R = 0.f, bgdot = 0.f;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float d = dL_dpix[ch * HW + pix];
        v[ch] = d;
        bgdot += cam.bg[ch] * d;
        R += (out_color[ch * HW + pix] - Tf * cam.bg[ch]) * d;
        if (DUAL) {
          const float db = dL_dpix_b[ch * HW + pix];
          v[3 + ch] = db;
          bgdot += cam.bg[ch] * db;
          R += (out_color_b[ch * HW + pix] - Tf * cam.bg[ch]) * db;
        }
      }
      v[6] = R + Tf * bgdot;
      nc = n_contrib[pix];
    }
    v[7] = __int_as_float(nc);
    *reinterpret_cast<f32x4*>(s_pix + tid * 8) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(s_pix + tid * 8 + 4) =
        f32x4{v[4], v[5], v[6], v[7]};
    atomicMax(&s_max, nc);
  }
  __syncthreads();
  const int max_nc = s_max;  // Gaussians behind it contribute to no pixel
  const int n_buckets = (min(r1 - r0, max_nc) + BUCKET - 1) / BUCKET;
  const float ddx = 0.5f * cam.W, ddy = 0.5f * cam.H;
  const float tx0 = (float)(blockIdx.x * TILE), ty0 = (float)(blockIdx.y * TILE);
  for (int b = wave; b < n_buckets; b += BW) {
    // the bucket's entry state of every pixel: T and A = Ca . dLa + Cb . dLb
    const float* ck = ckpt + ckpt_slot(r0, tile, b) * (CK * BLOCK);
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) {
      const int p = k * 64 + lane;
      const float* pc = s_pix + p * 8;
      float A = 0.f;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        A += ck[(1 + ch) * BLOCK + p] * pc[ch];
        if (DUAL) A += ck[(4 + ch) * BLOCK + p] * pc[3 + ch];
      }
      s_T0[wave][p] = ck[p];
      s_A0[wave][p] = A;
    }
    wave_lds_sync();
    const int gidx = b * BUCKET + lane;  // index in the tile's list
    const bool have = r0 + gidx < r1;
    const int g = have ? plist[r0 + gidx] : 0;
    const float gxp = have ? xy[g * 2] : 0.f, gyp = have ? xy[g * 2 + 1] : 0.f;
    const f32x4 co = have ? *reinterpret_cast<const f32x4*>(conic_o + g * 4)
                          : f32x4{0.f, 0.f, 0.f, 0.f};
    float ca[3] = {0.f, 0.f, 0.f}, cb[3] = {0.f, 0.f, 0.f};
    if (have) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        ca[ch] = colors[g * 3 + ch];
        if (DUAL) cb[ch] = colors_b[g * 3 + ch];
      }
    }
    float g_ca[3] = {0.f, 0.f, 0.f}, g_cb[3] = {0.f, 0.f, 0.f};
    float g_m[2] = {0.f, 0.f}, g_con[3] = {0.f, 0.f, 0.f}, g_op = 0.f;
    float T_out = 1.f, A_out = 0.f;
#pragma unroll 2
    for (int s = 0; s < BLOCK + BUCKET - 1; ++s) {
      const int p = s - lane;
      const bool pv = p >= 0 && p < BLOCK;
      const int pc_i = pv ? p : 0;
      // lane 0 starts pixel s from the checkpoint, the others continue what
      // lane l - 1 left one step ago
      const float T_in = from_prev_lane(T_out, s_T0[wave][s < BLOCK ? s : 0]);
      const float A_in = from_prev_lane(A_out, s_A0[wave][s < BLOCK ? s : 0]);
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(s_pix + pc_i * 8);
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(s_pix + pc_i * 8 + 4);
      T_out = T_in;
      A_out = A_in;
      const int nc = __float_as_int(c1[3]);
      bool act = pv && have && gidx < nc;
      const float dx = gxp - (tx0 + (float)(pc_i & 15));
      const float dy = gyp - (ty0 + (float)(pc_i >> 4));
      const float power =
          -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
      const float G = expf(power);  // as the forward: same skip decisions
      float alpha = fminf(0.99f, co[3] * G);
      act = act && power <= 0.f && alpha >= 1.f / 255.f;
      if (act) {
        const float w = alpha * T_in;
        float q = ca[0] * c0[0] + ca[1] * c0[1] + ca[2] * c0[2];
        if (DUAL) q += cb[0] * c0[3] + cb[1] * c1[0] + cb[2] * c1[1];
        const float qw = q * w;
        const float dL_dalpha =
            T_in * q - (c1[2] - A_in - qw) / (1.f - alpha);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) g_ca[ch] += w * c0[ch];
        if (DUAL) {
          g_cb[0] += w * c0[3];
          g_cb[1] += w * c1[0];
          g_cb[2] += w * c1[1];
        }
        const float dL_dG = co[3] * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddelx = -gdx * co[0] - gdy * co[1];
        const float dG_ddely = -gdy * co[2] - gdx * co[1];
        g_m[0] += dL_dG * dG_ddelx * ddx;
        g_m[1] += dL_dG * dG_ddely * ddy;
        g_con[0] += -0.5f * gdx * dx * dL_dG;
        g_con[1] += -gdx * dy * dL_dG;
        g_con[2] += -0.5f * gdy * dy * dL_dG;
        g_op += G * dL_dalpha;
        T_out = T_in * (1.f - alpha);
        A_out = A_in + qw;
      }
    }
    if (have) {
      float* row = key_grad + (int64_t)(r0 + gidx) * KEYROW;
      row[0] = g_ca[0];
      row[1] = g_ca[1];
      row[2] = g_ca[2];
      row[3] = g_cb[0];
      row[4] = g_cb[1];
      row[5] = g_cb[2];
      row[6] = g_m[0];
      row[7] = g_m[1];
      row[8] = g_con[0];
      row[9] = g_con[1];
      row[10] = g_con[2];
      row[11] = g_op;
    }
    wave_lds_sync();  // s_T0 / s_A0 of this wave are rewritten
  }
  // keys of this tile behind the last contributor: zero rows
  for (int k = n_buckets * BUCKET + tid; k < r1 - r0; k += BW * 64) {
    float* row = key_grad + (int64_t)(r0 + k) * KEYROW;
#pragma unroll
    for (int c = 0; c < KEYROW; ++c) row[c] = 0.f;
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
  const int64_t nt = (int64_t)((image_width + TILE - 1) / TILE) *
                     ((image_height + TILE - 1) / TILE);
  return (key_capacity / BUCKET + nt + 2) * 7 * BLOCK;
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
  if (!ranges || !out_color_a || !out_depth || !final_T || !n_contrib)
    return XRD_ERR_ARG;
  if ((colors_b == nullptr) != (out_color_b == nullptr)) return XRD_ERR_ARG;
  const dim3 grid((cam.W + TILE - 1) / TILE, (cam.H + TILE - 1) / TILE);
  if (colors_b)
    hipLaunchKernelGGL(gs_blend_fwd_kernel<true>, grid, dim3(TILE, TILE), 0,
                       (hipStream_t)stream, cam, ranges, point_list, xy,
                       colors_a, colors_b, conic_opacity, depths, out_color_a,
                       out_color_b, out_depth, final_T, n_contrib, ckpt);
  else
    hipLaunchKernelGGL(gs_blend_fwd_kernel<false>, grid, dim3(TILE, TILE), 0,
                       (hipStream_t)stream, cam, ranges, point_list, xy,
                       colors_a, nullptr, conic_opacity, depths, out_color_a,
                       nullptr, out_depth, final_T, n_contrib, ckpt);
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
    hipLaunchKernelGGL(gs_blend_bwd_kernel<true>, grid, dim3(BW * 64), 0, st,
                       cam, ranges, point_list, xy, conic_opacity, colors_a,
                       colors_b, final_T, n_contrib, out_color_a, out_color_b,
                       dL_dcolor_a, dL_dcolor_b, ckpt, key_grad);
    hipLaunchKernelGGL(gs_key_reduce_kernel<true>, dim3((n + 255) / 256),
                       dim3(256), 0, st, n, key_capacity, offsets, key_pos,
                       key_grad, dL_dmean2D, dL_dconic, dL_dopacity,
                       dL_dcolors_a, dL_dcolors_b);
  } else {
    hipLaunchKernelGGL(gs_blend_bwd_kernel<false>, grid, dim3(BW * 64), 0, st,
                       cam, ranges, point_list, xy, conic_opacity, colors_a,
                       nullptr, final_T, n_contrib, out_color_a, nullptr,
                       dL_dcolor_a, nullptr, ckpt, key_grad);
    hipLaunchKernelGGL(gs_key_reduce_kernel<false>, dim3((n + 255) / 256),
                       dim3(256), 0, st, n, key_capacity, offsets, key_pos,
                       key_grad, dL_dmean2D, dL_dconic, dL_dopacity,
                       dL_dcolors_a, nullptr);
  }
  return check_launch("xrd_gs_blend_bwd");
}

}  // extern "C"
