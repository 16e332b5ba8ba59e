// Tile-based 3-D Gaussian rasteriser (forward + backward) for SplaTAM on
// gfx950.  Replaces the unvendored CUDA dependency
// diff-gaussian-rasterization-w-depth @ cb65e4b (reference call sites
// slam/model_components/gaussian_cloud_splatam.py:63-69,267-268,
// slam/common/common.py:592-619).  Algorithm per SURVEY.md App. C.3 (oracle:
// oracle/gs_oracle.py; parity unpinned by the reference): EWA projection with
// the 0.3 low-pass, 3-sigma radius, 16x16 tiles, (tile | depth) keys,
// front-to-back alpha blending with the 1/255 skip and the 1e-4 stop.
//
// gfx950 mapping: a 16x16 tile = 256 threads = 4 waves of 64; Gaussians of a
// tile are staged through LDS in batches of 256; the backward reduces every
// per-Gaussian gradient over the 64 pixels of a wave with shuffles before ONE
// atomic per wave and value (the CUDA original issues one atomic per pixel).
// The scan and the 64-bit key sort between the phases are left to the caller
// (rocPRIM through torch.cumsum / torch.sort).
#include "common.h"

namespace xrd {
namespace {

constexpr int TILE = 16;
constexpr int BLOCK = TILE * TILE;

struct Cam {
  int H, W;
  float tanfovx, tanfovy;
  float bg[3];
  float scale_modifier;
  float view[16];  // transposed w2c (column-major), as the reference passes
  float proj[16];  // transposed full projection
};

__device__ __forceinline__ void quat_rot(const float* q, float (&R)[3][3]) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1.f - 2.f * (y * y + z * z);
  R[0][1] = 2.f * (x * y - r * z);
  R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z);
  R[1][1] = 1.f - 2.f * (x * x + z * z);
  R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y);
  R[2][1] = 2.f * (y * z + r * x);
  R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// everything the projection of one Gaussian needs, shared by fwd and bwd
struct Proj {
  float pv[3];        // view-space position
  float hom[4];       // clip-space
  float m_w;          // 1/(w+1e-7)
  float R[3][3];
  float S2[3];        // squared scales
  float Sig[3][3];    // 3-D covariance
  float tx, ty, tz;   // clamped view position used by the Jacobian
  float xmul, ymul;
  float fx, fy;
  float T[2][3];      // J * W
  float a, b, c;      // 2-D covariance (+0.3)
};

__device__ __forceinline__ void project(const Cam& cam, const float* p,
                                        const float* s, const float* q,
                                        Proj& P) {
  const float* V = cam.view;
  const float* M = cam.proj;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    P.pv[i] = V[i] * p[0] + V[4 + i] * p[1] + V[8 + i] * p[2] + V[12 + i];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    P.hom[i] = M[i] * p[0] + M[4 + i] * p[1] + M[8 + i] * p[2] + M[12 + i];
  P.m_w = 1.f / (P.hom[3] + 1e-7f);
  quat_rot(q, P.R);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float sk = s[k] * cam.scale_modifier;
    P.S2[k] = sk * sk;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) v += P.R[i][k] * P.S2[k] * P.R[j][k];
      P.Sig[i][j] = v;
    }
  P.fx = cam.W / (2.f * cam.tanfovx);
  P.fy = cam.H / (2.f * cam.tanfovy);
  const float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
  P.tz = P.pv[2];
  const float txtz = P.pv[0] / P.tz, tytz = P.pv[1] / P.tz;
  P.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  P.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  P.tx = fminf(limx, fmaxf(-limx, txtz)) * P.tz;
  P.ty = fminf(limy, fmaxf(-limy, tytz)) * P.tz;
  const float J00 = P.fx / P.tz, J02 = -P.fx * P.tx / (P.tz * P.tz);
  const float J11 = P.fy / P.tz, J12 = -P.fy * P.ty / (P.tz * P.tz);
  // W[i][j] = w2c[i][j] = view[j*4+i]
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    P.T[0][j] = J00 * V[j * 4 + 0] + J02 * V[j * 4 + 2];
    P.T[1][j] = J11 * V[j * 4 + 1] + J12 * V[j * 4 + 2];
  }
  float TS[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      TS[i][j] = P.T[i][0] * P.Sig[0][j] + P.T[i][1] * P.Sig[1][j] +
                 P.T[i][2] * P.Sig[2][j];
  P.a = TS[0][0] * P.T[0][0] + TS[0][1] * P.T[0][1] + TS[0][2] * P.T[0][2] + 0.3f;
  P.b = TS[0][0] * P.T[1][0] + TS[0][1] * P.T[1][1] + TS[0][2] * P.T[1][2];
  P.c = TS[1][0] * P.T[1][0] + TS[1][1] * P.T[1][1] + TS[1][2] * P.T[1][2] + 0.3f;
}

__global__ __launch_bounds__(256) void gs_preprocess_kernel(
    Cam cam, int n, const float* __restrict__ means, const float* __restrict__ scales,
    const float* __restrict__ rots, const float* __restrict__ opac,
    float* __restrict__ depths, float* __restrict__ xy,
    float* __restrict__ conic_o, int* __restrict__ radii,
    int* __restrict__ rect, int* __restrict__ tiles) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  radii[i] = 0;
  tiles[i] = 0;
  rect[i * 4 + 0] = rect[i * 4 + 1] = rect[i * 4 + 2] = rect[i * 4 + 3] = 0;
  // every output row is written: callers hand over uninitialised buffers
  depths[i] = 0.f;
  xy[i * 2 + 0] = xy[i * 2 + 1] = 0.f;
  conic_o[i * 4 + 0] = conic_o[i * 4 + 1] = conic_o[i * 4 + 2] =
      conic_o[i * 4 + 3] = 0.f;
  Proj P;
  project(cam, means + i * 3, scales + i * 3, rots + i * 4, P);
  if (P.pv[2] <= 0.2f) return;
  const float det = P.a * P.c - P.b * P.b;
  if (det == 0.f) return;
  const float inv = 1.f / det;
  const float mid = 0.5f * (P.a + P.c);
  const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  const float rad = ceilf(3.f * sqrtf(lam));
  const float px = ((P.hom[0] * P.m_w + 1.f) * cam.W - 1.f) * 0.5f;
  const float py = ((P.hom[1] * P.m_w + 1.f) * cam.H - 1.f) * 0.5f;
  const int gx = (cam.W + TILE - 1) / TILE, gy = (cam.H + TILE - 1) / TILE;
  const int x0 = min(gx, max(0, (int)((px - rad) / TILE)));
  const int y0 = min(gy, max(0, (int)((py - rad) / TILE)));
  const int x1 = min(gx, max(0, (int)((px + rad + TILE - 1) / TILE)));
  const int y1 = min(gy, max(0, (int)((py + rad + TILE - 1) / TILE)));
  if ((x1 - x0) * (y1 - y0) == 0) return;
  depths[i] = P.pv[2];
  radii[i] = (int)rad;
  xy[i * 2 + 0] = px;
  xy[i * 2 + 1] = py;
  conic_o[i * 4 + 0] = P.c * inv;
  conic_o[i * 4 + 1] = -P.b * inv;
  conic_o[i * 4 + 2] = P.a * inv;
  conic_o[i * 4 + 3] = opac[i];
  rect[i * 4 + 0] = x0;
  rect[i * 4 + 1] = y0;
  rect[i * 4 + 2] = x1;
  rect[i * 4 + 3] = y1;
  tiles[i] = (x1 - x0) * (y1 - y0);
}

// tile-band sharding (multi-GPU mapping, SURVEY 8e: each rank rasterises a band
// of tile rows): clip every Gaussian's tile rectangle to rows [ty0, ty1) and
// recount its tiles; everything after (binning, blend, backward) sees only the
// band's pairs.  radii keep the full-image value (visibility statistics).
__global__ __launch_bounds__(256) void gs_band_clip_kernel(
    int n, int ty0, int ty1, int* __restrict__ rect, int* __restrict__ tiles) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x0 = rect[i * 4], x1 = rect[i * 4 + 2];
  const int y0 = max(rect[i * 4 + 1], ty0), y1 = min(rect[i * 4 + 3], ty1);
  const int cnt = y1 > y0 ? (x1 - x0) * (y1 - y0) : 0;
  rect[i * 4 + 1] = cnt ? y0 : 0;
  rect[i * 4 + 3] = cnt ? y1 : 0;
  if (!cnt) rect[i * 4] = rect[i * 4 + 2] = 0;
  tiles[i] = cnt;
}

__global__ __launch_bounds__(256) void gs_duplicate_kernel(
    int n, const int* __restrict__ rect, const int64_t* __restrict__ offsets,
    const float* __restrict__ depths, int grid_x, int64_t* __restrict__ keys,
    int* __restrict__ values) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x0 = rect[i * 4], y0 = rect[i * 4 + 1], x1 = rect[i * 4 + 2],
            y1 = rect[i * 4 + 3];
  if ((x1 - x0) * (y1 - y0) == 0) return;
  int64_t off = (i == 0) ? 0 : offsets[i - 1];
  const uint32_t dbits = __float_as_uint(depths[i]);
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      const int64_t key = ((int64_t)(y * grid_x + x) << 32) | (int64_t)dbits;
      keys[off] = key;
      values[off] = i;
      ++off;
    }
}

__global__ void gs_ranges_kernel(int64_t L, const int64_t* __restrict__ keys,
                                 int* __restrict__ ranges) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const int tile = (int)(keys[i] >> 32);
  if (i == 0)
    ranges[tile * 2] = 0;
  else {
    const int prev = (int)(keys[i - 1] >> 32);
    if (prev != tile) {
      ranges[prev * 2 + 1] = (int)i;
      ranges[tile * 2] = (int)i;
    }
  }
  if (i == L - 1) ranges[tile * 2 + 1] = (int)L;
}

// DUAL: a second colour set blended with the SAME weights in the same pass.
// SplaTAM renders every view twice with identical geometry — rgb, then
// (z, 1, z^2) for depth / silhouette / depth^2
// (gaussian_cloud_splatam.py:63-69) — and the weights, the tile lists and
// the transmittance are the expensive part.
template <bool DUAL>
__global__ __launch_bounds__(BLOCK) void gs_render_fwd_kernel(
    Cam cam, const int* __restrict__ ranges, const int* __restrict__ plist,
    const float* __restrict__ xy, const float* __restrict__ colors,
    const float* __restrict__ colors_b, const float* __restrict__ conic_o,
    const float* __restrict__ depths, float* __restrict__ out_color,
    float* __restrict__ out_color_b, float* __restrict__ out_depth,
    float* __restrict__ final_T, int* __restrict__ n_contrib) {
  __shared__ float2 s_xy[BLOCK];
  __shared__ f32x4 s_co[BLOCK];
  __shared__ f32x4 s_cd[BLOCK];  // r,g,b,depth
  __shared__ f32x4 s_cb[DUAL ? BLOCK : 1];
  const int gx = (cam.W + TILE - 1) / TILE;
  const int tile = blockIdx.y * gx + blockIdx.x;
  const int tid = threadIdx.y * TILE + threadIdx.x;
  const int px = blockIdx.x * TILE + threadIdx.x, py = blockIdx.y * TILE + threadIdx.y;
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
    for (int j = 0; !done && j < min(BLOCK, todo); ++j) {
      ++contributor;
      const float dx = s_xy[j].x - pfx, dy = s_xy[j].y - pfy;
      const f32x4 co = s_co[j];
      const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
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
      if (DUAL) out_color_b[ch * cam.H * cam.W + pix] = Cb[ch] + T * cam.bg[ch];
    }
    out_depth[pix] = D;
  }
}

template <bool DUAL>
__global__ __launch_bounds__(BLOCK) void gs_render_bwd_kernel(
    Cam cam, const int* __restrict__ ranges, const int* __restrict__ plist,
    const float* __restrict__ xy, const float* __restrict__ conic_o,
    const float* __restrict__ colors, const float* __restrict__ colors_b,
    const float* __restrict__ final_T, const int* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dpix_b,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dopac, float* __restrict__ dL_dcolors,
    float* __restrict__ dL_dcolors_b) {
  __shared__ int s_id[BLOCK];
  __shared__ float2 s_xy[BLOCK];
  __shared__ f32x4 s_co[BLOCK];
  __shared__ f32x4 s_cd[BLOCK];
  __shared__ f32x4 s_cb[DUAL ? BLOCK : 1];
  const int gx = (cam.W + TILE - 1) / TILE;
  const int tile = blockIdx.y * gx + blockIdx.x;
  const int tid = threadIdx.y * TILE + threadIdx.x;
  const int lane = tid & 63;
  const int px = blockIdx.x * TILE + threadIdx.x, py = blockIdx.y * TILE + threadIdx.y;
  const bool inside = px < cam.W && py < cam.H;
  const float pfx = (float)px, pfy = (float)py;
  const int r0 = ranges[tile * 2], r1 = ranges[tile * 2 + 1];
  const int rounds = (r1 - r0 + BLOCK - 1) / BLOCK;
  int todo = r1 - r0;
  const int pix = py * cam.W + px;
  const float T_final = inside ? final_T[pix] : 0.f;
  float T = T_final;
  int contributor = todo;
  const int last_contributor = inside ? n_contrib[pix] : 0;
  float accum[3] = {0.f, 0.f, 0.f}, last_color[3] = {0.f, 0.f, 0.f};
  float accum_b[3] = {0.f, 0.f, 0.f}, last_color_b[3] = {0.f, 0.f, 0.f};
  float last_alpha = 0.f, dpx[3] = {0.f, 0.f, 0.f};
  float dpx_b[3] = {0.f, 0.f, 0.f};
  if (inside) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      dpx[ch] = dL_dpix[ch * cam.H * cam.W + pix];
      if (DUAL) dpx_b[ch] = dL_dpix_b[ch * cam.H * cam.W + pix];
    }
  }
  float bg_dot = cam.bg[0] * dpx[0] + cam.bg[1] * dpx[1] + cam.bg[2] * dpx[2];
  if (DUAL)
    bg_dot += cam.bg[0] * dpx_b[0] + cam.bg[1] * dpx_b[1] + cam.bg[2] * dpx_b[2];
  const float ddx = 0.5f * cam.W, ddy = 0.5f * cam.H;
  for (int rd = 0; rd < rounds; ++rd, todo -= BLOCK) {
    __syncthreads();
    const int prog = rd * BLOCK + tid;
    if (r0 + prog < r1) {
      const int g = plist[r1 - prog - 1];  // back to front
      s_id[tid] = g;
      s_xy[tid] = make_float2(xy[g * 2], xy[g * 2 + 1]);
      s_co[tid] = *reinterpret_cast<const f32x4*>(conic_o + g * 4);
      s_cd[tid] = f32x4{colors[g * 3], colors[g * 3 + 1], colors[g * 3 + 2], 0.f};
      if (DUAL)
        s_cb[tid] = f32x4{colors_b[g * 3], colors_b[g * 3 + 1],
                          colors_b[g * 3 + 2], 0.f};
    }
    __syncthreads();
    for (int j = 0; j < min(BLOCK, todo); ++j) {
      --contributor;
      float g_col[3] = {0.f, 0.f, 0.f}, g_m[2] = {0.f, 0.f},
            g_con[3] = {0.f, 0.f, 0.f}, g_op = 0.f;
      float g_colb[3] = {0.f, 0.f, 0.f};
      bool active = inside && contributor < last_contributor;
      float dx = 0.f, dy = 0.f, G = 0.f, alpha = 0.f;
      f32x4 co = {0.f, 0.f, 0.f, 0.f};
      if (active) {
        dx = s_xy[j].x - pfx;
        dy = s_xy[j].y - pfy;
        co = s_co[j];
        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.f) {
          active = false;
        } else {
          G = expf(power);
          alpha = fminf(0.99f, co[3] * G);
          if (alpha < 1.f / 255.f) active = false;
        }
      }
      if (active) {
        T = T / (1.f - alpha);
        const float dch = alpha * T;
        const f32x4 cd = s_cd[j];
        float dL_dalpha = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          accum[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum[ch];
          last_color[ch] = cd[ch];
          dL_dalpha += (cd[ch] - accum[ch]) * dpx[ch];
          g_col[ch] = dch * dpx[ch];
        }
        if (DUAL) {
          const f32x4 cb = s_cb[j];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            accum_b[ch] = last_alpha * last_color_b[ch] +
                          (1.f - last_alpha) * accum_b[ch];
            last_color_b[ch] = cb[ch];
            dL_dalpha += (cb[ch] - accum_b[ch]) * dpx_b[ch];
            g_colb[ch] = dch * dpx_b[ch];
          }
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
        const float dL_dG = co[3] * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddelx = -gdx * co[0] - gdy * co[1];
        const float dG_ddely = -gdy * co[2] - gdx * co[1];
        g_m[0] = dL_dG * dG_ddelx * ddx;
        g_m[1] = dL_dG * dG_ddely * ddy;
        g_con[0] = -0.5f * gdx * dx * dL_dG;
        g_con[1] = -gdx * dy * dL_dG;  // true d/d(conic.y): power has -B dx dy
        g_con[2] = -0.5f * gdy * dy * dL_dG;
        g_op = G * dL_dalpha;
      }
      if (__ballot(active) != 0ull) {
        // reduce over the 64 pixels of the wave, then one atomic per value
        float v[9] = {g_col[0], g_col[1], g_col[2], g_m[0], g_m[1],
                      g_con[0], g_con[1], g_con[2], g_op};
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = wave_sum_dpp(v[k]);
        if (DUAL) {
          float u[3] = {g_colb[0], g_colb[1], g_colb[2]};
#pragma unroll
          for (int k = 0; k < 3; ++k) u[k] = wave_sum_dpp(u[k]);
          if (lane == 0) {
            const int g = s_id[j];
            atomicAdd(dL_dcolors_b + g * 3 + 0, u[0]);
            atomicAdd(dL_dcolors_b + g * 3 + 1, u[1]);
            atomicAdd(dL_dcolors_b + g * 3 + 2, u[2]);
          }
        }
        if (lane == 0) {
          const int g = s_id[j];
          atomicAdd(dL_dcolors + g * 3 + 0, v[0]);
          atomicAdd(dL_dcolors + g * 3 + 1, v[1]);
          atomicAdd(dL_dcolors + g * 3 + 2, v[2]);
          atomicAdd(dL_dmean2D + g * 2 + 0, v[3]);
          atomicAdd(dL_dmean2D + g * 2 + 1, v[4]);
          atomicAdd(dL_dconic + g * 3 + 0, v[5]);
          atomicAdd(dL_dconic + g * 3 + 1, v[6]);
          atomicAdd(dL_dconic + g * 3 + 2, v[7]);
          atomicAdd(dL_dopac + g, v[8]);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void gs_preprocess_bwd_kernel(
    Cam cam, int n, const float* __restrict__ means, const float* __restrict__ scales,
    const float* __restrict__ rots, const int* __restrict__ radii,
    const float* __restrict__ dL_dmean2D, const float* __restrict__ dL_dconic,
    float* __restrict__ dL_dmeans, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gm[3] = {0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
  if (radii[i] > 0) {
    Proj P;
    const float* p = means + i * 3;
    project(cam, p, scales + i * 3, rots + i * 4, P);
    const float* V = cam.view;
    const float* M = cam.proj;
    // ---- conic -> 2-D covariance
    const float a = P.a, b = P.b, c = P.c;
    const float det = a * c - b * b;
    const float gA = dL_dconic[i * 3], gB = dL_dconic[i * 3 + 1], gC = dL_dconic[i * 3 + 2];
    const float d2 = 1.f / (det * det + 1e-7f);
    const float g_a = d2 * (-c * c * gA + b * c * gB - b * b * gC);
    const float g_b = d2 * (2.f * b * c * gA - (det + 2.f * b * b) * gB + 2.f * a * b * gC);
    const float g_c = d2 * (-b * b * gA + a * b * gB - a * a * gC);
    const float G2[2][2] = {{g_a, 0.5f * g_b}, {0.5f * g_b, g_c}};
    // dL/dSigma = T^T G2 T ; dL/dT = 2 G2 T Sigma
    float GT[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) GT[r][j] = G2[r][0] * P.T[0][j] + G2[r][1] * P.T[1][j];
    float Ms[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) Ms[r][j] = P.T[0][r] * GT[0][j] + P.T[1][r] * GT[1][j];
    float dT[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        dT[r][j] = 2.f * (GT[r][0] * P.Sig[0][j] + GT[r][1] * P.Sig[1][j] + GT[r][2] * P.Sig[2][j]);
    // dL/dJ = dT W^T ; W[i][j] = V[j*4+i]
    float dJ[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        dJ[r][k] = dT[r][0] * V[0 * 4 + k] + dT[r][1] * V[1 * 4 + k] + dT[r][2] * V[2 * 4 + k];
    const float tz = P.tz, tz2 = 1.f / (tz * tz), tz3 = tz2 / tz;
    const float g_tx = P.xmul * (-P.fx * tz2) * dJ[0][2];
    const float g_ty = P.ymul * (-P.fy * tz2) * dJ[1][2];
    const float g_tz = -P.fx * tz2 * dJ[0][0] - P.fy * tz2 * dJ[1][1] +
                       2.f * P.fx * P.tx * tz3 * dJ[0][2] + 2.f * P.fy * P.ty * tz3 * dJ[1][2];
    // view -> world: p_view = W p + t
#pragma unroll
    for (int k = 0; k < 3; ++k)
      gm[k] = V[k * 4 + 0] * g_tx + V[k * 4 + 1] * g_ty + V[k * 4 + 2] * g_tz;
    // ---- screen position -> world
    const float gx2 = dL_dmean2D[i * 2], gy2 = dL_dmean2D[i * 2 + 1];
    const float mw = P.m_w;
    const float mul1 = P.hom[0] * mw * mw, mul2 = P.hom[1] * mw * mw;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      gm[k] += (M[k * 4 + 0] * mw - M[k * 4 + 3] * mul1) * gx2 +
               (M[k * 4 + 1] * mw - M[k * 4 + 3] * mul2) * gy2;
    // ---- Sigma = R diag(S2) R^T
    float MR[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        MR[r][k] = Ms[r][0] * P.R[0][k] + Ms[r][1] * P.R[1][k] + Ms[r][2] * P.R[2][k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float rmr = P.R[0][k] * MR[0][k] + P.R[1][k] * MR[1][k] + P.R[2][k] * MR[2][k];
      gs[k] = 2.f * scales[i * 3 + k] * cam.scale_modifier * cam.scale_modifier * rmr;
    }
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) g[r][k] = 2.f * MR[r][k] * P.S2[k];
    const float* q = rots + i * 4;
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    gq[0] = 2.f * (-z * g[0][1] + y * g[0][2] + z * g[1][0] - x * g[1][2] - y * g[2][0] + x * g[2][1]);
    gq[1] = 2.f * (y * g[0][1] + z * g[0][2] + y * g[1][0] - 2.f * x * g[1][1] - r * g[1][2] +
                   z * g[2][0] + r * g[2][1] - 2.f * x * g[2][2]);
    gq[2] = 2.f * (-2.f * y * g[0][0] + x * g[0][1] + r * g[0][2] + x * g[1][0] + z * g[1][2] -
                   r * g[2][0] + z * g[2][1] - 2.f * y * g[2][2]);
    gq[3] = 2.f * (-2.f * z * g[0][0] - r * g[0][1] + x * g[0][2] + r * g[1][0] - 2.f * z * g[1][1] +
                   y * g[1][2] + x * g[2][0] + y * g[2][1]);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dL_dmeans[i * 3 + k] = gm[k];
    dL_dscales[i * 3 + k] = gs[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) dL_drots[i * 4 + k] = gq[k];
}

int to_cam(const xrd_gs_camera* c, Cam& cam) {
  if (!c || c->image_height < 1 || c->image_width < 1) return XRD_ERR_ARG;
  cam.H = c->image_height;
  cam.W = c->image_width;
  cam.tanfovx = c->tanfovx;
  cam.tanfovy = c->tanfovy;
  for (int i = 0; i < 3; ++i) cam.bg[i] = c->bg[i];
  cam.scale_modifier = c->scale_modifier;
  for (int i = 0; i < 16; ++i) {
    cam.view[i] = c->viewmatrix[i];
    cam.proj[i] = c->projmatrix[i];
  }
  return XRD_OK;
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int xrd_gs_preprocess(const xrd_gs_camera* c, int n, const float* means3D,
                      const float* scales, const float* rotations,
                      const float* opacities, float* depths, float* xy,
                      float* conic_opacity, int32_t* radii, int32_t* rect,
                      int32_t* tiles_touched, xrd_stream_t stream) {
  Cam cam;
  int rc = to_cam(c, cam);
  if (rc) return rc;
  if (n < 0) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!means3D || !scales || !rotations || !opacities || !depths || !xy ||
      !conic_opacity || !radii || !rect || !tiles_touched)
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(gs_preprocess_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, cam, n, means3D, scales, rotations,
                     opacities, depths, xy, conic_opacity, radii, rect,
                     tiles_touched);
  return check_launch("xrd_gs_preprocess");
}

int xrd_gs_band_clip(int n, int tile_row0, int tile_row1, int32_t* rect,
                     int32_t* tiles_touched, xrd_stream_t stream) {
  if (n < 0 || tile_row0 < 0 || tile_row1 < tile_row0) return XRD_ERR_ARG;
  if (!rect || !tiles_touched) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  hipLaunchKernelGGL(gs_band_clip_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, n, tile_row0, tile_row1, rect,
                     tiles_touched);
  return check_launch("xrd_gs_band_clip");
}

int xrd_gs_duplicate_keys(int n, int image_width, const int32_t* rect,
                          const int64_t* offsets_inclusive,
                          const float* depths, int64_t* keys, int32_t* values,
                          xrd_stream_t stream) {
  if (n < 0 || image_width < 1) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!rect || !offsets_inclusive || !depths || !keys || !values) return XRD_ERR_ARG;
  hipLaunchKernelGGL(gs_duplicate_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, n, rect, offsets_inclusive, depths,
                     (image_width + TILE - 1) / TILE, keys, values);
  return check_launch("xrd_gs_duplicate_keys");
}

int xrd_gs_tile_ranges(int64_t n_keys, const int64_t* sorted_keys,
                       int32_t* ranges, xrd_stream_t stream) {
  if (n_keys < 0) return XRD_ERR_ARG;
  if (n_keys == 0) return XRD_OK;
  if (!sorted_keys || !ranges) return XRD_ERR_ARG;
  hipLaunchKernelGGL(gs_ranges_kernel, dim3((unsigned)((n_keys + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, n_keys, sorted_keys,
                     ranges);
  return check_launch("xrd_gs_tile_ranges");
}

int xrd_gs_render_fwd(const xrd_gs_camera* c, const int32_t* ranges,
                      const int32_t* point_list, const float* xy,
                      const float* colors, const float* conic_opacity,
                      const float* depths, float* out_color, float* out_depth,
                      float* final_T, int32_t* n_contrib, xrd_stream_t stream) {
  Cam cam;
  int rc = to_cam(c, cam);
  if (rc) return rc;
  if (!ranges || !out_color || !out_depth || !final_T || !n_contrib) return XRD_ERR_ARG;
  const dim3 grid((cam.W + TILE - 1) / TILE, (cam.H + TILE - 1) / TILE);
  hipLaunchKernelGGL(gs_render_fwd_kernel<false>, grid, dim3(TILE, TILE), 0,
                     (hipStream_t)stream, cam, ranges, point_list, xy, colors,
                     nullptr, conic_opacity, depths, out_color, nullptr,
                     out_depth, final_T, n_contrib);
  return check_launch("xrd_gs_render_fwd");
}

int xrd_gs_render_fwd2(const xrd_gs_camera* c, const int32_t* ranges,
                       const int32_t* point_list, const float* xy,
                       const float* colors_a, const float* colors_b,
                       const float* conic_opacity, const float* depths,
                       float* out_color_a, float* out_color_b,
                       float* out_depth, float* final_T, int32_t* n_contrib,
                       xrd_stream_t stream) {
  Cam cam;
  int rc = to_cam(c, cam);
  if (rc) return rc;
  if (!ranges || !out_color_a || !out_color_b || !out_depth || !final_T ||
      !n_contrib)
    return XRD_ERR_ARG;
  const dim3 grid((cam.W + TILE - 1) / TILE, (cam.H + TILE - 1) / TILE);
  hipLaunchKernelGGL(gs_render_fwd_kernel<true>, grid, dim3(TILE, TILE), 0,
                     (hipStream_t)stream, cam, ranges, point_list, xy,
                     colors_a, colors_b, conic_opacity, depths, out_color_a,
                     out_color_b, out_depth, final_T, n_contrib);
  return check_launch("xrd_gs_render_fwd2");
}

int xrd_gs_render_bwd(const xrd_gs_camera* c, const int32_t* ranges,
                      const int32_t* point_list, const float* xy,
                      const float* conic_opacity, const float* colors,
                      const float* final_T, const int32_t* n_contrib,
                      const float* dL_dcolor, float* dL_dmean2D,
                      float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                      xrd_stream_t stream) {
  Cam cam;
  int rc = to_cam(c, cam);
  if (rc) return rc;
  if (!ranges || !final_T || !n_contrib || !dL_dcolor || !dL_dmean2D ||
      !dL_dconic || !dL_dopacity || !dL_dcolors)
    return XRD_ERR_ARG;
  const dim3 grid((cam.W + TILE - 1) / TILE, (cam.H + TILE - 1) / TILE);
  hipLaunchKernelGGL(gs_render_bwd_kernel<false>, grid, dim3(TILE, TILE), 0,
                     (hipStream_t)stream, cam, ranges, point_list, xy,
                     conic_opacity, colors, nullptr, final_T, n_contrib,
                     dL_dcolor, nullptr, dL_dmean2D, dL_dconic, dL_dopacity,
                     dL_dcolors, nullptr);
  return check_launch("xrd_gs_render_bwd");
}

int xrd_gs_render_bwd2(const xrd_gs_camera* c, const int32_t* ranges,
                       const int32_t* point_list, const float* xy,
                       const float* conic_opacity, const float* colors_a,
                       const float* colors_b, const float* final_T,
                       const int32_t* n_contrib, const float* dL_dcolor_a,
                       const float* dL_dcolor_b, float* dL_dmean2D,
                       float* dL_dconic, float* dL_dopacity,
                       float* dL_dcolors_a, float* dL_dcolors_b,
                       xrd_stream_t stream) {
  Cam cam;
  int rc = to_cam(c, cam);
  if (rc) return rc;
  if (!ranges || !final_T || !n_contrib || !dL_dcolor_a || !dL_dcolor_b ||
      !dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dcolors_a ||
      !dL_dcolors_b)
    return XRD_ERR_ARG;
  const dim3 grid((cam.W + TILE - 1) / TILE, (cam.H + TILE - 1) / TILE);
  hipLaunchKernelGGL(gs_render_bwd_kernel<true>, grid, dim3(TILE, TILE), 0,
                     (hipStream_t)stream, cam, ranges, point_list, xy,
                     conic_opacity, colors_a, colors_b, final_T, n_contrib,
                     dL_dcolor_a, dL_dcolor_b, dL_dmean2D, dL_dconic,
                     dL_dopacity, dL_dcolors_a, dL_dcolors_b);
  return check_launch("xrd_gs_render_bwd2");
}

int xrd_gs_preprocess_bwd(const xrd_gs_camera* c, int n, const float* means3D,
                          const float* scales, const float* rotations,
                          const int32_t* radii, const float* dL_dmean2D,
                          const float* dL_dconic, float* dL_dmeans3D,
                          float* dL_dscales, float* dL_drotations,
                          xrd_stream_t stream) {
  Cam cam;
  int rc = to_cam(c, cam);
  if (rc) return rc;
  if (n < 0) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!means3D || !scales || !rotations || !radii || !dL_dmean2D ||
      !dL_dconic || !dL_dmeans3D || !dL_dscales || !dL_drotations)
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(gs_preprocess_bwd_kernel, dim3((n + 255) / 256), dim3(256),
                     0, (hipStream_t)stream, cam, n, means3D, scales, rotations,
                     radii, dL_dmean2D, dL_dconic, dL_dmeans3D, dL_dscales,
                     dL_drotations);
  return check_launch("xrd_gs_preprocess_bwd");
}

}  // extern "C"
