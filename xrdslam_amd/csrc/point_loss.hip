// Point-SLAM mapping: compositing + loss + their backward as one launch
// (raw2outputs_nerf_color2, slam/model_components/utils.py:247-294, and the
// mapping branch of get_loss_dict, slam/models/conv_onet_pointslam.py:
// 190-204).  One thread per ray, S <= 16 samples in registers:
//   alpha_s = sigmoid(coef occ_s)   (occ_s = -100 where the sample has no
//                                    neighbours, conv_onet_pointslam.py:441)
//   w_s = alpha_s prod_{j<s} (1 - alpha_j + 1e-10),  W = sum w + 1e-10
//   depth = sum w z / W,  colour = sum w rgb / W
//   a ray counts when target_d > 0, >= min_valid of its samples have
//   neighbours, its depth is not NaN (and the batch mask, if given, keeps it)
//   loss = sum |target_d - depth| + w_color sum |target_rgb - colour|
// The loss is a plain sum with unit weight, so the gradient w.r.t. raw
// ([rgb, occ] per sample) is final when the kernel returns; the autograd
// wrapper scales it by the incoming gradient.  Restated from the reference's
// formulas, not copied; parity: tests/test_pointslam_hip.py.
#include "common.h"

namespace xrd {
namespace {

constexpr int kMaxS = 16;

__global__ __launch_bounds__(256) void point_map_loss_kernel(
    int n, int S, const float* __restrict__ raw,
    const uint8_t* __restrict__ point_mask, const float* __restrict__ z_vals,
    const float* __restrict__ target_d, const float* __restrict__ target_rgb,
    const uint8_t* __restrict__ ray_valid, float coef, float w_color,
    int min_valid, float* __restrict__ loss, float* __restrict__ g_raw) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  // The per-ray arithmetic runs in f64 (S <= 16 samples a ray: free): behind
  // a saturated sample 1 - alpha is a difference of nearly equal numbers and
  // its f32 rounding (relative ~1e-3) lands in every later weight; the f64
  // evaluation of the reference is the value two f32 evaluations scatter
  // around (tests/test_pointslam_hip.py judges against it).
  float l_geo = 0.f, l_rgb = 0.f;
  if (ray < n) {
    double alpha[kMaxS], T[kMaxS], w[kMaxS], z[kMaxS];
    int cnt = 0;
    double run = 1.0, wsum = 0.0;
    for (int s = 0; s < S; ++s) {
      const bool has = point_mask[ray * S + s] != 0;
      cnt += has;
      const double occ = has ? (double)raw[(ray * S + s) * 4 + 3] : -100.0;
      alpha[s] = 1.0 / (1.0 + exp(-((double)coef * occ)));
      T[s] = run;
      w[s] = alpha[s] * run;
      run = run * (1.0 - alpha[s] + 1e-10);
      wsum += w[s];
      z[s] = (double)z_vals[ray * S + s];
    }
    const double W = wsum + 1e-10;
    double A = 0.0;
    for (int s = 0; s < S; ++s) A += w[s] * z[s];
    const double depth = A / W;
    const double td = (double)target_d[ray];
    bool m = td > 0.0 && cnt >= min_valid && !(depth != depth);
    if (ray_valid != nullptr) m = m && ray_valid[ray] != 0;
    double col[3] = {0.0, 0.0, 0.0}, g_col[3] = {0.0, 0.0, 0.0};
    const bool color = target_rgb != nullptr;
    if (color) {
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          col[c] += w[s] * (double)raw[(ray * S + s) * 4 + c];
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] /= W;
    }
    double g_depth = 0.0;
    if (m) {
      const double e = td - depth;
      l_geo = (float)fabs(e);
      g_depth = e > 0.0 ? -1.0 : (e < 0.0 ? 1.0 : 0.0);
      if (color) {
        double lr = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double ec = (double)target_rgb[ray * 3 + c] - col[c];
          lr += fabs(ec);
          g_col[c] = (double)w_color * (ec > 0.0 ? -1.0 : (ec < 0.0 ? 1.0 : 0.0));
        }
        l_rgb = (float)(lr * (double)w_color);
      }
    }
    // backward: g_w_s, then the transmittance chain from the last sample
    double tail = 0.0;   // sum_{k > s} g_w_k w_k
    for (int s = S - 1; s >= 0; --s) {
      double g_w = g_depth * (z[s] - depth) / W;
      f32x4 out = {0.f, 0.f, 0.f, 0.f};
      if (color) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          g_w += g_col[c] * ((double)raw[(ray * S + s) * 4 + c] - col[c]) / W;
          out[c] = (float)(g_col[c] * w[s] / W);
        }
      }
      const double g_alpha = g_w * T[s] - tail / (1.0 - alpha[s] + 1e-10);
      tail += g_w * w[s];
      out[3] = (float)(g_alpha * (double)coef * alpha[s] * (1.0 - alpha[s]));
      *reinterpret_cast<f32x4*>(g_raw + (ray * S + s) * 4) = out;
    }
  }
  // block sums -> two atomics per block
  __shared__ float red[2][4];
  const float sg = wave_sum(l_geo), sr = wave_sum(l_rgb);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    red[0][wave] = sg;
    red[1][wave] = sr;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(loss, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    atomicAdd(loss + 1, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" int xrd_point_map_loss(int n_rays, int n_samples, const float* raw,
                                  const uint8_t* point_mask,
                                  const float* z_vals, const float* target_d,
                                  const float* target_rgb,
                                  const uint8_t* ray_valid, float sigmoid_coef,
                                  float w_color, int min_valid_points,
                                  float* loss, float* g_raw,
                                  xrd_stream_t stream) {
  if (n_rays < 0 || n_samples < 1 || n_samples > kMaxS) return XRD_ERR_ARG;
  if (!loss) return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int rc = zero_floats(loss, 2, st);
  if (rc != XRD_OK) return rc;
  if (n_rays == 0) return XRD_OK;
  if (!raw || !point_mask || !z_vals || !target_d || !g_raw)
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(point_map_loss_kernel, dim3((n_rays + 255) / 256),
                     dim3(256), 0, st, n_rays, n_samples, raw, point_mask,
                     z_vals, target_d, target_rgb, ray_valid, sigmoid_coef,
                     w_color, min_valid_points, loss, g_raw);
  return check_launch("xrd_point_map_loss");
}

// ---- batch selection + sample points -----------------------------------------------
// PointSLAM.get_model_input after the per-frame sampling
// (slam/algorithms/point_slam.py:246-300) + the sample placement of
// render_batch_ray (slam/models/conv_onet_pointslam.py:330-347) as ONE launch:
//   valid = d > 0; med = lower median of d[valid]; top = max d[valid];
//   keep  = valid & (d <= min(10 med, 1.2 top))       (the batch filter)
//   z     = near d (1 - t) + far d t, t = linspace(0, 1, S)
//   pts   = o + dir z;  per-point query radius = the pixel's radius
// The median is a 4-pass radix select over the depth bits in one block (the
// torch formulation sorts the batch); arithmetic as separate roundings like
// the torch ops it replaces.
namespace xrd {
namespace {

constexpr int BATCH_THREADS = 1024;
constexpr int BATCH_RAYS = 256;   // rays placed by one block of the batch kernel
constexpr int SEL_REGS = 16;      // values a thread caches for the select

// k-th smallest (k = s_k) of the non-negative floats whose bit patterns the
// threads hold in key[0..cnt) (each thread its own), 8 bits a pass over an LDS
// histogram; every thread of the block returns the bits of the result.
// `hist` [256] and the two words are LDS scratch; s_k holds k on entry.
__device__ __forceinline__ unsigned radix_select_block(
    const unsigned (&key)[SEL_REGS], int cnt, int* hist, unsigned* s_prefix,
    unsigned* s_k) {
  const int tid = threadIdx.x;
  if (tid == 0) *s_prefix = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = *s_prefix;
    const unsigned himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
#pragma unroll
    for (int r = 0; r < SEL_REGS; ++r)
      if (r < cnt && (key[r] & himask) == prefix)
        atomicAdd(&hist[(key[r] >> shift) & 255], 1);
    __syncthreads();
    if (tid < 64) {
      // wave 0: lane l owns bins 4l..4l+3; exclusive prefix over the lanes
      const int h0 = hist[tid * 4], h1 = hist[tid * 4 + 1],
                h2 = hist[tid * 4 + 2], h3 = hist[tid * 4 + 3];
      const int mine = h0 + h1 + h2 + h3;
      int incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (tid >= d) incl += o;
      }
      const unsigned k = *s_k;
      const unsigned excl = (unsigned)(incl - mine);
      if (excl <= k && k < (unsigned)incl) {   // exactly one lane
        unsigned acc = excl;
        int bin = tid * 4;
        if (acc + (unsigned)h0 <= k) { acc += h0; ++bin;
          if (acc + (unsigned)h1 <= k) { acc += h1; ++bin;
            if (acc + (unsigned)h2 <= k) { acc += h2; ++bin; } } }
        *s_k = k - acc;
        *s_prefix = prefix | ((unsigned)bin << shift);
      }
    }
    __syncthreads();
  }
  return *s_prefix;
}

__global__ __launch_bounds__(BATCH_THREADS) void point_batch_kernel(
    int n, int S, const float* __restrict__ ro, const float* __restrict__ rd,
    const float* __restrict__ d, const float* __restrict__ radius_stack,
    const int64_t* __restrict__ idx, int n_per, int wcrop, int hedge,
    int wedge, int width, int64_t hw, float near_c, float far_c,
    uint8_t* __restrict__ keep, float* __restrict__ radius,
    float* __restrict__ z_vals, float* __restrict__ pts,
    float* __restrict__ radius_pts, float* __restrict__ stats) {
  // separate roundings like the torch ops this replaces (plain operators:
  // the __f*_rn intrinsics inline WITH the library's contract flag and are
  // fused again)
#pragma clang fp contract(off)
  __shared__ int hist[256];
  __shared__ unsigned s_prefix, s_k, s_cnt, s_top;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_cnt = 0;
    s_top = 0;
  }
  __syncthreads();
  // every block selects the median of the whole batch (n <= 16 K values, a
  // few KB from L2) and then places its own 256 rays
  unsigned key[SEL_REGS];
  int cnt = 0;
  unsigned top = 0;
  for (int i = tid; i < n; i += BATCH_THREADS) {
    const float v = d[i];
    if (v > 0.f) {
      const unsigned b = __float_as_uint(v);   // positive floats order as bits
      top = max(top, b);
#pragma unroll
      for (int r = 0; r < SEL_REGS; ++r)
        if (r == cnt) key[r] = b;
      ++cnt;
    }
  }
  if (cnt != 0) {
    atomicAdd(&s_cnt, (unsigned)cnt);
    atomicMax(&s_top, top);
  }
  __syncthreads();
  const unsigned total = s_cnt;
  if (tid == 0) s_k = total > 0 ? (total - 1) / 2 : 0;
  __syncthreads();
  const unsigned mbits = radix_select_block(key, cnt, hist, &s_prefix, &s_k);
  const float med = total > 0 ? __uint_as_float(mbits) : NAN;
  const float topf = total > 0 ? __uint_as_float(s_top) : -INFINITY;
  // torch.minimum propagates NaN; every comparison with it is false
  const float lim = fminf(10.f * med, 1.2f * topf);
  if (blockIdx.x == 0 && tid == 0 && stats != nullptr) {
    stats[0] = med;
    stats[1] = topf;
  }
  const float step = S > 1 ? 1.f / (float)(S - 1) : 0.f;
  const int i = blockIdx.x * BATCH_RAYS + (tid >> 2);
  if (i >= n) return;
  const float v = d[i];
  float r = 0.f;
  if (radius_stack != nullptr) {
    const int64_t e = idx[i];
    const int64_t row = hedge + e / wcrop, col = wedge + e % wcrop;
    r = radius_stack[(int64_t)(i / n_per) * hw + row * width + col];
  }
  if ((tid & 3) == 0) {
    keep[i] = (total > 0 && v > 0.f && v <= lim) ? 1 : 0;
    if (radius_stack != nullptr) radius[i] = r;
  }
  const float a = near_c * v, b = far_c * v;
  const float o0 = ro[i * 3], o1 = ro[i * 3 + 1], o2 = ro[i * 3 + 2];
  const float d0 = rd[i * 3], d1 = rd[i * 3 + 1], d2 = rd[i * 3 + 2];
  for (int s = tid & 3; s < S; s += 4) {
    // torch.linspace: from the start in the first half, from the end in
    // the second
    float t;
    if (s < S / 2) {
      t = step * (float)s;
    } else {
      const float back = step * (float)(S - 1 - s);
      t = 1.f - back;
    }
    const float u = 1.f - t;
    const float za = a * u, zb = b * t;
    const float z = za + zb;
    const int64_t m = (int64_t)i * S + s;
    z_vals[m] = z;
    const float p0 = d0 * z, p1 = d1 * z, p2 = d2 * z;
    pts[m * 3] = o0 + p0;
    pts[m * 3 + 1] = o1 + p1;
    pts[m * 3 + 2] = o2 + p2;
    if (radius_stack != nullptr) radius_pts[m] = r;
  }
}

}  // namespace
}  // namespace xrd

extern "C" int xrd_point_batch(
    int n_rays, int n_samples, const float* rays_o, const float* rays_d,
    const float* target_d, const float* radius_stack, const int64_t* pixel_idx,
    int rays_per_frame, int crop_width, int hedge, int wedge, int image_width,
    int64_t image_pixels, float near_coef, float far_coef, uint8_t* keep,
    float* radius, float* z_vals, float* pts, float* radius_pts, float* stats,
    xrd_stream_t stream) {
  if (n_rays < 1 || n_samples < 1 || n_samples > 64) return XRD_ERR_ARG;
  if (n_rays > xrd::BATCH_THREADS * xrd::SEL_REGS) return XRD_ERR_UNSUPPORTED;
  if (!rays_o || !rays_d || !target_d || !keep || !z_vals || !pts)
    return XRD_ERR_ARG;
  if (radius_stack != nullptr &&
      (!pixel_idx || !radius || !radius_pts || rays_per_frame < 1 ||
       crop_width < 1 || image_width < 1 || image_pixels < 1))
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(xrd::point_batch_kernel,
                     dim3((n_rays + xrd::BATCH_RAYS - 1) / xrd::BATCH_RAYS),
                     dim3(xrd::BATCH_THREADS), 0, (hipStream_t)stream, n_rays,
                     n_samples, rays_o, rays_d, target_d, radius_stack,
                     pixel_idx, rays_per_frame, crop_width, hedge, wedge,
                     image_width, image_pixels, near_coef, far_coef, keep,
                     radius, z_vals, pts, radius_pts, stats);
  return xrd::check_launch("xrd_point_batch");
}

// ---- tracking loss ------------------------------------------------------------------
// ConvOnet2.get_loss_dict, tracking branch (slam/models/conv_onet_pointslam.py:
// 144-189) on a batch that kept its shape (the batch filter arrives as
// ray_valid): tmp = |d - depth| / sqrt(var + 1e-10) (handle_dynamic; else
// |d - depth|), med = LOWER median of tmp over the rays of the batch (NaN if
// one of them is NaN), mask = tmp < 10 med & d > 0 & depth, var not NaN &
// ray_valid;  geo = sum_mask clamp(|d - depth| / sqrt(var + 1e-10), 0, 1e3),
// rgb = w_color sum_mask |rgb - colour|_1.  One block: radix-select median,
// sums, and the gradients w.r.t. depth and colour (var is detached).
namespace xrd {
namespace {

__global__ __launch_bounds__(BATCH_THREADS) void point_track_loss_kernel(
    int n, int handle_dynamic, int use_color, float w_color,
    const float* __restrict__ depth, const float* __restrict__ var,
    const float* __restrict__ color, const float* __restrict__ tgt_d,
    const float* __restrict__ tgt_rgb, const uint8_t* __restrict__ ray_valid,
    float* __restrict__ loss, float* __restrict__ g_depth,
    float* __restrict__ g_color) {
#pragma clang fp contract(off)
  __shared__ int hist[256];
  __shared__ unsigned s_prefix, s_k, s_cnt, s_nan;
  __shared__ double s_sum[2];
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_cnt = 0;
    s_nan = 0;
    s_sum[0] = s_sum[1] = 0.0;
  }
  __syncthreads();
  auto tmp_of = [&](int i) {
    const float e = fabsf(tgt_d[i] - depth[i]);
    return handle_dynamic ? e / sqrtf(var[i] + 1e-10f) : e;
  };
  unsigned key[SEL_REGS];
  int cnt = 0;
  unsigned any_nan = 0;
  for (int i = tid; i < n; i += BATCH_THREADS) {
    if (ray_valid != nullptr && !ray_valid[i]) continue;
    const float t = tmp_of(i);
    if (t != t) any_nan = 1;
    // tmp >= 0 (or +inf): its bits order like the values; -0 -> +0
    const unsigned b = __float_as_uint(t) & 0x7fffffffu;
#pragma unroll
    for (int r = 0; r < SEL_REGS; ++r)
      if (r == cnt) key[r] = b;
    ++cnt;
  }
  if (cnt) atomicAdd(&s_cnt, (unsigned)cnt);
  if (any_nan) atomicOr(&s_nan, 1u);
  __syncthreads();
  const unsigned total = s_cnt;
  const bool poisoned = s_nan != 0 || total == 0;
  if (tid == 0) s_k = total > 0 ? (total - 1) / 2 : 0;
  __syncthreads();
  unsigned mbits = 0;
  if (!poisoned)   // block-uniform
    mbits = radix_select_block(key, cnt, hist, &s_prefix, &s_k);
  const float med = poisoned ? NAN : __uint_as_float(mbits);
  const float lim = 10.f * med;
  double geo = 0.0, rgb = 0.0;
  for (int i = tid; i < n; i += BATCH_THREADS) {
    const float d = tgt_d[i], dep = depth[i], v = var[i];
    const float t = tmp_of(i);
    const bool m = t < lim && d > 0.f && dep == dep && v == v &&
                   (ray_valid == nullptr || ray_valid[i] != 0);
    float gd = 0.f;
    if (m) {
      const float s = sqrtf(v + 1e-10f);
      const float q = fabsf(d - dep) / s;
      geo += (double)fminf(fmaxf(q, 0.f), 1e3f);
      // clamp passes the gradient inside [0, 1e3]; |x|' = sign(x)
      if (q >= 0.f && q <= 1e3f && d != dep)
        gd = (d - dep > 0.f ? -1.f : 1.f) / s;
    }
    g_depth[i] = gd;
    if (use_color) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float e = tgt_rgb[i * 3 + c] - color[i * 3 + c];
        if (m) rgb += (double)fabsf(e);
        g_color[i * 3 + c] =
            m ? (e > 0.f ? -w_color : (e < 0.f ? w_color : 0.f)) : 0.f;
      }
    }
  }
  geo = wave_sum(geo);
  rgb = wave_sum(rgb);
  if ((tid & 63) == 0) {
    atomicAdd(&s_sum[0], geo);
    atomicAdd(&s_sum[1], rgb);
  }
  __syncthreads();
  if (tid == 0) {
    loss[0] = (float)s_sum[0];
    loss[1] = use_color ? w_color * (float)s_sum[1] : 0.f;
  }
}

}  // namespace
}  // namespace xrd

extern "C" int xrd_point_track_loss(
    int n_rays, int handle_dynamic, int use_color, float w_color,
    const float* depth, const float* var, const float* color,
    const float* target_d, const float* target_rgb, const uint8_t* ray_valid,
    float* loss, float* g_depth, float* g_color, xrd_stream_t stream) {
  if (n_rays < 1) return XRD_ERR_ARG;
  if (n_rays > xrd::BATCH_THREADS * xrd::SEL_REGS) return XRD_ERR_UNSUPPORTED;
  if (!depth || !var || !target_d || !loss || !g_depth) return XRD_ERR_ARG;
  if (use_color && (!color || !target_rgb || !g_color)) return XRD_ERR_ARG;
  hipLaunchKernelGGL(xrd::point_track_loss_kernel, dim3(1),
                     dim3(xrd::BATCH_THREADS), 0, (hipStream_t)stream, n_rays,
                     handle_dynamic, use_color, w_color, depth, var, color,
                     target_d, target_rgb, ray_valid, loss, g_depth, g_color);
  return xrd::check_launch("xrd_point_track_loss");
}

// ---- compositing alone (tracking, render_img, and the generic backward) ------------
// raw2outputs_nerf_color2 (slam/model_components/utils.py:247-294) with the
// no-neighbour override of render_batch_ray (conv_onet_pointslam.py:441):
//   depth = sum w z / W, colour = sum w rgb / W, var = sum w (z - depth)^2.
// rgb / occ are addressed with a stride so that both the interleaved
// [rgb, occ] rows of the plugin path and separate kernel outputs fit.
namespace xrd {
namespace {

// f64 per-ray arithmetic, see point_map_loss_kernel
struct RayW {
  double alpha[kMaxS], T[kMaxS], w[kMaxS], z[kMaxS];
  double W, depth;
};

__device__ __forceinline__ void ray_weights(int ray, int S,
                                            const float* __restrict__ occ,
                                            int occ_stride,
                                            const uint8_t* __restrict__ pm,
                                            const float* __restrict__ z_vals,
                                            float coef, RayW& r) {
  double run = 1.0, wsum = 0.0, A = 0.0;
  for (int s = 0; s < S; ++s) {
    const bool has = pm == nullptr || pm[ray * S + s] != 0;
    const double o =
        has ? (double)occ[(int64_t)(ray * S + s) * occ_stride] : -100.0;
    r.alpha[s] = 1.0 / (1.0 + exp(-((double)coef * o)));
    r.T[s] = run;
    r.w[s] = r.alpha[s] * run;
    run = run * (1.0 - r.alpha[s] + 1e-10);
    wsum += r.w[s];
    r.z[s] = (double)z_vals[ray * S + s];
  }
  r.W = wsum + 1e-10;
  for (int s = 0; s < S; ++s) A += r.w[s] * r.z[s];
  r.depth = A / r.W;
}

__global__ __launch_bounds__(256) void point_composite_fwd_kernel(
    int n, int S, const float* __restrict__ rgb, int rgb_stride,
    const float* __restrict__ occ, int occ_stride,
    const uint8_t* __restrict__ pm, const float* __restrict__ z_vals,
    float coef, float* __restrict__ depth, float* __restrict__ var,
    float* __restrict__ color) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n) return;
  RayW r;
  ray_weights(ray, S, occ, occ_stride, pm, z_vals, coef, r);
  depth[ray] = (float)r.depth;
  double v = 0.0, col[3] = {0.0, 0.0, 0.0};
  for (int s = 0; s < S; ++s) {
    const double t = r.z[s] - r.depth;
    v += r.w[s] * t * t;
    if (rgb != nullptr) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        col[c] +=
            r.w[s] * (double)rgb[(int64_t)(ray * S + s) * rgb_stride + c];
    }
  }
  var[ray] = (float)v;
#pragma unroll
  for (int c = 0; c < 3; ++c) color[ray * 3 + c] = (float)(col[c] / r.W);
}

__global__ __launch_bounds__(256) void point_composite_bwd_kernel(
    int n, int S, const float* __restrict__ rgb, int rgb_stride,
    const float* __restrict__ occ, int occ_stride,
    const uint8_t* __restrict__ pm, const float* __restrict__ z_vals,
    float coef, const float* __restrict__ g_depth,
    const float* __restrict__ g_var, const float* __restrict__ g_color,
    float* __restrict__ g_rgb, int g_rgb_stride, float* __restrict__ g_occ,
    int g_occ_stride) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n) return;
  RayW r;
  ray_weights(ray, S, occ, occ_stride, pm, z_vals, coef, r);
  const double gd = g_depth ? (double)g_depth[ray] : 0.0;
  const double gv = g_var ? (double)g_var[ray] : 0.0;
  double gc[3] = {0.0, 0.0, 0.0}, col[3] = {0.0, 0.0, 0.0};
  const bool color = rgb != nullptr && g_color != nullptr;
  double S1 = 0.0;   // sum w (z - depth): d var / d depth = -2 S1
  for (int s = 0; s < S; ++s) {
    S1 += r.w[s] * (r.z[s] - r.depth);
    if (color) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        col[c] +=
            r.w[s] * (double)rgb[(int64_t)(ray * S + s) * rgb_stride + c];
    }
  }
  if (color) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      col[c] /= r.W;
      gc[c] = (double)g_color[ray * 3 + c];
    }
  }
  const double gdep = gd - 2.0 * gv * S1;   // total gradient reaching depth
  double tail = 0.0;
  for (int s = S - 1; s >= 0; --s) {
    const double t = r.z[s] - r.depth;
    double g_w = gdep * t / r.W + gv * t * t;
    if (color) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double v =
            (double)rgb[(int64_t)(ray * S + s) * rgb_stride + c];
        g_w += gc[c] * (v - col[c]) / r.W;
        if (g_rgb != nullptr)
          g_rgb[(int64_t)(ray * S + s) * g_rgb_stride + c] =
              (float)(gc[c] * r.w[s] / r.W);
      }
    } else if (g_rgb != nullptr) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        g_rgb[(int64_t)(ray * S + s) * g_rgb_stride + c] = 0.f;
    }
    const double g_alpha = g_w * r.T[s] - tail / (1.0 - r.alpha[s] + 1e-10);
    tail += g_w * r.w[s];
    g_occ[(int64_t)(ray * S + s) * g_occ_stride] =
        (float)(g_alpha * (double)coef * r.alpha[s] * (1.0 - r.alpha[s]));
  }
}

}  // namespace
}  // namespace xrd

extern "C" {

int xrd_point_composite_fwd(int n_rays, int n_samples, const float* rgb,
                            int rgb_stride, const float* occ, int occ_stride,
                            const uint8_t* point_mask, const float* z_vals,
                            float sigmoid_coef, float* depth, float* var,
                            float* color, xrd_stream_t stream) {
  if (n_rays < 0 || n_samples < 1 || n_samples > kMaxS) return XRD_ERR_ARG;
  if (n_rays == 0) return XRD_OK;
  if (!occ || !z_vals || !depth || !var || !color || occ_stride < 1 ||
      (rgb && rgb_stride < 3))
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(point_composite_fwd_kernel, dim3((n_rays + 255) / 256),
                     dim3(256), 0, (hipStream_t)stream, n_rays, n_samples, rgb,
                     rgb_stride, occ, occ_stride, point_mask, z_vals,
                     sigmoid_coef, depth, var, color);
  return check_launch("xrd_point_composite_fwd");
}

int xrd_point_composite_bwd(int n_rays, int n_samples, const float* rgb,
                            int rgb_stride, const float* occ, int occ_stride,
                            const uint8_t* point_mask, const float* z_vals,
                            float sigmoid_coef, const float* g_depth,
                            const float* g_var, const float* g_color,
                            float* g_rgb, int g_rgb_stride, float* g_occ,
                            int g_occ_stride, xrd_stream_t stream) {
  if (n_rays < 0 || n_samples < 1 || n_samples > kMaxS) return XRD_ERR_ARG;
  if (n_rays == 0) return XRD_OK;
  if (!occ || !z_vals || !g_occ || occ_stride < 1 || g_occ_stride < 1 ||
      (rgb && rgb_stride < 3) || (g_rgb && g_rgb_stride < 3))
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(point_composite_bwd_kernel, dim3((n_rays + 255) / 256),
                     dim3(256), 0, (hipStream_t)stream, n_rays, n_samples, rgb,
                     rgb_stride, occ, occ_stride, point_mask, z_vals,
                     sigmoid_coef, g_depth, g_var, g_color, g_rgb,
                     g_rgb_stride, g_occ, g_occ_stride);
  return check_launch("xrd_point_composite_bwd");
}

}  // extern "C"

// ---- the whole render as one call each way (xrd_point_render_*) ---------
// render_batch_ray given the sample points and their neighbours
// (conv_onet_pointslam.py:302-461): geometry decoder, colour decoder (colour
// stage), compositing.  The per-point buffers belong to the caller and carry
// the forward's results to the backward.
namespace xrd {
namespace {
__global__ __launch_bounds__(256) void add_inplace_kernel(
    int64_t n, float* __restrict__ a, const float* __restrict__ b) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}
}  // namespace
}  // namespace xrd

extern "C" {

int64_t xrd_point_render_scratch_floats(int64_t n_points) {
  return n_points < 0 ? 0 : 7 * n_points;
}

int xrd_point_render_fwd(
    int n_rays, int n_samples, const float* points, const int64_t* neighbors,
    const int32_t* n_neighbors, const float* cloud, const float* geo_feats,
    const uint8_t* feat_mask, const float* col_feats, const float* radius,
    float radius_all, int min_nn, const float* empty_geo,
    const float* empty_col, const float* packed_geo, const float* packed_col,
    const float* z_vals, float sigmoid_coef, float* occ, uint8_t* has,
    uint64_t* relu_masks, float* rgb, float* save_c, float* save_h,
    float* save_y, float* depth, float* var, float* color,
    xrd_stream_t stream) {
  if (n_rays < 0 || n_samples < 1 || n_samples > kMaxS) return XRD_ERR_ARG;
  const int64_t m = (int64_t)n_rays * n_samples;
  if (m == 0) return XRD_OK;
  if (!occ || !has || !z_vals || !depth || !var || !color) return XRD_ERR_ARG;
  if (col_feats && (!rgb || !packed_col || !empty_col)) return XRD_ERR_ARG;
  int rc = xrd_point_geo_fwd(m, points, neighbors, n_neighbors, cloud,
                             geo_feats, feat_mask, radius, radius_all, min_nn,
                             empty_geo, packed_geo, occ, has, relu_masks,
                             stream);
  if (rc != XRD_OK) return rc;
  if (col_feats) {
    rc = xrd_point_color_fwd(m, points, neighbors, n_neighbors, cloud,
                             col_feats, radius, radius_all, min_nn, empty_col,
                             packed_col, rgb, save_c, save_h, save_y, stream);
    if (rc != XRD_OK) return rc;
  }
  return xrd_point_composite_fwd(n_rays, n_samples, col_feats ? rgb : nullptr,
                                 3, occ, 1, has, z_vals, sigmoid_coef, depth,
                                 var, color, stream);
}

int xrd_point_render_bwd(
    int n_rays, int n_samples, const float* points, const int64_t* neighbors,
    const int32_t* n_neighbors, const float* cloud, const float* geo_feats,
    const uint8_t* feat_mask, const float* col_feats, const float* radius,
    float radius_all, int min_nn, const float* empty_geo,
    const float* packed_geo, const float* packed_col, const float* z_vals,
    float sigmoid_coef, const float* occ, const uint8_t* has,
    const uint64_t* relu_masks, const float* rgb, const float* save_c,
    const float* save_h, const float* save_y, const float* g_depth,
    const float* g_var, const float* g_color, float* scratch, float* g_points,
    float* g_geo_feats, float* g_col_feats, float* g_flat, float* ops,
    float* workspace, xrd_stream_t stream) {
  if (n_rays < 0 || n_samples < 1 || n_samples > kMaxS) return XRD_ERR_ARG;
  const int64_t m = (int64_t)n_rays * n_samples;
  if (m == 0) return XRD_OK;
  if (!occ || !has || !z_vals || !scratch) return XRD_ERR_ARG;
  float* g_rgb = scratch;            // [m,3]
  float* g_occ = scratch + 3 * m;    // [m]
  float* g_pts2 = scratch + 4 * m;   // [m,3] colour path's share of d/d points
  const bool col = col_feats != nullptr;
  int rc = xrd_point_composite_bwd(n_rays, n_samples, col ? rgb : nullptr, 3,
                                   occ, 1, has, z_vals, sigmoid_coef, g_depth,
                                   g_var, col ? g_color : nullptr,
                                   col ? g_rgb : nullptr, 3, g_occ, 1, stream);
  if (rc != XRD_OK) return rc;
  if (col) {
    rc = xrd_point_color_bwd(m, points, neighbors, n_neighbors, cloud,
                             col_feats, radius, radius_all, min_nn, packed_col,
                             rgb, save_c, save_h, save_y, g_rgb,
                             g_points ? g_pts2 : nullptr, g_col_feats, g_flat,
                             ops, workspace, stream);
    if (rc != XRD_OK) return rc;
  }
  rc = xrd_point_geo_bwd(m, points, neighbors, n_neighbors, cloud, geo_feats,
                         feat_mask, radius, radius_all, min_nn, empty_geo,
                         packed_geo, relu_masks, g_occ, g_points, g_geo_feats,
                         stream);
  if (rc != XRD_OK) return rc;
  if (col && g_points) {
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((3 * m + 255) / 256)),
                       dim3(256), 0, (hipStream_t)stream, 3 * m, g_points,
                       g_pts2);
    return check_launch("xrd_point_render_bwd");
  }
  return XRD_OK;
}

}  // extern "C"
