// Small fused kernels around the render call of one NICE-SLAM iteration.  The
// reference expresses these steps as ~200 tiny PyTorch kernels per iteration
// (pixel meshgrid + gather + ray rotation + bbox filter, the L1 losses with the
// robust median mask, the pose quaternion -> matrix chain, Adam on a handful
// of pose parameters); at ~1 ms of real work per iteration the launches ARE
// the cost, so each step becomes one launch (forward) + one launch (backward).
//   xrd_sample_rays / _bwd   slam/common/common.py:39-122,188-227 (get_samples)
//                            + slam/algorithms/nice_slam.py:181-194 (bbox filter)
//   xrd_nice_loss            slam/models/conv_onet.py:145-185 (get_loss_dict)
//   xrd_pose_quat_fwd/_bwd   slam/utils/opt_pose.py:51-76 (matrix(), quat)
//   xrd_adam_dense           torch.optim.Adam on small dense tensors
//   xrd_track_best           slam/algorithms/base_algorithm.py:262-265
#include "common.h"

namespace xrd {
namespace {

// ---------------------------------------------------------------- sampling
struct SampleArgs {
  int n, W, H0, W0, wcrop;
  float fx, fy, cx, cy;
  double bound[6];
};

__global__ __launch_bounds__(256) void sample_rays_kernel(
    SampleArgs a, const int64_t* __restrict__ idx,
    const float* __restrict__ depth_img, const float* __restrict__ rgb_img,
    const float* __restrict__ c2w, float* __restrict__ rays_o,
    float* __restrict__ rays_d, float* __restrict__ tgt_d,
    float* __restrict__ tgt_rgb, uint8_t* __restrict__ keep,
    float* __restrict__ dmax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // the batch's largest kept depth: one atomic a WAVE (every kept ray issuing
  // its own was ~1000 same-address atomics a launch, ~12 ns each)
  float d_kept = 0.f;
  if (i < a.n) {
    const int64_t k = idx[i];
    const int row = a.H0 + (int)(k / a.wcrop), col = a.W0 + (int)(k % a.wcrop);
    const int64_t pix = (int64_t)row * a.W + col;
    const float d = depth_img[pix];
    const float dir[3] = {((float)col - a.cx) / a.fx, -((float)row - a.cy) / a.fy,
                          -1.f};
    float o[3], rd[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      rd[r] = dir[0] * c2w[r * 4 + 0] + dir[1] * c2w[r * 4 + 1] +
              dir[2] * c2w[r * 4 + 2];
      o[r] = c2w[r * 4 + 3];
      rays_o[i * 3 + r] = o[r];
      rays_d[i * 3 + r] = rd[r];
      tgt_rgb[i * 3 + r] = rgb_img[pix * 3 + r];
    }
    tgt_d[i] = d;
    // rays whose sensor depth lies beyond the bound are dropped
    double t_exit = 1e300;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double t0 = (a.bound[2 * r] - (double)o[r]) / (double)rd[r];
      const double t1 = (a.bound[2 * r + 1] - (double)o[r]) / (double)rd[r];
      t_exit = fmin(t_exit, fmax(t0, t1));
    }
    const bool kp = t_exit >= (double)d;
    keep[i] = kp ? 1 : 0;
    if (kp && d > 0.f) d_kept = d;
  }
  if (dmax) {
    const float m = wave_max(d_kept);
    if ((threadIdx.x & 63) == 0 && m > 0.f)
      atomicMax(reinterpret_cast<int*>(dmax), __float_as_int(m));
  }
}

// g_c2w[r][k] = sum_i g_rays_d[i][r] * dir_i[k];  g_c2w[r][3] = sum_i g_rays_o[i][r]
__global__ __launch_bounds__(256) void sample_rays_bwd_kernel(
    SampleArgs a, const int64_t* __restrict__ idx,
    const float* __restrict__ g_o, const float* __restrict__ g_d,
    float* __restrict__ g_c2w) {
  __shared__ float red[4][12];
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  for (int i = threadIdx.x; i < a.n; i += 256) {
    const int64_t k = idx[i];
    const int row = a.H0 + (int)(k / a.wcrop), col = a.W0 + (int)(k % a.wcrop);
    const float dir[3] = {((float)col - a.cx) / a.fx,
                          -((float)row - a.cy) / a.fy, -1.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float gd = g_d[i * 3 + r];
      acc[r * 4 + 0] += gd * dir[0];
      acc[r * 4 + 1] += gd * dir[1];
      acc[r * 4 + 2] += gd * dir[2];
      acc[r * 4 + 3] += g_o[i * 3 + r];
    }
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) red[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int k = threadIdx.x;
    g_c2w[k] = k < 12 ? (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]) : 0.f;
  }
}

// -------------------------------------------------------------------- loss
constexpr int kLossMax = 8192;  // rays per call handled by the one-block loss

__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
  return s;
}

__global__ __launch_bounds__(1024) void nice_loss_kernel(
    int n, int is_mapping, int use_color, int handle_dynamic, float w_color,
    const double* __restrict__ depth, const double* __restrict__ var,
    const float* __restrict__ rgb, const float* __restrict__ tgt_d,
    const float* __restrict__ tgt_rgb, const uint8_t* __restrict__ keep,
    double* __restrict__ loss, double* __restrict__ g_depth,
    float* __restrict__ g_rgb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* sorted = reinterpret_cast<double*>(smem);  // [npow2]
  __shared__ double sh[16];
  __shared__ int s_cnt;
  const int tid = threadIdx.x, T = blockDim.x;
  double thr = 1e300;
  if (!is_mapping && handle_dynamic) {
    // lower median of the residuals of the kept rays (torch.median)
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int local = 0;
    for (int i = tid; i < np2; i += T) {
      double v = 1e300;
      if (i < n && (!keep || keep[i])) {
        v = fabs((double)tgt_d[i] - depth[i]) / sqrt(var[i] + 1e-10);
        ++local;
      }
      sorted[i] = v;
    }
    // one LDS atomic per wave: 1024 same-address atomics serialise (measured:
    // 7 of this launch's 14 us at n = 200)
    const int wave_local = (int)wave_sum((float)local);
    if ((tid & 63) == 0 && wave_local) atomicAdd(&s_cnt, wave_local);
    __syncthreads();
    const int cnt = s_cnt;
    if (np2 <= 256) {
      // a tracking batch (200 rays): the order statistic by rank counting —
      // thread i counts the residuals that sort before its own (ties by
      // index), every read an LDS broadcast; 3 barriers instead of the 36
      // of the bitonic network below.  Larger batches sort (measured: rank
      // counting is slower than the network from ~1000 rays on)
      __shared__ double s_med;
      __shared__ int s_rank[256];
      // np2 <= T are both powers of two: T / np2 threads share an element,
      // each counting over its part of the batch (n = 200: 64 compares a
      // thread instead of 200 in a quarter of the waves)
      const int groups = min(T / np2, np2);
      const int i = tid & (np2 - 1), part = tid / np2, span = np2 / groups;
      if (tid == 0) s_med = 1e300;
      if (tid < np2) s_rank[tid] = 0;
      __syncthreads();
      if (i < n && part < groups && cnt > 0) {
        const double v = sorted[i];
        const int j0 = part * span, j1 = min(n, j0 + span);
        int r = 0;
#pragma unroll 4
        for (int j = j0; j < j1; ++j) {
          const double u = sorted[j];
          r += (u < v || (u == v && j < i)) ? 1 : 0;
        }
        if (r) atomicAdd(&s_rank[i], r);
      }
      __syncthreads();
      if (tid < n && cnt > 0 && s_rank[tid] == (cnt - 1) / 2)
        s_med = sorted[tid];
      __syncthreads();
      thr = cnt > 0 ? 10.0 * s_med : 1e300;
    } else {
      for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < np2; i += T) {
            const int ixj = i ^ j;
            if (ixj > i) {
              const bool up = (i & k) == 0;
              const double a = sorted[i], b = sorted[ixj];
              if ((a > b) == up) {
                sorted[i] = b;
                sorted[ixj] = a;
              }
            }
          }
          __syncthreads();
        }
      thr = cnt > 0 ? 10.0 * sorted[(cnt - 1) / 2] : 1e300;
      __syncthreads();
    }
  }
  double ld = 0.0, lc = 0.0;
  for (int i = tid; i < n; i += T) {
    const bool kp = !keep || keep[i];
    const double diff = (double)tgt_d[i] - depth[i];
    double gd = 0.0;
    bool use_c;
    if (!is_mapping) {
      const double inv = 1.0 / sqrt(var[i] + 1e-10);
      const double res = fabs(diff) * inv;
      const bool m = kp && tgt_d[i] > 0.f && (!handle_dynamic || res < thr);
      if (m) {
        ld += res;
        gd = (diff > 0 ? -1.0 : (diff < 0 ? 1.0 : 0.0)) * inv;
      }
      use_c = m && use_color;
    } else {
      const bool m = kp && tgt_d[i] > 0.f;
      if (m) {
        ld += fabs(diff);
        gd = diff > 0 ? -1.0 : (diff < 0 ? 1.0 : 0.0);
      }
      use_c = kp && use_color;
    }
    g_depth[i] = gd;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float g = 0.f;
      if (use_c) {
        const float dc = tgt_rgb[i * 3 + c] - rgb[i * 3 + c];
        lc += (double)fabsf(dc);
        g = w_color * (dc > 0.f ? -1.f : (dc < 0.f ? 1.f : 0.f));
      }
      g_rgb[i * 3 + c] = g;
    }
  }
  const double sd = block_sum(ld, sh);
  const double sc = block_sum(lc, sh);
  if (tid == 0) loss[0] = sd + (double)((float)w_color * (float)sc);
}

// -------------------------------------------------------------------- pose
// c2w[16] (row-major 4x4) from translation t[3] and quaternion q=(r,i,j,k),
// R = I + s*B(q), s = 2/|q|^2 (opt_pose.py:69 via pytorch3d quaternion_to_matrix)
__device__ __forceinline__ void quat_to_c2w(const float* __restrict__ t,
                                            const float* __restrict__ q,
                                            float* __restrict__ c2w) {
  const float r = q[0], i = q[1], j = q[2], k = q[3];
  const float s = 2.f / (r * r + i * i + j * j + k * k);
  const float R[9] = {1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                      s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                      s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)};
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) c2w[a * 4 + b] = R[a * 3 + b];
    c2w[a * 4 + 3] = t[a];
  }
  c2w[12] = c2w[13] = c2w[14] = 0.f;
  c2w[15] = 1.f;
}

__global__ void pose_quat_fwd_kernel(const float* __restrict__ t,
                                     const float* __restrict__ q,
                                     float* __restrict__ c2w) {
  if (threadIdx.x != 0) return;
  quat_to_c2w(t, q, c2w);
}

__device__ __forceinline__ void quat_c2w_bwd(const float* __restrict__ q,
                                             const float* __restrict__ g_c2w,
                                             float* __restrict__ g_t,
                                             float* __restrict__ g_q) {
  const float r = q[0], i = q[1], j = q[2], k = q[3];
  const float N = r * r + i * i + j * j + k * k, s = 2.f / N;
  float G[3][3];
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) G[a][b] = g_c2w[a * 4 + b];
    g_t[a] = g_c2w[a * 4 + 3];
  }
  const float B[3][3] = {{-(j * j + k * k), i * j - k * r, i * k + j * r},
                         {i * j + k * r, -(i * i + k * k), j * k - i * r},
                         {i * k - j * r, j * k + i * r, -(i * i + j * j)}};
  float GB = 0.f;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) GB += G[a][b] * B[a][b];
  // d(sum G:B)/dq
  const float dr = -k * G[0][1] + j * G[0][2] + k * G[1][0] - i * G[1][2] - j * G[2][0] + i * G[2][1];
  const float di = j * G[0][1] + k * G[0][2] + j * G[1][0] - 2 * i * G[1][1] - r * G[1][2] +
                   k * G[2][0] + r * G[2][1] - 2 * i * G[2][2];
  const float dj = -2 * j * G[0][0] + i * G[0][1] + r * G[0][2] + i * G[1][0] + k * G[1][2] -
                   r * G[2][0] + k * G[2][1] - 2 * j * G[2][2];
  const float dk = -2 * k * G[0][0] - r * G[0][1] + i * G[0][2] + r * G[1][0] - 2 * k * G[1][1] +
                   j * G[1][2] + i * G[2][0] + j * G[2][1];
  const float ds = -s * s;  // ds/dq_x = -s^2 * q_x
  g_q[0] = s * dr + GB * ds * r;
  g_q[1] = s * di + GB * ds * i;
  g_q[2] = s * dj + GB * ds * j;
  g_q[3] = s * dk + GB * ds * k;
}

__global__ void pose_quat_bwd_kernel(const float* __restrict__ q,
                                     const float* __restrict__ g_c2w,
                                     float* __restrict__ g_t,
                                     float* __restrict__ g_q) {
  if (threadIdx.x != 0) return;
  quat_c2w_bwd(q, g_c2w, g_t, g_q);
}

// ------------------------------------------------- sampling, F frames at once
// One launch for the F frames of a mapping window: blockIdx.y = frame.  Every
// thread rebuilds its frame's c2w from the pose parameters (7 broadcast loads,
// ~40 flops) with the arithmetic of pose_quat_fwd_kernel; the result equals
// F x (pose_quat_fwd + sample_rays) up to the last ulp of rays_d.
constexpr int kMaxFrames = 16;
struct FramePtrs {
  const float* depth[kMaxFrames];
  const float* rgb[kMaxFrames];
  const float* t[kMaxFrames];
  const float* q[kMaxFrames];
};

__global__ __launch_bounds__(256) void sample_rays_multi_kernel(
    SampleArgs a, FramePtrs fr, const int64_t* __restrict__ idx,
    float* __restrict__ c2w_out, float* __restrict__ rays_o,
    float* __restrict__ rays_d, float* __restrict__ tgt_d,
    float* __restrict__ tgt_rgb, uint8_t* __restrict__ keep,
    float* __restrict__ dmax) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float c2w[16];
  quat_to_c2w(fr.t[f], fr.q[f], c2w);
  if (i == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) c2w_out[f * 16 + k] = c2w[k];
  }
  float d_kept = 0.f;   // one dmax atomic a wave (sample_rays_kernel)
  if (i < a.n) {
    const int64_t g = (int64_t)f * a.n + i;
    const int64_t k = idx[g];
    const int row = a.H0 + (int)(k / a.wcrop), col = a.W0 + (int)(k % a.wcrop);
    const int64_t pix = (int64_t)row * a.W + col;
    const float d = fr.depth[f][pix];
    const float dir[3] = {((float)col - a.cx) / a.fx, -((float)row - a.cy) / a.fy,
                          -1.f};
    float o[3], rd[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      rd[r] = dir[0] * c2w[r * 4 + 0] + dir[1] * c2w[r * 4 + 1] +
              dir[2] * c2w[r * 4 + 2];
      o[r] = c2w[r * 4 + 3];
      rays_o[g * 3 + r] = o[r];
      rays_d[g * 3 + r] = rd[r];
      tgt_rgb[g * 3 + r] = fr.rgb[f][pix * 3 + r];
    }
    tgt_d[g] = d;
    double t_exit = 1e300;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double t0 = (a.bound[2 * r] - (double)o[r]) / (double)rd[r];
      const double t1 = (a.bound[2 * r + 1] - (double)o[r]) / (double)rd[r];
      t_exit = fmin(t_exit, fmax(t0, t1));
    }
    const bool kp = t_exit >= (double)d;
    keep[g] = kp ? 1 : 0;
    if (kp && d > 0.f) d_kept = d;
  }
  if (dmax) {
    const float m = wave_max(d_kept);
    if ((threadIdx.x & 63) == 0 && m > 0.f)
      atomicMax(reinterpret_cast<int*>(dmax), __float_as_int(m));
  }
}

// block f: g_c2w of frame f (like sample_rays_bwd_kernel), then its pose
// parameter gradients g_pose[f] = [g_t(3), g_q(4)]
__global__ __launch_bounds__(256) void sample_rays_multi_bwd_kernel(
    SampleArgs a, FramePtrs fr, const int64_t* __restrict__ idx,
    const float* __restrict__ g_o, const float* __restrict__ g_d,
    float* __restrict__ g_pose) {
  __shared__ float red[4][12];
  __shared__ float gc[16];
  const int f = blockIdx.x;
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  for (int i = threadIdx.x; i < a.n; i += 256) {
    const int64_t g = (int64_t)f * a.n + i;
    const int64_t k = idx[g];
    const int row = a.H0 + (int)(k / a.wcrop), col = a.W0 + (int)(k % a.wcrop);
    const float dir[3] = {((float)col - a.cx) / a.fx,
                          -((float)row - a.cy) / a.fy, -1.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float gd = g_d[g * 3 + r];
      acc[r * 4 + 0] += gd * dir[0];
      acc[r * 4 + 1] += gd * dir[1];
      acc[r * 4 + 2] += gd * dir[2];
      acc[r * 4 + 3] += g_o[g * 3 + r];
    }
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) red[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int k = threadIdx.x;
    gc[k] = k < 12 ? (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]) : 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    quat_c2w_bwd(fr.q[f], gc, g_pose + f * 7, g_pose + f * 7 + 3);
}

// -------------------------------------------------------------------- adam
__device__ __forceinline__ void adam_dense_body(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
    float wd, int32_t* __restrict__ step_dev, int tick) {
  __shared__ float coef[2];
  // tick: 0 = step_dev[0] is this step's number (incremented by the caller);
  // 1 = this step is step_dev[0] + 1; 2 = same, and the last block to finish
  // stores it (step_dev = {steps taken, ticket}): the last launch of a group
  // of parameters that share the counter advances it
  const int t_now = step_dev[0] + (tick ? 1 : 0);
  if (threadIdx.x < 2) {   // the two f64 pow() side by side (adam.hip)
    const bool first = threadIdx.x == 0;
    const double bc = 1.0 - pow(first ? (double)b1 : (double)b2, (double)t_now);
    coef[threadIdx.x] =
        first ? (float)((double)lr / bc) : (float)(1.0 / sqrt(bc));
  }
  __syncthreads();
  const float c0 = coef[0], c1 = coef[1];
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    if (wd != 0.f) gi += wd * pi;
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + (1.f - b2) * gi * gi;
    pi = pi - c0 * (mi / (sqrtf(vi) * c1 + eps));
  };
  // four elements a thread and step (16-byte accesses) when the arrays allow
  const bool vec = ((reinterpret_cast<uintptr_t>(p) |
                     reinterpret_cast<uintptr_t>(g) |
                     reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = pv[k], mk = mv[k], vk = vv[k];
      upd(pk, gv[k], mk, vk);
      pv[k] = pk;
      mv[k] = mk;
      vv[k] = vk;
    }
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(pi, g[i], mi, vi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
  if (tick == 2) {
    __syncthreads();
    if (threadIdx.x == 0 &&
        atomicAdd(step_dev + 1, 1) == (int)gridDim.x - 1) {
      step_dev[1] = 0;
      step_dev[0] = t_now;
    }
  }
}

__global__ __launch_bounds__(256) void adam_dense_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
    float wd, int32_t* __restrict__ step_dev, int tick) {
  adam_dense_body(p, g, m, v, n, lr, b1, b2, eps, wd, step_dev, tick);
}

// several tensors in one launch (blockIdx.y = the tensor): the Gaussian
// cloud's five tensors, a model's table + decoder + poses each have their own
// optimiser (learning rate, step count) and were one 4-8 us launch each
struct AdamDenseSets {
  xrd_adam_dense_set s[XRD_ADAM_DENSE_MAX_SETS];
};
__global__ __launch_bounds__(256) void adam_dense_multi_kernel(
    AdamDenseSets a, float b1, float b2, float eps) {
  const xrd_adam_dense_set& s = a.s[blockIdx.y];
  adam_dense_body(s.param, s.grad, s.m, s.v, s.n, s.lr, b1, b2, eps,
                  s.weight_decay, s.step_ticket, s.advance ? 2 : 1);
}

__global__ void track_best_kernel(const double* __restrict__ loss,
                                  const float* __restrict__ c2w,
                                  double* __restrict__ best_loss,
                                  float* __restrict__ best_c2w,
                                  uint8_t* __restrict__ valid) {
  const bool better = loss[0] < best_loss[0];
  __syncthreads();
  if (better) {
    if (threadIdx.x < 16) best_c2w[threadIdx.x] = c2w[threadIdx.x];
    if (threadIdx.x == 0) {
      best_loss[0] = loss[0];
      valid[0] = 1;
    }
  }
}

}  // namespace
}  // namespace xrd

using namespace xrd;


// ---------------------------------------------------------------------------
// Co-SLAM mapping batch (slam/algorithms/coslam.py:139-150, 152-210)
// ---------------------------------------------------------------------------
namespace xrd {
namespace {

// c2w [n,16] from axis-angle r[n,3] and translation t[n,3]
// (OptimizablePose.matrix(), rot_rep='axis_angle', slam/utils/opt_pose.py:51-95:
// Rodrigues R = I + sin(a) K + (1 - cos(a)) K^2, K = skew(r/a), exactly I when
// |r| <= 1e-8), one thread per pose
__global__ __launch_bounds__(64) void pose_aa_fwd_kernel(
    int n, const float* __restrict__ r3, const float* __restrict__ t3,
    float* __restrict__ c2w) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n) return;
  const float x = r3[p * 3], y = r3[p * 3 + 1], z = r3[p * 3 + 2];
  const float a = sqrtf(x * x + y * y + z * z);
  float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if (a > 1e-8f) {
    const float w0 = x / a, w1 = y / a, w2 = z / a;
    const float s = sinf(a), c1 = 1.f - cosf(a);
    const float K[9] = {0.f, -w2, w1, w2, 0.f, -w0, -w1, w0, 0.f};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        float kk = 0.f;
        for (int k = 0; k < 3; ++k) kk += K[i * 3 + k] * K[k * 3 + j];
        R[i * 3 + j] += K[i * 3 + j] * s + c1 * kk;
      }
  }
  float* M = c2w + (size_t)p * 16;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[i * 3 + j];
    M[i * 4 + 3] = t3[p * 3 + i];
  }
  M[12] = M[13] = M[14] = 0.f;
  M[15] = 1.f;
}

__global__ __launch_bounds__(64) void pose_aa_bwd_kernel(
    int n, const float* __restrict__ r3, const float* __restrict__ g_c2w,
    float* __restrict__ g_r, float* __restrict__ g_t) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n) return;
  const float* Gm = g_c2w + (size_t)p * 16;
  float G[9];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) G[i * 3 + j] = Gm[i * 4 + j];
    g_t[p * 3 + i] = Gm[i * 4 + 3];
  }
  const float x = r3[p * 3], y = r3[p * 3 + 1], z = r3[p * 3 + 2];
  const float a = sqrtf(x * x + y * y + z * z);
  float gr[3] = {0.f, 0.f, 0.f};
  if (a > 1e-8f) {
    const float w[3] = {x / a, y / a, z / a};
    const float s = sinf(a), c = cosf(a);
    const float K[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
    float sGK = 0.f, sGK2 = 0.f;  // <G, K> and <G, K^2>
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        float kk = 0.f;
        for (int k = 0; k < 3; ++k) kk += K[i * 3 + k] * K[k * 3 + j];
        sGK += G[i * 3 + j] * K[i * 3 + j];
        sGK2 += G[i * 3 + j] * kk;
      }
    // dL/dK = s G + (1-c) (G K^T + K^T G)
    float dK[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        float v = 0.f;
        for (int k = 0; k < 3; ++k)
          v += G[i * 3 + k] * K[j * 3 + k] + K[k * 3 + i] * G[k * 3 + j];
        dK[i * 3 + j] = s * G[i * 3 + j] + (1.f - c) * v;
      }
    const float dw[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
    const float da = c * sGK + s * sGK2;
    const float wd = w[0] * dw[0] + w[1] * dw[1] + w[2] * dw[2];
    for (int i = 0; i < 3; ++i) gr[i] = (dw[i] - w[i] * wd) / a + da * w[i];
  }
  for (int i = 0; i < 3; ++i) g_r[p * 3 + i] = gr[i];
}

// Keyed pseudo-random permutation of [0, 2^bits) (4-round balanced Feistel),
// walked until the value falls below n: perm(0..n_out-1) are n_out DISTINCT
// uniform-looking indices in [0, n) in O(1) each — random.sample() without the
// O(n) shuffle/sort, with n free to grow between launches.
__device__ __forceinline__ uint32_t feistel_round(uint32_t r, uint64_t key) {
  uint32_t x = r * 0x9E3779B1u + (uint32_t)key;
  x ^= x >> 15;
  x *= 0x85EBCA77u;
  x ^= x >> 13;
  x *= 0xC2B2AE3Du;
  x ^= x >> 16;
  return x ^ (uint32_t)(key >> 32);
}
__global__ __launch_bounds__(256) void sample_distinct_kernel(
    int64_t n, int n_out, int half_bits, const int64_t* __restrict__ keys,
    int64_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const uint32_t mask = (1u << half_bits) - 1u;
  uint64_t k[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) k[r] = (uint64_t)keys[r];
  uint64_t y = (uint64_t)i;
  do {
    uint32_t L = (uint32_t)(y >> half_bits) & mask, R = (uint32_t)y & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t t = L ^ (feistel_round(R, k[r]) & mask);
      L = R;
      R = t;
    }
    y = ((uint64_t)L << half_bits) | R;
  } while (y >= (uint64_t)n);
  out[i] = (int64_t)y;
}

// the same with the population size read on the device (a persistent graph
// samples a bank that grows between its replays)
__global__ __launch_bounds__(256) void sample_distinct_dev_kernel(
    const int64_t* __restrict__ n_dev, int n_out,
    const int64_t* __restrict__ keys, int64_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const int64_t n = n_dev[0];
  if (n <= i) {   // fewer than n_out items: no distinct sample exists
    out[i] = n > 0 ? (int64_t)i % n : 0;
    return;
  }
  int bits = 1;
  while ((1ll << bits) < n) ++bits;
  const int half_bits = (bits + 1) / 2;
  const uint32_t mask = (1u << half_bits) - 1u;
  uint64_t k[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) k[r] = (uint64_t)keys[r];
  uint64_t y = (uint64_t)i;
  do {
    uint32_t L = (uint32_t)(y >> half_bits) & mask, R = (uint32_t)y & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t t = L ^ (feistel_round(R, k[r]) & mask);
      L = R;
      R = t;
    }
    y = ((uint64_t)L << half_bits) | R;
  } while (y >= (uint64_t)n);
  out[i] = (int64_t)y;
}

// rays_d = R[id] dir, rays_o = t[id] for per-ray pose ids
__global__ __launch_bounds__(256) void pose_rays_fwd_kernel(
    int n, const float* __restrict__ dirs, int dir_stride,
    const int64_t* __restrict__ ids, const float* __restrict__ c2w,
    float* __restrict__ rays_o, float* __restrict__ rays_d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* M = c2w + ids[i] * 16;
  const float dx = dirs[(size_t)i * dir_stride], dy = dirs[(size_t)i * dir_stride + 1],
              dz = dirs[(size_t)i * dir_stride + 2];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    rays_d[i * 3 + a] = dx * M[4 * a] + dy * M[4 * a + 1] + dz * M[4 * a + 2];
    rays_o[i * 3 + a] = M[4 * a + 3];
  }
}

constexpr int kPoseLds = 1024;  // poses accumulated in LDS per block
__global__ __launch_bounds__(256) void pose_rays_bwd_kernel(
    int n, int n_pose, const float* __restrict__ dirs, int dir_stride,
    const int64_t* __restrict__ ids, const float* __restrict__ g_o,
    const float* __restrict__ g_d, float* __restrict__ g_c2w) {
  __shared__ float acc[kPoseLds * 12];
  const bool use_lds = n_pose <= kPoseLds;
  if (use_lds) {
    for (int k = threadIdx.x; k < n_pose * 12; k += 256) acc[k] = 0.f;
    __syncthreads();
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n;
  const int id = live ? (int)ids[i] : -1;
  float g[12];
  if (live) {
    const float d[3] = {dirs[(size_t)i * dir_stride], dirs[(size_t)i * dir_stride + 1],
                        dirs[(size_t)i * dir_stride + 2]};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) g[4 * a + b] = g_d[i * 3 + a] * d[b];
      g[4 * a + 3] = g_o[i * 3 + a];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k) g[k] = 0.f;
  }
  // rays of one frame are contiguous: usually the whole wave shares the pose
  const int id0 = __shfl(id, 0);
  if (__all(id == id0 || !live) && id0 >= 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const float v = wave_sum(g[k]);
      if ((threadIdx.x & 63) == 0) {
        if (use_lds) atomicAdd(&acc[id0 * 12 + k], v);
        else atomicAdd(g_c2w + (size_t)id0 * 16 + k, v);
      }
    }
  } else if (live) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      if (use_lds) atomicAdd(&acc[id * 12 + k], g[k]);
      else atomicAdd(g_c2w + (size_t)id * 16 + k, g[k]);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < n_pose * 12; k += 256) {
      const float v = acc[k];
      if (v != 0.f) atomicAdd(g_c2w + (size_t)(k / 12) * 16 + (k % 12), v);
    }
  }
}

// ---- pose hand-over between frames without a host round trip ---------------
// A tracking call ends with "the best pose of the frame" on the device; the
// reference turns it into the frame's pose parameters and the next frame's
// constant-velocity start on the host (slam/common/frame.py:24-36,
// slam/utils/opt_pose.py:97-110, slam/pipeline/tracker.py:185-199).  With the
// tracking iterations inside one hipGraph that host hop (device -> numpy ->
// pytorch3d-style conversions -> upload) is the GPU's idle time between two
// frames; these two single-thread kernels keep the chain on the device.
//
// pose_from_matrix: OptimizablePose.from_matrix — rotation matrix -> unit
// quaternion (r,i,j,k) through the best conditioned of the four candidates
// (largest of 1 +- m00 +- m11 +- m22), sign r >= 0, optionally -> axis-angle
// (theta = 2 atan2(|v|, r)); out = [t(3), rot(3 or 4)], float32 arithmetic in
// the order of the host formulas.  ``dev_max`` (optional): the reference's
// consistency check of an initial pose (frame.py:24-29: |c2w - matrix of the
// parameters| <= 1e-3) without its host read — the largest deviation of the
// rotation rebuilt from the quaternion is folded into dev_max[0] (NaN sticks).
__global__ void pose_from_matrix_kernel(const float* __restrict__ c2w,
                                        int quat_rep,
                                        float* __restrict__ out,
                                        float* __restrict__ dev_max) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float m[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) m[i][j] = c2w[i * 4 + j];
  const float t[4] = {1.f + m[0][0] + m[1][1] + m[2][2],
                      1.f + m[0][0] - m[1][1] - m[2][2],
                      1.f - m[0][0] + m[1][1] - m[2][2],
                      1.f - m[0][0] - m[1][1] + m[2][2]};
  int best = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (t[k] > t[best]) best = k;
  const float d = 2.0f * sqrtf(fmaxf(t[best], 1e-12f));
  float q[4];
  if (best == 0) {
    q[0] = d / 4;
    q[1] = (m[2][1] - m[1][2]) / d;
    q[2] = (m[0][2] - m[2][0]) / d;
    q[3] = (m[1][0] - m[0][1]) / d;
  } else if (best == 1) {
    q[0] = (m[2][1] - m[1][2]) / d;
    q[1] = d / 4;
    q[2] = (m[0][1] + m[1][0]) / d;
    q[3] = (m[0][2] + m[2][0]) / d;
  } else if (best == 2) {
    q[0] = (m[0][2] - m[2][0]) / d;
    q[1] = (m[0][1] + m[1][0]) / d;
    q[2] = d / 4;
    q[3] = (m[1][2] + m[2][1]) / d;
  } else {
    q[0] = (m[1][0] - m[0][1]) / d;
    q[1] = (m[0][2] + m[2][0]) / d;
    q[2] = (m[1][2] + m[2][1]) / d;
    q[3] = d / 4;
  }
  if (q[0] < 0.f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = -q[k];
  }
  if (dev_max != nullptr) {
    // quaternion_to_matrix (opt_pose.py: two_s = 2 / |q|^2)
    const float two_s =
        2.0f / (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float r[3][3] = {
        {1 - two_s * (q[2] * q[2] + q[3] * q[3]),
         two_s * (q[1] * q[2] - q[3] * q[0]),
         two_s * (q[1] * q[3] + q[2] * q[0])},
        {two_s * (q[1] * q[2] + q[3] * q[0]),
         1 - two_s * (q[1] * q[1] + q[3] * q[3]),
         two_s * (q[2] * q[3] - q[1] * q[0])},
        {two_s * (q[1] * q[3] - q[2] * q[0]),
         two_s * (q[2] * q[3] + q[1] * q[0]),
         1 - two_s * (q[1] * q[1] + q[2] * q[2])}};
    float err = dev_max[0];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float e = fabsf(r[i][j] - m[i][j]);
        // larger, or NaN; a NaN already held (an earlier frame's) stays
        if (err == err && !(e <= err)) err = e;
      }
    // (the reference compares the full 4x4, frame.py:24-29: the translation
    // column is copied through; the bottom row has to be 0 0 0 1)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float e = fabsf(c2w[12 + j] - (j == 3 ? 1.f : 0.f));
      if (err == err && !(e <= err)) err = e;
    }
    dev_max[0] = err;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) out[a] = c2w[a * 4 + 3];
  if (quat_rep) {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[3 + k] = q[k];
  } else {
    const float n = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float theta = 2.0f * atan2f(n, q[0]);
    const float scale = n < 1e-8f ? 2.0f : theta / n;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[3 + k] = q[1 + k] * scale;
  }
}

// pose_predict: constant-velocity start of the next frame,
// (prev @ inv(prev2)) @ prev.  The inverse is a general 4x4 one (Gauss-Jordan
// with partial pivoting, in double, rounded to float like numpy's float32
// LAPACK result to ~1 ulp); the two products are float32.
__global__ void pose_predict_kernel(const float* __restrict__ prev,
                                    const float* __restrict__ prev2,
                                    float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = (double)prev2[i * 4 + j];
      a[i][4 + j] = i == j ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r)
      if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 8; ++j) {
        const double tmp = a[c][j];
        a[c][j] = a[piv][j];
        a[piv][j] = tmp;
      }
    const double inv = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= inv;
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
    }
  }
  float inv2[4][4], delta[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) inv2[i][j] = (float)a[i][4 + j];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) acc += prev[i * 4 + k] * inv2[k][j];
      delta[i][j] = acc;
    }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) acc += delta[i][k] * prev[k * 4 + j];
      out[i * 4 + j] = acc;
    }
}

}  // namespace
}  // namespace xrd

extern "C" {

int xrd_sample_rays(int n, int image_width, int h0, int w0, int crop_width,
                    float fx, float fy, float cx, float cy,
                    const double* bound6, const int64_t* crop_idx,
                    const float* depth_img, const float* rgb_img,
                    const float* c2w, float* rays_o, float* rays_d,
                    float* tgt_d, float* tgt_rgb, uint8_t* keep, float* dmax,
                    xrd_stream_t stream) {
  if (n < 0 || image_width < 1 || crop_width < 1 || !bound6) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!crop_idx || !depth_img || !rgb_img || !c2w || !rays_o || !rays_d ||
      !tgt_d || !tgt_rgb || !keep)
    return XRD_ERR_ARG;
  SampleArgs a{n, image_width, h0, w0, crop_width, fx, fy, cx, cy, {}};
  for (int k = 0; k < 6; ++k) a.bound[k] = bound6[k];
  hipLaunchKernelGGL(sample_rays_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, a, crop_idx, depth_img, rgb_img, c2w,
                     rays_o, rays_d, tgt_d, tgt_rgb, keep, dmax);
  return check_launch("xrd_sample_rays");
}

int xrd_sample_rays_bwd(int n, int image_width, int h0, int w0, int crop_width,
                        float fx, float fy, float cx, float cy,
                        const int64_t* crop_idx, const float* g_rays_o,
                        const float* g_rays_d, float* g_c2w,
                        xrd_stream_t stream) {
  if (n < 0 || crop_width < 1 || !g_c2w) return XRD_ERR_ARG;
  if (n > 0 && (!crop_idx || !g_rays_o || !g_rays_d)) return XRD_ERR_ARG;
  SampleArgs a{n, image_width, h0, w0, crop_width, fx, fy, cx, cy, {}};
  hipLaunchKernelGGL(sample_rays_bwd_kernel, dim3(1), dim3(256), 0,
                     (hipStream_t)stream, a, crop_idx, g_rays_o, g_rays_d,
                     g_c2w);
  return check_launch("xrd_sample_rays_bwd");
}

static bool fill_frames(FramePtrs& fr, int F, const float* const* depth,
                        const float* const* rgb, const float* const* t,
                        const float* const* q, bool need_images) {
  for (int f = 0; f < F; ++f) {
    if (!t[f] || !q[f]) return false;
    if (need_images && (!depth[f] || !rgb[f])) return false;
    fr.depth[f] = need_images ? depth[f] : nullptr;
    fr.rgb[f] = need_images ? rgb[f] : nullptr;
    fr.t[f] = t[f];
    fr.q[f] = q[f];
  }
  return true;
}

int xrd_sample_rays_multi(int n_frames, int n, int image_width, int h0, int w0,
                          int crop_width, float fx, float fy, float cx,
                          float cy, const double* bound6,
                          const int64_t* crop_idx,
                          const float* const* depth_imgs,
                          const float* const* rgb_imgs,
                          const float* const* pose_t,
                          const float* const* pose_q, float* c2w_out,
                          float* rays_o, float* rays_d, float* tgt_d,
                          float* tgt_rgb, uint8_t* keep, float* dmax,
                          xrd_stream_t stream) {
  if (n_frames < 1 || n < 1 || image_width < 1 || crop_width < 1 || !bound6 ||
      !crop_idx || !depth_imgs || !rgb_imgs || !pose_t || !pose_q ||
      !c2w_out || !rays_o || !rays_d || !tgt_d || !tgt_rgb || !keep)
    return XRD_ERR_ARG;
  if (n_frames > kMaxFrames) return XRD_ERR_UNSUPPORTED;
  FramePtrs fr{};
  if (!fill_frames(fr, n_frames, depth_imgs, rgb_imgs, pose_t, pose_q, true))
    return XRD_ERR_ARG;
  SampleArgs a{n, image_width, h0, w0, crop_width, fx, fy, cx, cy, {}};
  for (int k = 0; k < 6; ++k) a.bound[k] = bound6[k];
  hipLaunchKernelGGL(sample_rays_multi_kernel,
                     dim3((n + 255) / 256, n_frames), dim3(256), 0,
                     (hipStream_t)stream, a, fr, crop_idx, c2w_out, rays_o,
                     rays_d, tgt_d, tgt_rgb, keep, dmax);
  return check_launch("xrd_sample_rays_multi");
}

int xrd_sample_rays_multi_bwd(int n_frames, int n, int image_width, int h0,
                              int w0, int crop_width, float fx, float fy,
                              float cx, float cy, const int64_t* crop_idx,
                              const float* const* pose_t,
                              const float* const* pose_q,
                              const float* g_rays_o, const float* g_rays_d,
                              float* g_pose, xrd_stream_t stream) {
  if (n_frames < 1 || n < 1 || crop_width < 1 || !crop_idx || !pose_t ||
      !pose_q || !g_rays_o || !g_rays_d || !g_pose)
    return XRD_ERR_ARG;
  if (n_frames > kMaxFrames) return XRD_ERR_UNSUPPORTED;
  FramePtrs fr{};
  if (!fill_frames(fr, n_frames, nullptr, nullptr, pose_t, pose_q, false))
    return XRD_ERR_ARG;
  SampleArgs a{n, image_width, h0, w0, crop_width, fx, fy, cx, cy, {}};
  hipLaunchKernelGGL(sample_rays_multi_bwd_kernel, dim3(n_frames), dim3(256),
                     0, (hipStream_t)stream, a, fr, crop_idx, g_rays_o,
                     g_rays_d, g_pose);
  return check_launch("xrd_sample_rays_multi_bwd");
}

int xrd_nice_loss(int n, int is_mapping, int use_color, int handle_dynamic,
                  float w_color, const double* depth, const double* var,
                  const float* rgb, const float* tgt_d, const float* tgt_rgb,
                  const uint8_t* keep, double* loss, double* g_depth,
                  float* g_rgb, xrd_stream_t stream) {
  if (n < 1 || !depth || !var || !rgb || !tgt_d || !tgt_rgb || !loss ||
      !g_depth || !g_rgb)
    return XRD_ERR_ARG;
  if (n > kLossMax) return XRD_ERR_UNSUPPORTED;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  const size_t lds = (size_t)np2 * sizeof(double);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nice_loss_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize,
                        kLossMax * (int)sizeof(double));
    attr = true;
  }
  hipLaunchKernelGGL(nice_loss_kernel, dim3(1), dim3(1024), lds,
                     (hipStream_t)stream, n, is_mapping, use_color,
                     handle_dynamic, w_color, depth, var, rgb, tgt_d, tgt_rgb,
                     keep, loss, g_depth, g_rgb);
  return check_launch("xrd_nice_loss");
}

int xrd_pose_from_matrix(int rot_rep, const float* c2w16, float* vec,
                         xrd_stream_t stream) {
  if (!c2w16 || !vec) return XRD_ERR_ARG;
  if (rot_rep != XRD_ROT_AXIS_ANGLE && rot_rep != XRD_ROT_QUAT)
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(pose_from_matrix_kernel, dim3(1), dim3(64), 0,
                     (hipStream_t)stream, c2w16, rot_rep == XRD_ROT_QUAT, vec,
                     (float*)nullptr);
  return check_launch("xrd_pose_from_matrix");
}

int xrd_pose_from_matrix_checked(int rot_rep, const float* c2w16, float* vec,
                                 float* dev_max, xrd_stream_t stream) {
  if (!c2w16 || !vec || !dev_max) return XRD_ERR_ARG;
  if (rot_rep != XRD_ROT_AXIS_ANGLE && rot_rep != XRD_ROT_QUAT)
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(pose_from_matrix_kernel, dim3(1), dim3(64), 0,
                     (hipStream_t)stream, c2w16, rot_rep == XRD_ROT_QUAT, vec,
                     dev_max);
  return check_launch("xrd_pose_from_matrix_checked");
}

int xrd_pose_predict(const float* prev16, const float* prev2_16,
                     float* next16, xrd_stream_t stream) {
  if (!prev16 || !prev2_16 || !next16) return XRD_ERR_ARG;
  hipLaunchKernelGGL(pose_predict_kernel, dim3(1), dim3(64), 0,
                     (hipStream_t)stream, prev16, prev2_16, next16);
  return check_launch("xrd_pose_predict");
}

int xrd_pose_quat_fwd(const float* t3, const float* q4, float* c2w16,
                      xrd_stream_t stream) {
  if (!t3 || !q4 || !c2w16) return XRD_ERR_ARG;
  hipLaunchKernelGGL(pose_quat_fwd_kernel, dim3(1), dim3(64), 0,
                     (hipStream_t)stream, t3, q4, c2w16);
  return check_launch("xrd_pose_quat_fwd");
}

int xrd_pose_quat_bwd(const float* q4, const float* g_c2w16, float* g_t3,
                      float* g_q4, xrd_stream_t stream) {
  if (!q4 || !g_c2w16 || !g_t3 || !g_q4) return XRD_ERR_ARG;
  hipLaunchKernelGGL(pose_quat_bwd_kernel, dim3(1), dim3(64), 0,
                     (hipStream_t)stream, q4, g_c2w16, g_t3, g_q4);
  return check_launch("xrd_pose_quat_bwd");
}

int xrd_adam_dense(float* param, const float* grad, float* m, float* v,
                   int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, const int32_t* step_dev,
                   xrd_stream_t stream) {
  if (n < 0 || !step_dev) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!param || !grad || !m || !v) return XRD_ERR_ARG;
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(adam_dense_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, param, grad, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, const_cast<int32_t*>(step_dev),
                     0);
  return check_launch("xrd_adam_dense");
}

int xrd_adam_dense_tick(float* param, const float* grad, float* m, float* v,
                        int64_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int32_t* step_ticket,
                        int advance, xrd_stream_t stream) {
  if (n < 0 || !step_ticket) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!param || !grad || !m || !v) return XRD_ERR_ARG;
  // (the ticket costs one same-address atomic a block)
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(adam_dense_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, param, grad, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, step_ticket, advance ? 2 : 1);
  return check_launch("xrd_adam_dense_tick");
}

int xrd_adam_dense_multi(int n_sets, const xrd_adam_dense_set* sets,
                         float beta1, float beta2, float eps,
                         xrd_stream_t stream) {
  if (n_sets < 0 || n_sets > XRD_ADAM_DENSE_MAX_SETS || (n_sets && !sets))
    return XRD_ERR_ARG;
  AdamDenseSets a = {};
  int n = 0;
  int64_t most = 0;
  for (int i = 0; i < n_sets; ++i) {
    const xrd_adam_dense_set& s = sets[i];
    if (s.n < 0 || !s.step_ticket) return XRD_ERR_ARG;
    // every set must own its counter: a set that advances a counter another
    // set of the same launch still reads would race
    for (int j = 0; j < i; ++j)
      if (sets[j].step_ticket == s.step_ticket) return XRD_ERR_ARG;
    if (s.n == 0) continue;
    if (!s.param || !s.grad || !s.m || !s.v) return XRD_ERR_ARG;
    a.s[n++] = s;
    most = s.n > most ? s.n : most;
  }
  if (n == 0) return XRD_OK;
  int64_t blocks = (most + 1023) / 1024;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(adam_dense_multi_kernel,
                     dim3((unsigned)blocks, (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, a, beta1, beta2, eps);
  return check_launch("xrd_adam_dense_multi");
}

int xrd_track_best(const double* loss, const float* c2w16, double* best_loss,
                   float* best_c2w16, uint8_t* valid, xrd_stream_t stream) {
  if (!loss || !c2w16 || !best_loss || !best_c2w16 || !valid) return XRD_ERR_ARG;
  hipLaunchKernelGGL(track_best_kernel, dim3(1), dim3(64), 0,
                     (hipStream_t)stream, loss, c2w16, best_loss, best_c2w16,
                     valid);
  return check_launch("xrd_track_best");
}

int xrd_pose_aa_fwd(int n, const float* r3, const float* t3, float* c2w16,
                    xrd_stream_t stream) {
  if (n < 1 || !r3 || !t3 || !c2w16) return XRD_ERR_ARG;
  hipLaunchKernelGGL(pose_aa_fwd_kernel, dim3((n + 63) / 64), dim3(64), 0,
                     (hipStream_t)stream, n, r3, t3, c2w16);
  return check_launch("xrd_pose_aa_fwd");
}

int xrd_pose_aa_bwd(int n, const float* r3, const float* g_c2w16, float* g_r3,
                    float* g_t3, xrd_stream_t stream) {
  if (n < 1 || !r3 || !g_c2w16 || !g_r3 || !g_t3) return XRD_ERR_ARG;
  hipLaunchKernelGGL(pose_aa_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0,
                     (hipStream_t)stream, n, r3, g_c2w16, g_r3, g_t3);
  return check_launch("xrd_pose_aa_bwd");
}

int xrd_sample_distinct(int64_t n_total, int n_out, const int64_t* keys4,
                        int64_t* out_idx, xrd_stream_t stream) {
  if (n_total < 1 || n_out < 0 || n_out > n_total || !keys4 || (n_out && !out_idx))
    return XRD_ERR_ARG;
  if (n_total > (1ll << 40)) return XRD_ERR_UNSUPPORTED;
  if (n_out == 0) return XRD_OK;
  int bits = 1;
  while ((1ll << bits) < n_total) ++bits;
  const int half = (bits + 1) / 2;
  hipLaunchKernelGGL(sample_distinct_kernel, dim3((n_out + 255) / 256), dim3(256),
                     0, (hipStream_t)stream, n_total, n_out, half, keys4, out_idx);
  return check_launch("xrd_sample_distinct");
}

int xrd_sample_distinct_dev(const int64_t* n_total, int n_out,
                            const int64_t* keys4, int64_t* out_idx,
                            xrd_stream_t stream) {
  if (n_out < 0 || !n_total || !keys4 || (n_out && !out_idx))
    return XRD_ERR_ARG;
  if (n_out == 0) return XRD_OK;
  hipLaunchKernelGGL(sample_distinct_dev_kernel, dim3((n_out + 255) / 256),
                     dim3(256), 0, (hipStream_t)stream, n_total, n_out, keys4,
                     out_idx);
  return check_launch("xrd_sample_distinct_dev");
}

// Co-SLAM mapping batch (slam/algorithms/coslam.py:139-150,152-210): rows of
// the keyframe ray bank at bank_idx + the current frame's pixels pix ->
// rows [n_bank + n_cur, 7] = (camera-frame direction, rgb, depth) and the pose
// id of every row (bank row / rays per keyframe; *cur_id for the current
// frame).  One launch for two index gathers, a floor division, three image
// gathers and four concatenations.
namespace xrd {
namespace {
__global__ __launch_bounds__(256) void coslam_map_rows_kernel(
    int n_bank, const int64_t* __restrict__ bank_idx,
    const float* __restrict__ bank, int rays_per_kf, int n_cur,
    const int64_t* __restrict__ pix, const float* __restrict__ dirs,
    const float* __restrict__ rgb, const float* __restrict__ depth,
    const int64_t* __restrict__ cur_id, float* __restrict__ rows,
    int64_t* __restrict__ ids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_bank + n_cur) return;
  float v[7];
  int64_t id;
  if (i < n_bank) {
    const int64_t r = bank_idx[i];
#pragma unroll
    for (int c = 0; c < 7; ++c) v[c] = bank[r * 7 + c];
    id = r / rays_per_kf;
  } else {
    const int64_t p = pix[i - n_bank];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      v[c] = dirs[p * 3 + c];
      v[3 + c] = rgb[p * 3 + c];
    }
    v[6] = depth[p];
    id = cur_id[0];
  }
#pragma unroll
  for (int c = 0; c < 7; ++c) rows[(size_t)i * 7 + c] = v[c];
  ids[i] = id;
}
}  // namespace
}  // namespace xrd

int xrd_coslam_map_rows(int n_bank, const int64_t* bank_idx, const float* bank,
                        int rays_per_keyframe, int n_cur, const int64_t* pix,
                        const float* ray_dirs, const float* rgb,
                        const float* depth, const int64_t* cur_id, float* rows,
                        int64_t* ids, xrd_stream_t stream) {
  if (n_bank < 0 || n_cur < 0 || rays_per_keyframe < 1) return XRD_ERR_ARG;
  if (!rows || !ids) return XRD_ERR_ARG;
  if (n_bank > 0 && (!bank_idx || !bank)) return XRD_ERR_ARG;
  if (n_cur > 0 && (!pix || !ray_dirs || !rgb || !depth || !cur_id))
    return XRD_ERR_ARG;
  const int n = n_bank + n_cur;
  if (n == 0) return XRD_OK;
  hipLaunchKernelGGL(xrd::coslam_map_rows_kernel, dim3((n + 255) / 256),
                     dim3(256), 0, (hipStream_t)stream, n_bank, bank_idx, bank,
                     rays_per_keyframe, n_cur, pix, ray_dirs, rgb, depth,
                     cur_id, rows, ids);
  return xrd::check_launch("xrd_coslam_map_rows");
}

int xrd_pose_rays_fwd(int n, const float* dirs, int dir_stride,
                      const int64_t* pose_ids, const float* c2w, float* rays_o,
                      float* rays_d, xrd_stream_t stream) {
  if (n < 0 || dir_stride < 3 || (n && (!dirs || !pose_ids || !c2w || !rays_o ||
                                        !rays_d)))
    return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  hipLaunchKernelGGL(pose_rays_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, n, dirs, dir_stride, pose_ids, c2w,
                     rays_o, rays_d);
  return check_launch("xrd_pose_rays_fwd");
}

int xrd_pose_rays_bwd(int n, int n_pose, const float* dirs, int dir_stride,
                      const int64_t* pose_ids, const float* g_rays_o,
                      const float* g_rays_d, float* g_c2w, xrd_stream_t stream) {
  if (n < 0 || n_pose < 1 || dir_stride < 3 || !g_c2w ||
      (n && (!dirs || !pose_ids || !g_rays_o || !g_rays_d)))
    return XRD_ERR_ARG;
  int rc = zero_floats(g_c2w, (size_t)n_pose * 16, stream);
  if (rc != XRD_OK || n == 0) return rc;
  hipLaunchKernelGGL(pose_rays_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, n, n_pose, dirs, dir_stride, pose_ids,
                     g_rays_o, g_rays_d, g_c2w);
  return check_launch("xrd_pose_rays_bwd");
}

}  // extern "C"
