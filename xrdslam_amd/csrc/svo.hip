// Ray / sparse-voxel-octree intersection and inverse-CDF ray sampling for
// Vox-Fusion on gfx950 — the two LIVE kernels of the reference's `grid`
// extension:
//   svo_intersect_point_kernel   third_party/sparse_voxels/src/intersect_gpu.cu:191-270
//   inverse_cdf_sampling_kernel  third_party/sparse_voxels/src/sample_gpu.cu:133-239
// Semantics (hit order, n_max cut-off, sentinels, the sampling kernel's
// quirks) follow the reference exactly; voxel ids are bit-exact against
// oracle/svo_oracle.c.  What is different is the mapping to the machine:
//   * the reference launches <<<B, 2^floor(log2 M)>>> — 4 threads per block
//     for 1024 rays — and needs the octree replicated B times; here one thread
//     owns one ray in 64-thread blocks over a 2-D grid, and a tree may be
//     shared by all batches (tree_batch_stride = 0);
//   * the DFS stack (int[256] of scratch per thread in the reference) lives in
//     LDS, transposed so that a wave's pushes hit 64 different banks.
#include "common.h"

#pragma clang fp contract(off)  // keep the float expressions as written

namespace xrd {
namespace {

constexpr int kStack = 128;   // >= 1 + 7 * levels; 256^3 trees need 57
constexpr int kRaysPerBlock = 64;

__device__ __forceinline__ void ray_aabb(const float (&o)[3],
                                         const float (&d)[3], const float* c,
                                         float half, float& lo, float& hi) {
  float f_low = 0.f, f_high = 100000.f;
  lo = hi = -1.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float inv = 1.0f / d[k];
    float a = (c[k] - half - o[k]) * inv;
    float b = (c[k] + half - o[k]) * inv;
    if (b < a) {
      const float t = a;
      a = b;
      b = t;
    }
    if (b < f_low) return;
    if (a > f_high) return;
    f_low = (a > f_low) ? a : f_low;
    f_high = (b < f_high) ? b : f_high;
    if (f_low > f_high) return;
  }
  lo = f_low;
  hi = f_high;
}

__global__ __launch_bounds__(kRaysPerBlock) void svo_intersect_kernel(
    int n, int m, float voxelsize, int n_max, int64_t tree_stride,
    const float* __restrict__ ray_start, const float* __restrict__ ray_dir,
    const float* __restrict__ points, const int* __restrict__ children,
    int* __restrict__ idx, float* __restrict__ min_depth,
    float* __restrict__ max_depth, int* __restrict__ overflow) {
  __shared__ int stack[kStack][kRaysPerBlock];
  const int bi = blockIdx.y;
  const int j = blockIdx.x * kRaysPerBlock + threadIdx.x;
  if (j >= m) return;
  const int lane = threadIdx.x;
  const float* P = points + (int64_t)bi * tree_stride * 3;
  const int* C = children + (int64_t)bi * tree_stride * 9;
  const int64_t rbase = ((int64_t)bi * m + j);
  float o[3], d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o[k] = ray_start[rbase * 3 + k];
    d[k] = ray_dir[rbase * 3 + k];
  }
  int* I = idx + rbase * n_max;
  float* MN = min_depth + rbase * n_max;
  float* MX = max_depth + rbase * n_max;
  for (int l = 0; l < n_max; ++l) I[l] = -1;
  const float half_voxel = voxelsize * 0.5;
  int ptr = 0, cnt = 0;
  stack[0][lane] = 0;  // root is node 0
  while (ptr > -1 && cnt < n_max) {
    const int k = stack[ptr][lane];
    const int side = C[k * 9 + 8];
    float lo, hi;
    ray_aabb(o, d, P + k * 3, half_voxel * (float)side, lo, hi);
    ptr--;
    if (lo > -1.0f) {
      if (side == 1) {  // terminal node
        I[cnt] = k;
        MN[cnt] = lo;
        MX[cnt] = hi;
        ++cnt;
        continue;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = C[k * 9 + u];
        if (c > -1) {
          if (ptr + 1 >= kStack) {
            if (overflow) *overflow = 1;
          } else {
            stack[++ptr][lane] = c;
          }
        }
      }
    }
  }
  (void)n;
}

__global__ __launch_bounds__(64) void inverse_cdf_kernel(
    int num_rays, int max_hits, int max_steps, float fixed_step_size,
    const int* __restrict__ pts_idx, const float* __restrict__ min_depth,
    const float* __restrict__ max_depth,
    const float* __restrict__ uniform_noise, const float* __restrict__ probs,
    const float* __restrict__ steps, int* __restrict__ sampled_idx,
    float* __restrict__ sampled_depth, float* __restrict__ sampled_dists) {
  const int bi = blockIdx.y;
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= num_rays) return;
  const int* PI = pts_idx + (int64_t)bi * num_rays * max_hits;
  const float* MN = min_depth + (int64_t)bi * num_rays * max_hits;
  const float* MX = max_depth + (int64_t)bi * num_rays * max_hits;
  const float* PR = probs + (int64_t)bi * num_rays * max_hits;
  const float* ST = steps + (int64_t)bi * num_rays;
  const float* UN = uniform_noise + (int64_t)bi * num_rays * max_steps;
  int* SI = sampled_idx + (int64_t)bi * num_rays * max_steps;
  float* SD = sampled_depth + (int64_t)bi * num_rays * max_steps;
  float* SS = sampled_dists + (int64_t)bi * num_rays * max_steps;
  const int H = j * max_hits, K = j * max_steps;
  int curr_bin = 0, s = 0;
  float curr_min_depth = MN[H], curr_max_depth = MX[H];
  float curr_min_cdf = 0, curr_max_cdf = PR[H];
  float step_size = 1.0 / ST[j];
  float z_low = curr_min_depth;
  const int total_steps = (int)ceil((double)ST[j]);
  bool done = false;
  if (fixed_step_size > 0.0) step_size = fixed_step_size;
  for (int curr_step = 0; curr_step < total_steps; curr_step++) {
    const float curr_cdf = ((float)curr_step + UN[K + curr_step]) * step_size;
    while (curr_cdf > curr_max_cdf) {
      SI[K + s] = PI[H + curr_bin];
      SS[K + s] = (curr_max_depth - z_low);
      SD[K + s] = (curr_max_depth + z_low) * .5;
      curr_bin++;
      s++;
      if ((curr_bin >= max_hits) || (PI[H + curr_bin] == -1)) {
        done = true;
        break;
      }
      curr_min_depth = MN[H + curr_bin];
      curr_max_depth = MX[H + curr_bin];
      curr_min_cdf = curr_max_cdf;
      curr_max_cdf = curr_max_cdf + PR[H + curr_bin];
      z_low = curr_min_depth;
    }
    if (done) break;
    const float u = (curr_cdf - curr_min_cdf) / (curr_max_cdf - curr_min_cdf);
    const float z = curr_min_depth + u * (curr_max_depth - curr_min_depth);
    SI[K + s] = PI[H + curr_bin];
    SS[K + s] = (z - z_low);
    SD[K + s] = (z + z_low) * .5;
    z_low = z;
    s++;
  }
  // remaining bins; the reference's "(~done)" is always true and its
  // termination test reads pts_idx WITHOUT the ray offset (sample_gpu.cu:224,231)
  while ((z_low < curr_max_depth) && (num_rays > (H + curr_bin))) {
    SI[K + s] = PI[H + curr_bin];
    SS[K + s] = (curr_max_depth - z_low);
    SD[K + s] = (curr_max_depth + z_low) * .5;
    curr_bin++;
    s++;
    if ((curr_bin >= max_hits) || (PI[curr_bin] == -1)) break;
    curr_min_depth = MN[H + curr_bin];
    curr_max_depth = MX[H + curr_bin];
    z_low = curr_min_depth;
  }
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int xrd_svo_intersect(int b, int n_nodes, int m_rays, float voxelsize,
                      int n_max, int tree_shared, const float* ray_start,
                      const float* ray_dir, const float* points,
                      const int32_t* children, int32_t* idx, float* min_depth,
                      float* max_depth, int32_t* overflow_flag,
                      xrd_stream_t stream) {
  if (b < 0 || n_nodes < 1 || m_rays < 0 || n_max < 1) return XRD_ERR_ARG;
  if (b == 0 || m_rays == 0) return XRD_OK;
  if (!ray_start || !ray_dir || !points || !children || !idx || !min_depth ||
      !max_depth)
    return XRD_ERR_ARG;
  if (b > 65535) return XRD_ERR_UNSUPPORTED;
  const dim3 grid((m_rays + kRaysPerBlock - 1) / kRaysPerBlock, b);
  hipLaunchKernelGGL(svo_intersect_kernel, grid, dim3(kRaysPerBlock), 0,
                     (hipStream_t)stream, n_nodes, m_rays, voxelsize, n_max,
                     (int64_t)(tree_shared ? 0 : n_nodes), ray_start, ray_dir,
                     points, children, idx, min_depth, max_depth,
                     overflow_flag);
  return check_launch("xrd_svo_intersect");
}

int xrd_inverse_cdf_sampling(int b, int num_rays, int max_hits, int max_steps,
                             float fixed_step_size, const int32_t* pts_idx,
                             const float* min_depth, const float* max_depth,
                             const float* uniform_noise, const float* probs,
                             const float* steps, int32_t* sampled_idx,
                             float* sampled_depth, float* sampled_dists,
                             xrd_stream_t stream) {
  if (b < 0 || num_rays < 0 || max_hits < 1 || max_steps < 1) return XRD_ERR_ARG;
  if (b == 0 || num_rays == 0) return XRD_OK;
  if (!pts_idx || !min_depth || !max_depth || !uniform_noise || !probs ||
      !steps || !sampled_idx || !sampled_depth || !sampled_dists)
    return XRD_ERR_ARG;
  if (b > 65535) return XRD_ERR_UNSUPPORTED;
  const dim3 grid((num_rays + 63) / 64, b);
  hipLaunchKernelGGL(inverse_cdf_kernel, grid, dim3(64), 0,
                     (hipStream_t)stream, num_rays, max_hits, max_steps,
                     fixed_step_size, pts_idx, min_depth, max_depth,
                     uniform_noise, probs, steps, sampled_idx, sampled_depth,
                     sampled_dists);
  return check_launch("xrd_inverse_cdf_sampling");
}

}  // extern "C"
