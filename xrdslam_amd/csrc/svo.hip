// Ray / sparse-voxel-octree intersection and inverse-CDF ray sampling for
// Vox-Fusion on gfx950 — the two LIVE kernels of the reference's `grid`
// extension:
//   svo_intersect_point_kernel   third_party/sparse_voxels/src/intersect_gpu.cu:191-270
//   inverse_cdf_sampling_kernel  third_party/sparse_voxels/src/sample_gpu.cu:133-239
// Semantics (hit order, n_max cut-off, sentinels, the sampling kernel's
// quirks) follow the reference exactly; voxel ids are bit-exact against
// oracle/svo_oracle.c.  What is different is the mapping to the machine:
//   * the reference launches <<<B, 2^floor(log2 M)>>> — 4 threads per block
//     for 1024 rays — and needs the octree replicated B times; here one WAVE
//     owns one ray (the 8 children of a node are tested on 8 lanes), a 2-D
//     grid covers rays x batches, and a tree may be shared by all batches
//     (tree_batch_stride = 0);
//   * the walk is level-synchronous (svo_intersect.h: one or two L2 round
//     trips a LEVEL instead of one per entered node; the reference's output
//     order is restored from a path key); the depth-first walk — its stack
//     (int[256] of scratch per thread in the reference) in LDS, holding only
//     nodes the ray is known to enter — remains for what the first gives up
//     on.
#include "common.h"
#include "svo_intersect.h"
#include "svo_sample.h"

#pragma clang fp contract(off)  // keep the float expressions as written

namespace xrd {
namespace {

constexpr int kRaysPerBlock = 64;

// One WAVE per ray (body: svo_intersect.h).
__global__ __launch_bounds__(kRaysPerBlock * 4) void svo_intersect_kernel(
    int n, int m, float voxelsize, int n_max, int64_t tree_stride,
    const float* __restrict__ ray_start, const float* __restrict__ ray_dir,
    const float* __restrict__ points, const int* __restrict__ children,
    int* __restrict__ idx, float* __restrict__ min_depth,
    float* __restrict__ max_depth, int* __restrict__ overflow) {
  __shared__ int lds[4][kSvoLds];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bi = blockIdx.y;
  const int j = blockIdx.x * 4 + wave;
  if (j >= m) return;
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
  bool ovf = false;
  int cnt = 0;
  // level by level; the depth-first walk only where that one gives up
  const bool done = svo_intersect_ray_bfs(
      lane, lds[wave], o, d, P, C, voxelsize, n_max,
      [&](int n_hit, const int* node, const float* lo, const float* hi) {
        cnt = n_hit;
        for (int l = lane; l < n_hit; l += 64) {
          I[l] = node[l];
          MN[l] = lo[l];
          MX[l] = hi[l];
        }
      });
  if (!done) {
    int* s = lds[wave];
    cnt = svo_intersect_ray(
        lane, s, s + kSvoStack, reinterpret_cast<float*>(s + 2 * kSvoStack),
        reinterpret_cast<float*>(s + 3 * kSvoStack), s + 4 * kSvoStack, o, d,
        P, C, voxelsize, n_max, ovf,
        [&](int slot, int node, float lo, float hi) {
          if (lane == 0) {
            I[slot] = node;
            MN[slot] = lo;
            MX[slot] = hi;
          }
        });
  }
  if (ovf && overflow && lane == 0) *overflow = 1;
  for (int l = cnt + lane; l < n_max; l += 64) I[l] = -1;  // unused slots
  (void)n;
}

// Inverse-CDF sampling, one WAVE per ray (body: svo_sample.h).  Writes beyond
// max_steps are dropped and pts_idx reads one past the buffer (the reference
// performs both) return -1.
constexpr int kCdfWaves = 4;

__global__ __launch_bounds__(kCdfWaves * 64) void inverse_cdf_kernel(
    int num_rays, int max_hits, int max_steps, float fixed_step_size,
    int64_t pi_total, const int* __restrict__ pts_idx,
    const float* __restrict__ min_depth, const float* __restrict__ max_depth,
    const float* __restrict__ uniform_noise, const float* __restrict__ probs,
    const float* __restrict__ steps, int* __restrict__ sampled_idx,
    float* __restrict__ sampled_depth, float* __restrict__ sampled_dists) {
  extern __shared__ float cdf_lds[];  // [kCdfWaves][max_hits]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bi = blockIdx.y;
  const int j = blockIdx.x * kCdfWaves + wave;
  if (j >= num_rays) return;
  const int64_t boff = (int64_t)bi * num_rays * max_hits;
  const int* PI = pts_idx + boff;
  const int H = j * max_hits, K = j * max_steps;
  const float* UN = uniform_noise + (int64_t)bi * num_rays * max_steps + K;
  int* SI = sampled_idx + (int64_t)bi * num_rays * max_steps + K;
  float* SD = sampled_depth + (int64_t)bi * num_rays * max_steps + K;
  float* SS = sampled_dists + (int64_t)bi * num_rays * max_steps + K;
  inverse_cdf_ray(
      lane, cdf_lds + wave * max_hits, max_hits, num_rays, H,
      min_depth + boff + H, max_depth + boff + H, probs + boff + H,
      steps[(int64_t)bi * num_rays + j], fixed_step_size,
      [&](int i) -> int { return (boff + i < pi_total) ? PI[i] : -1; },
      [&](int c) -> float { return UN[c]; },
      [&](int slot, int id, float zhi, float zlo) {
        if (slot < max_steps) {
          SI[slot] = id;
          SS[slot] = zhi - zlo;
          SD[slot] = (zhi + zlo) * 0.5f;
        }
      });
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
  const dim3 grid((m_rays + 3) / 4, b);
  hipLaunchKernelGGL(svo_intersect_kernel, grid, dim3(kRaysPerBlock * 4), 0,
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
  const dim3 grid((num_rays + kCdfWaves - 1) / kCdfWaves, b);
  const size_t lds = (size_t)kCdfWaves * max_hits * sizeof(float);
  if (lds > 60 * 1024) return XRD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(inverse_cdf_kernel, grid, dim3(kCdfWaves * 64), lds,
                     (hipStream_t)stream, num_rays, max_hits, max_steps,
                     fixed_step_size, (int64_t)b * num_rays * max_hits, pts_idx,
                     min_depth, max_depth, uniform_noise, probs, steps,
                     sampled_idx, sampled_depth, sampled_dists);
  return check_launch("xrd_inverse_cdf_sampling");
}

}  // extern "C"
