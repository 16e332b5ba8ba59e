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

// Inverse-CDF sampling, one WAVE per ray (sample_gpu.cu:133-239 walks a ray's
// steps serially on one thread, carrying (bin, z_low); tests/
// svo_parallel_model.py states and checks the re-formulation used here):
//   * cum[b] = serial float prefix sum of the ray's probs (same addition
//     order as the reference), kept in LDS;
//   * lane c owns step c: cdf(c), bin(c) = first b with !(cdf > cum[b])
//     (running max over the lanes by a wave scan), its in-bin sample goes to
//     slot c + bin(c); the bin boundaries crossed since step c-1 go to slots
//     c + b; z_low comes from lane c-1 by shuffle when it lies in the same
//     bin, else it is the bin's entry depth;
//   * the first lane whose bin reaches the number of valid bins ends the ray
//     ("done" in the reference);
//   * the reference's trailing loop over the remaining bins, with its quirks
//     (`~done` always true, `pts_idx[curr_bin]` read without the ray offset,
//     the `num_rays > H + curr_bin` guard), runs on lane 0.
// Writes beyond max_steps are dropped and pts_idx reads one past the buffer
// (the reference performs both) return -1.
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
  float* cum = cdf_lds + wave * max_hits;
  const int64_t boff = (int64_t)bi * num_rays * max_hits;
  const int* PI = pts_idx + boff;
  const float* MN = min_depth + boff;
  const float* MX = max_depth + boff;
  const float* PR = probs + boff;
  const float* UN = uniform_noise + (int64_t)bi * num_rays * max_steps;
  int* SI = sampled_idx + (int64_t)bi * num_rays * max_steps;
  float* SD = sampled_depth + (int64_t)bi * num_rays * max_steps;
  float* SS = sampled_dists + (int64_t)bi * num_rays * max_steps;
  const int H = j * max_hits, K = j * max_steps;
  auto pi = [&](int i) -> int {
    return (boff + i < pi_total) ? PI[i] : -1;
  };
  auto emit = [&](int slot, int id, float zhi, float zlo) {
    if (slot < max_steps) {
      SI[K + slot] = id;
      SS[K + slot] = zhi - zlo;
      SD[K + slot] = (zhi + zlo) * 0.5f;
    }
  };
  // valid bins: bin 0 always, then up to the first -1
  int nbv = max_hits;
  for (int b0 = 0; b0 < max_hits; b0 += 64) {
    const int b = b0 + lane;
    const bool stop = b >= 1 && b < max_hits && pi(H + b) == -1;
    const uint64_t m = __ballot(stop);
    if (m) {
      nbv = b0 + __builtin_ctzll(m);
      break;
    }
  }
  {  // serial prefix sum, one writer
    float acc = 0.f;
    for (int b = 0; b < nbv; ++b) {
      acc = acc + PR[H + b];
      if (lane == 0) cum[b] = acc;
    }
  }
  wave_lds_sync();
  const float st = steps[(int64_t)bi * num_rays + j];
  float step_size = (float)(1.0 / (double)st);
  if (fixed_step_size > 0.0) step_size = fixed_step_size;
  const int total = (int)ceil((double)st);
  int carry_bin = 0;         // bin of the last step of the previous round
  float carry_z = MN[H];     // its z
  int run_bin = 0;
  bool done = false;
  int tail_bin = 0, tail_s = 0;
  float tail_zlow = carry_z, tail_max = MX[H];
  for (int base = 0; base < total; base += 64) {
    const int c = base + lane;
    const bool act = c < total;
    int f = 0;
    float cdf = 0.f;
    if (act) {
      cdf = ((float)c + UN[K + c]) * step_size;
      while (f < nbv && cdf > cum[f]) ++f;
    }
    int bn = f;  // inclusive running max over the lanes, then the carry
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(bn, o);
      if (lane >= o) bn = bn > u ? bn : u;
    }
    bn = bn > run_bin ? bn : run_bin;
    const uint64_t dmask = __ballot(act && bn >= nbv);
    const int first_done = dmask ? __builtin_ctzll(dmask) : 64;
    const bool mine = act && lane <= first_done;
    const bool is_done = lane == first_done;
    float z = 0.f;
    if (mine && !is_done) {
      const float cmin = bn > 0 ? cum[bn - 1] : 0.f;
      const float u = (cdf - cmin) / (cum[bn] - cmin);
      const float lo = MN[H + bn];
      z = fmaf(u, MX[H + bn] - lo, lo);  // nvcc contracts this expression
    }
    int pb = __shfl_up(bn, 1);
    float pz = __shfl_up(z, 1);
    if (lane == 0) {
      pb = carry_bin;
      pz = carry_z;
    }
    if (mine) {
      const int hi = bn < nbv ? bn : nbv;
      for (int b = pb; b < hi; ++b)  // boundaries crossed since step c-1
        emit(c + b, pi(H + b), MX[H + b], b == pb ? pz : MN[H + b]);
      if (!is_done)
        emit(c + bn, pi(H + bn), z, bn == pb ? pz : MN[H + bn]);
    }
    if (first_done < 64) {
      // state after the reference's `done` break, from the done lane
      const float zl = (nbv - 1 == pb) ? pz : MN[H + nbv - 1];
      tail_zlow = __shfl(zl, first_done);
      tail_bin = nbv;
      tail_s = base + first_done + nbv;
      tail_max = MX[H + nbv - 1];
      done = true;
      break;
    }
    const int last = (total - base < 64 ? total - base : 64) - 1;
    carry_bin = __shfl(bn, last);
    carry_z = __shfl(z, last);
    run_bin = carry_bin;
  }
  if (!done) {
    tail_bin = carry_bin;
    tail_s = total + carry_bin;
    tail_zlow = carry_z;
    tail_max = MX[H + carry_bin];
  }
  if (lane != 0) return;
  while (tail_zlow < tail_max && num_rays > H + tail_bin) {
    emit(tail_s, pi(H + tail_bin), tail_max, tail_zlow);
    ++tail_bin;
    ++tail_s;
    if (tail_bin >= max_hits || pi(tail_bin) == -1) break;
    tail_max = MX[H + tail_bin];
    tail_zlow = MN[H + tail_bin];
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
