// Vox-Fusion: the ray side of one optimisation iteration on gfx950 — what the
// reference does between `svo_intersect` and the decoder, and between the
// decoder and the loss, as ~100 small PyTorch launches and three host syncs
// per iteration:
//   ray_intersect      voxel_helpers_voxfusion.py:647-687  (mask, sort by entry
//                      depth, max_distance cut, trim to the largest hit count)
//   ray_sample         :690-714, InverseCDFRaySampling :399-481 (probs, steps,
//                      [200, R, P] regrouping, sampler, trim to the longest row)
//   render_rays        slam/models/sparse_voxel.py:160-275 (hit-ray compaction,
//                      sample -> point compaction, padded sdf = 1 / colour = 0,
//                      sdf2weights :277-304, colour / depth sums)
//   get_loss_dict      :103-143 + get_sdf_loss / get_masks
//                      (slam/model_components/utils.py:100-186)
// Everything has STATIC capacity here (n_max hits, S_cap samples per ray, P_cap
// points) and the data-dependent sizes the reference trims its tensors to —
// which its results depend on: the [G,R,P] regrouping quirks of the sampler and
// the means over the padded [hit rays, longest row] array — live in a small
// device record (`meta`), so that an iteration is a fixed launch sequence
// without a host sync and can be captured in a hipGraph.  One wave owns one
// ray in every kernel; samples sit on lanes.
//
//   vox_hit_sort   rank sort of <= 64 hits on the lanes of a wave
//   vox_ray_scan   hit-ray ranks (the sampler's [G,R] regrouping is over the
//                  COMPACTED hit rays)
//   vox_sample     svo_sample.h (bit-exact with the reference kernel)
//   vox_point_scan / vox_compact   sample -> point offsets, xyz = o + d z,
//                  free-space / band counts of the loss weights
//   vox_render_fwd compositing + the four loss sums
//   vox_loss_finalize
//   vox_render_bwd analytic gradient of the loss w.r.t. per-point sdf / rgb
//   vox_ray_grad   per-point position gradients -> rays_o / rays_d gradients
// Reference behaviour restated, never copied.  Parity: tests/test_vox_rays_hip.py
// (against the modular path, itself pinned to the reference-made golden).
#include "common.h"
#include "svo_intersect.h"
#include "svo_sample.h"  // also switches fp contraction off for this file

namespace xrd {
namespace {

enum Meta {
  M_NHITCOL = 0,   // largest hit count of a ray   (reference: n_hit / P)
  M_NHITRAYS = 1,  // rays with at least one hit   (N of the sampler)
  M_MAXSTEPS = 2,  // ceil(max steps) + P          (row length of the sampler)
  M_SMAX = 3,      // longest sample row           (max_len)
  M_NPTS = 4,      // valid samples (clamped to the capacity)
  M_OVERFLOW = 5,  // bit 0: S_cap, bit 1: P_cap, bit 2: non-contiguous row,
                   // bit 3: traversal stack
  M_NFRONT = 6,    // free-space samples
  M_NMID = 7,      // samples inside the truncation band
  M_NVALID = 8,    // hit rays with a usable sensor depth
  M_MAXCEIL = 9,   // max over hit rays of ceil(steps)
  M_STACKOVF = 10, // octree traversal stack overflow (xrd_svo_intersect)
  // sticky part: NOT cleared by a launch — the record of every launch is
  // folded in when the next launch resets the slots above, so a caller that
  // reads once per optimize_update sees an overflow of ANY iteration (the
  // caller clears these three after reading)
  M_STICKY_OVF = 11,   // OR of M_OVERFLOW (| 8 for M_STACKOVF)
  M_STICKY_ROW = 12,   // max of M_MAXSTEPS
  M_STICKY_PTS = 13,   // max number of valid samples a launch wanted
  M_WANT_PTS = 14      // valid samples before clamping to the capacity
};

constexpr int kWaves = 8;          // rays per block
constexpr float kPadDepth = 10.f;  // MAX_DEPTH of the padded samples
constexpr int kGroups = 200;       // G of InverseCDFRaySampling.forward

// ---------------------------------------------------------------- ray pipeline
// intersect -> hit sort -> hit-ray ranks -> sampling -> point offsets ->
// compaction as THREE launches (round 5; six launches + the reset before:
// ~11 us of launch / drain each for a few microseconds of work, 77 us x 45
// iterations a frame).  The steps need three batch-wide results — the number
// of hit rays and the two maxima the sampler's [200, R, P] regrouping is
// built from, the compacted list of hit rays (a ray's group reads OTHER rays'
// hit rows), the point offsets.  A block owns a CONTIGUOUS range of rays and
// leaves its partial counts in meta[M_PART]; the NEXT launch's blocks each sum
// the partials they need (a prefix over <= 1024 block counts) and rebuild the
// compacted hit-ray list in LDS from the hit flags: the two single-block scan
// launches and the reset launch are gone, and nothing is an atomic on one
// address.
// (Tried first: ONE launch whose resident blocks meet at three grid barriers
// — 121 us: an agent-scope release / acquire per block and barrier writes
// back / invalidates the XCDs' L2s, and bulk data crosses blocks at every
// barrier.)  The per-ray arithmetic is the code of the former kernels:
// bit-identical results (tests/test_vox_rays_hip.py).
constexpr int kPipeBlocks = 1024;
enum MetaExt {
  M_PART = 16,                         // [kPipeBlocks][8] per-block partials
  M_LEN = M_PART + 8 * kPipeBlocks
};
enum Part { PT_HITS = 0, PT_MAXCOL, PT_MAXCEIL, PT_STACK, PT_CNT, PT_SMAX,
            PT_OVF, PT_WANT };

struct PipeArgs {
  int n_rays, n_max, s_cap, n_nodes;
  int64_t p_cap;
  float voxel_size, max_distance, inv_step, trunc, max_depth;
  const float *centres, *rays_o, *rays_d, *target_d, *noise;
  const int* children;
  const uint8_t* ray_keep;
  int *hit_idx, *hit, *rank, *hit_rays, *s_idx, *cnt, *offs, *vox, *meta;
  float *hit_min, *hit_max, *probs, *steps, *s_depth, *xyz;
  double* loss_acc;
};

// sum / max of the blocks' partials (slot `what`) over blocks [0, upto) and
// over all blocks: every wave computes them redundantly (<= 768 values)
__device__ __forceinline__ void part_sums(const int* __restrict__ part,
                                          int what, int upto, int lane,
                                          int& before, int& total) {
  int b = 0, s = 0;
  for (int i = lane; i < (int)gridDim.x; i += 64) {
    const int v = part[i * 8 + what];
    s += v;
    if (i < upto) b += v;
  }
  before = __reduce_add_sync(0xffffffffffffffffull, b);
  total = __reduce_add_sync(0xffffffffffffffffull, s);
}
__device__ __forceinline__ int part_max(const int* __restrict__ part, int what,
                                        int lane) {
  int m = 0;
  for (int i = lane; i < (int)gridDim.x; i += 64) {
    const int v = part[i * 8 + what];
    m = v > m ? v : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int u = __shfl_xor(m, o);
    m = u > m ? u : m;
  }
  return m;
}
__device__ __forceinline__ int part_or(const int* __restrict__ part, int what,
                                       int lane) {
  int m = 0;
  for (int i = lane; i < (int)gridDim.x; i += 64) m |= part[i * 8 + what];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o);
  return m;
}

// ---- 1: intersection (or the caller's hits) and the sort by entry depth ------
__global__ __launch_bounds__(kWaves * 64) void vox_hits_kernel(const PipeArgs A) {
  // LDS per wave: the DFS stack of the intersection (4 x kSvoStack words) and
  // the hit row (3 x 64)
  __shared__ int svo_s[kWaves][kSvoLds];
  __shared__ int row_s[kWaves][3 * 64];
  __shared__ int red[8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* const meta = A.meta;
  int* const part = meta + M_PART;
  const int n_rays = A.n_rays, n_max = A.n_max, s_cap = A.s_cap;
  // contiguous rays of this block (multiples of the 8 waves); the three
  // launches use the same grid
  const int rpb = ((n_rays + (int)gridDim.x - 1) / (int)gridDim.x + kWaves - 1) /
                  kWaves * kWaves;
  const int r_lo = blockIdx.x * rpb;
  const int r_hi = r_lo + rpb < n_rays ? r_lo + rpb : n_rays;
  if (threadIdx.x < 8) red[threadIdx.x] = 0;
  __syncthreads();
  int* s_node = svo_s[wave];
  int* s_side = s_node + kSvoStack;
  float* s_lo = reinterpret_cast<float*>(s_side + kSvoStack);
  float* s_hi = s_lo + kSvoStack;
  int* s_kids = s_node + 4 * kSvoStack;
  int* s_id = row_s[wave];
  float* s_a = reinterpret_cast<float*>(s_id + 64);
  float* s_b = s_a + 64;
  for (int ray = r_lo + wave; ray < r_hi; ray += kWaves) {
    const int64_t row = (int64_t)ray * n_max;
    int id = -1;
    float a = 0.f, b = 0.f;
    if (A.centres != nullptr) {
      const float o[3] = {A.rays_o[ray * 3], A.rays_o[ray * 3 + 1],
                          A.rays_o[ray * 3 + 2]};
      const float d[3] = {A.rays_d[ray * 3], A.rays_d[ray * 3 + 1],
                          A.rays_d[ray * 3 + 2]};
      bool ovf = false;
      int cnt = 0;
      // level by level (svo_intersect.h); the depth-first walk only where
      // that one gives up
      const bool done = svo_intersect_ray_bfs(
          lane, s_node, o, d, A.centres, A.children, A.voxel_size, n_max,
          [&](int n_hit, const int* node, const float* lo, const float* hi) {
            cnt = n_hit;
            if (lane < n_hit) {
              id = node[lane];
              a = lo[lane];
              b = hi[lane];
            }
          });
      if (!done)
        cnt = svo_intersect_ray(
            lane, s_node, s_side, s_lo, s_hi, s_kids, o, d, A.centres,
            A.children, A.voxel_size, n_max, ovf,
            [&](int slot, int node, float lo, float hi) {
              if (lane == slot) {
                id = node;
                a = lo;
                b = hi;
              }
            });
      if (ovf && lane == 0) atomicOr(&red[PT_STACK], 1);
      if (lane >= cnt) id = -1;
    } else if (lane < n_max) {
      id = A.hit_idx[row + lane];
      a = A.hit_min[row + lane];
      b = A.hit_max[row + lane];
    }
    const bool in = lane < n_max;
    if (id == -1) a = b = A.max_distance;
    // stable rank by entry depth
    int rank = 0;
    for (int j = 0; j < n_max; ++j) {
      const float aj = __shfl(a, j);
      rank += (aj < a || (aj == a && j < lane)) ? 1 : 0;
    }
    if (in) {
      s_id[rank] = id;
      s_a[rank] = a;
      s_b[rank] = b;
    }
    wave_lds_sync();
    if (in) {
      id = s_id[lane];
      a = s_a[lane];
      b = s_b[lane];
      if (a > A.max_distance) id = -1;
      if (id == -1) a = b = A.max_distance;
    }
    wave_lds_sync();
    const int count = __popcll(__ballot(in && id != -1));
    const float len = (in && id != -1) ? b - a : 0.f;
    if (in) s_a[lane] = len;
    wave_lds_sync();
    float sum = 0.f;
    for (int j = 0; j < n_max; ++j) sum = sum + s_a[j];
    if (in) {
      A.hit_idx[row + lane] = id;
      A.hit_min[row + lane] = a;
      A.hit_max[row + lane] = b;
      A.probs[row + lane] = len / sum;
    }
    wave_lds_sync();
    if (lane == 0) {
      // torch divides by a python scalar as a multiplication with its reciprocal
      const float st = sum * A.inv_step;
      A.steps[ray] = st;
      A.hit[ray] = count > 0 ? 1 : 0;
      if (count > 0) {
        atomicAdd(&red[PT_HITS], 1);
        atomicMax(&red[PT_MAXCOL], count);
        atomicMax(&red[PT_MAXCEIL], (int)ceilf(st));
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 4) part[blockIdx.x * 8 + threadIdx.x] = red[threadIdx.x];
}

// ---- 2: the batch's size record, hit-ray ranks, sampling ------------------------
__global__ __launch_bounds__(kWaves * 64) void vox_sample_kernel(const PipeArgs A) {
  // dynamic LDS: the compacted hit-ray list [n_rays], then per wave the
  // cumulative probabilities (64) and the sample row (2 x s_cap)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int red[8];
  __shared__ int wsum[kWaves];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* const meta = A.meta;
  int* const part = meta + M_PART;
  const int n_rays = A.n_rays, n_max = A.n_max, s_cap = A.s_cap;
  // contiguous rays of this block (multiples of the 8 waves); the three
  // launches use the same grid
  const int rpb = ((n_rays + (int)gridDim.x - 1) / (int)gridDim.x + kWaves - 1) /
                  kWaves * kWaves;
  const int r_lo = blockIdx.x * rpb;
  const int r_hi = r_lo + rpb < n_rays ? r_lo + rpb : n_rays;
  if (threadIdx.x < 8) red[threadIdx.x] = 0;
  __syncthreads();
  int* hit_rays_s = reinterpret_cast<int*>(smem_raw);
  int* wbase = hit_rays_s + (n_rays + 3) / 4 * 4 + wave * (64 + 2 * s_cap);
  float* cum_s = reinterpret_cast<float*>(wbase);
  int* LI = wbase + 64;
  float* LD = reinterpret_cast<float*>(LI + s_cap);
  // ---- 2: the batch's size record, hit-ray ranks ---------------------------
  int hits_before, n_hit_rays;
  part_sums(part, PT_HITS, blockIdx.x, lane, hits_before, n_hit_rays);
  (void)hits_before;
  const int P = part_max(part, PT_MAXCOL, lane);
  const int max_ceil = part_max(part, PT_MAXCEIL, lane);
  const int stack_ovf = part_or(part, PT_STACK, lane);
  const int max_steps_all = max_ceil + P;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // the previous launch's record -> sticky slots, then this launch's
    meta[M_STICKY_OVF] |= meta[M_OVERFLOW] | (meta[M_STACKOVF] ? 8 : 0);
    if (meta[M_MAXSTEPS] > meta[M_STICKY_ROW])
      meta[M_STICKY_ROW] = meta[M_MAXSTEPS];
    if (meta[M_WANT_PTS] > meta[M_STICKY_PTS])
      meta[M_STICKY_PTS] = meta[M_WANT_PTS];
    for (int i = 0; i < M_STICKY_OVF; ++i) meta[i] = 0;
    meta[M_WANT_PTS] = 0;
    meta[15] = 0;
    meta[M_NHITCOL] = P;
    meta[M_NHITRAYS] = n_hit_rays;
    meta[M_MAXCEIL] = max_ceil;
    meta[M_MAXSTEPS] = max_steps_all;
    meta[M_STACKOVF] = stack_ovf;
    for (int i = 0; i < 4; ++i) A.loss_acc[i] = 0.0;
  }
  // the compacted list of hit rays, rebuilt by EVERY block in LDS (a ray's
  // group reads other rays' rows): thread t scans a contiguous slice of the
  // hit flags, the block scans the slices' counts.  Ranks and the global list
  // are written by the block that owns the ray.
  {
    const int per = (n_rays + (int)blockDim.x - 1) / (int)blockDim.x;
    const int i0 = threadIdx.x * per < n_rays ? threadIdx.x * per : n_rays;
    const int i1 = i0 + per < n_rays ? i0 + per : n_rays;
    int c = 0;
    for (int i = i0; i < i1; ++i) c += A.hit[i] != 0;
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(inc, o);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int at = inc - c;
    for (int w = 0; w < wave; ++w) at += wsum[w];
    for (int i = i0; i < i1; ++i) {
      const bool h = A.hit[i] != 0;
      if (i >= r_lo && i < r_hi) {
        A.rank[i] = at;
        if (h) A.hit_rays[at] = i;
      }
      if (h) hit_rays_s[at++] = i;
    }
  }
  __syncthreads();
  // ---- 3: sampling -------------------------------------------------------------
  for (int ray = r_lo + wave; ray < r_hi; ray += kWaves) {
    const bool work = A.hit[ray] != 0;
    if (!work) {
      if (lane == 0) A.cnt[ray] = 0;
      continue;
    }
    const int R = (n_hit_rays + kGroups - 1) / kGroups;  // rays per group
    const int r = A.rank[ray];
    const int g = r / R, j = r - g * R;
    const int H = j * P;
    int max_steps = max_steps_all;
    if (max_steps > s_cap) {
      if (lane == 0) atomicOr(&red[PT_OVF], 1);
      max_steps = s_cap;
    }
    for (int s = lane; s < max_steps; s += 64) {
      LI[s] = -1;
      LD[s] = kPadDepth;
    }
    wave_lds_sync();
    const int* own = A.hit_idx + (int64_t)ray * n_max;
    const int64_t goff = (int64_t)g * R * P, total = (int64_t)kGroups * R * P;
    const float* UN = A.noise ? A.noise + (int64_t)ray * s_cap : nullptr;
    inverse_cdf_ray(
        lane, cum_s, P, R, H, A.hit_min + (int64_t)ray * n_max,
        A.hit_max + (int64_t)ray * n_max, A.probs + (int64_t)ray * n_max,
        A.steps[ray], -1.f,
        [&](int i) -> int {
          // flat index i of the group's [R, P] hit array
          if (i >= H && i < H + P) return own[i - H];
          if (goff + i >= total) return -1;
          const int rr = i / P, col = i - rr * P;
          // rows past the last hit ray are copies of the first one
          const int q = g * R + rr;
          const int src = hit_rays_s[q < n_hit_rays ? q : 0];
          return A.hit_idx[(int64_t)src * n_max + col];
        },
        [&](int c) -> float {
          if (UN == nullptr || c >= s_cap) return 0.5f;
          return fminf(fmaxf(UN[c], 0.001f), 0.999f);
        },
        [&](int slot, int id, float zhi, float zlo) {
          if (slot < max_steps) {
            LI[slot] = id;
            LD[slot] = (zhi + zlo) * 0.5f;
          }
        });
    wave_lds_sync();
    // valid samples of the row (they form a prefix; anything else is
    // reported); padded samples carry depth = MAX_DEPTH (ray_sample :709-711)
    int* SI = A.s_idx + (int64_t)ray * s_cap;
    float* SD = A.s_depth + (int64_t)ray * s_cap;
    int count = 0, last = -1;
    for (int s0 = 0; s0 < max_steps; s0 += 64) {
      const int s = s0 + lane;
      const int id = s < max_steps ? LI[s] : -1;
      const uint64_t m = __ballot(id != -1);
      count += __popcll(m);
      if (m) last = s0 + 63 - __builtin_clzll(m);
      if (s < max_steps) {
        SI[s] = id;
        SD[s] = id == -1 ? kPadDepth : LD[s];
      }
    }
    wave_lds_sync();
    if (lane == 0) {
      A.cnt[ray] = count;
      atomicMax(&red[PT_SMAX], count);
      if (last + 1 != count) atomicOr(&red[PT_OVF], 4);
      // (sharded mapping: rays of other ranks are sampled — they shape the
      // regrouping and the size record — but get no points)
      if (A.ray_keep == nullptr || A.ray_keep[ray]) atomicAdd(&red[PT_CNT], count);
    }
  }
  __syncthreads();
  if (threadIdx.x >= 4 && threadIdx.x < 8)
    part[blockIdx.x * 8 + threadIdx.x] = red[threadIdx.x];
}

// ---- 3: point offsets, compaction, the loss weights' sample counts -----------
__global__ __launch_bounds__(kWaves * 64) void vox_compact_kernel(const PipeArgs A) {
  __shared__ int red[8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* const meta = A.meta;
  int* const part = meta + M_PART;
  const int n_rays = A.n_rays, n_max = A.n_max, s_cap = A.s_cap;
  // contiguous rays of this block (multiples of the 8 waves); the three
  // launches use the same grid
  const int rpb = ((n_rays + (int)gridDim.x - 1) / (int)gridDim.x + kWaves - 1) /
                  kWaves * kWaves;
  const int r_lo = blockIdx.x * rpb;
  const int r_hi = r_lo + rpb < n_rays ? r_lo + rpb : n_rays;
  if (threadIdx.x < 8) red[threadIdx.x] = 0;
  __syncthreads();
  // ---- 4: point offsets, compaction -------------------------------------------
  int pts_before, pts_total;
  part_sums(part, PT_CNT, blockIdx.x, lane, pts_before, pts_total);
  const int s_longest = part_max(part, PT_SMAX, lane);
  const int ovf_bits = part_or(part, PT_OVF, lane);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // (n_rays == 0: the argument check lets a caller pass no buffers)
    if (A.offs != nullptr) A.offs[n_rays] = pts_total;
    meta[M_SMAX] = s_longest;
    meta[M_OVERFLOW] = ovf_bits | ((int64_t)pts_total > A.p_cap ? 2 : 0);
    meta[M_NPTS] = (int64_t)pts_total > A.p_cap ? (int)A.p_cap : pts_total;
    meta[M_WANT_PTS] = pts_total;
  }
  if (wave == 0) {   // exclusive offsets of this block's rays, in ray order
    int base = pts_before;
    for (int r0 = r_lo; r0 < r_hi; r0 += 64) {
      const int ray = r0 + lane;
      int v = 0;
      if (ray < r_hi && (A.ray_keep == nullptr || A.ray_keep[ray]))
        v = A.cnt[ray];
      int inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
      }
      if (ray < r_hi) A.offs[ray] = base + inc - v;
      base += __shfl(inc, 63);
    }
  }
  if (threadIdx.x < 8) red[threadIdx.x] = 0;
  __syncthreads();   // (offs of this block's rays are written; LDS counters)
  __threadfence_block();
  for (int ray = r_lo + wave; ray < r_hi; ray += kWaves) {
    if (!A.hit[ray]) continue;
    // a ray of another rank's shard counts towards the batch-global sample
    // counts below (they are the loss normalisers) but leaves no points and
    // is a ray without a hit for everything after this kernel
    const bool keep = A.ray_keep == nullptr || A.ray_keep[ray] != 0;
    const int s_max = s_longest < s_cap ? s_longest : s_cap;
    const int n = A.cnt[ray];
    const int64_t p0 = A.offs[ray];
    const float o[3] = {A.rays_o[ray * 3], A.rays_o[ray * 3 + 1],
                        A.rays_o[ray * 3 + 2]};
    const float d[3] = {A.rays_d[ray * 3], A.rays_d[ray * 3 + 1],
                        A.rays_d[ray * 3 + 2]};
    const float td = A.target_d[ray];
    const float lo = td - A.trunc, hi = td + A.trunc;
    int n_front = 0, n_mid = 0;
    for (int s0 = 0; s0 < s_max; s0 += 64) {
      const int s = s0 + lane;
      const bool in = s < s_max;
      const bool valid = s < n;
      const float z = valid ? A.s_depth[(int64_t)ray * s_cap + s] : kPadDepth;
      if (keep && valid && p0 + s < A.p_cap) {
        const int64_t p = p0 + s;
#pragma unroll
        for (int k = 0; k < 3; ++k) A.xyz[p * 3 + k] = o[k] + d[k] * z;
        A.vox[p] = A.s_idx[(int64_t)ray * s_cap + s];
      }
      const bool front = in && z < lo;
      const bool back = in && z > hi;
      n_front += __popcll(__ballot(front));
      n_mid += __popcll(__ballot(in && !front && !back && td > 0.f));
    }
    if (lane == 0) {
      if (n_front) atomicAdd(&red[0], n_front);
      if (n_mid) atomicAdd(&red[1], n_mid);
      if (td > 0.01f && td < A.max_depth) atomicAdd(&red[2], 1);
      if (!keep) {
        A.cnt[ray] = 0;
        A.hit[ray] = 0;
      }
    }
  }
  __syncthreads();
  // (block 0 zeroed the counters before barrier 1: these adds come after it)
  if (threadIdx.x < 3 && red[threadIdx.x] != 0)
    atomicAdd(meta + M_NFRONT + threadIdx.x, red[threadIdx.x]);
}

// ---------------------------------------------------------------- compositing
struct RayView {
  int n, s_max;
  int64_t p0, p_cap;
  const float* z;     // the ray's sample depths
  const float* sdf;   // per point
  __device__ __forceinline__ bool live(int s) const {
    return s < n && p0 + s < p_cap;
  }
  __device__ __forceinline__ float sdf_at(int s) const {
    return live(s) ? sdf[p0 + s] : 1.f;   // padded samples: free space
  }
  __device__ __forceinline__ float z_at(int s) const {
    return s < n ? z[s] : kPadDepth;
  }
};

__device__ __forceinline__ float sigmoidf(float x) {
  return 1.f / (1.f + expf(-x));
}

// depth of the first sign change of the (padded) sdf row, sdf2weights :284-291
__device__ __forceinline__ float first_crossing_depth(const RayView& v,
                                                      int lane) {
  for (int s0 = 0; s0 + 1 < v.s_max; s0 += 64) {
    const int s = s0 + lane;
    const bool cross =
        s + 1 < v.s_max && v.sdf_at(s + 1) * v.sdf_at(s) < 0.f;
    const uint64_t m = __ballot(cross);
    if (m) return v.z_at(s0 + __builtin_ctzll(m));
  }
  return v.z_at(0);
}

struct LossScale {  // written by vox_loss_finalize
  float c_rgb, c_depth, c_fs, c_sdf;
};

// one wave: weights, colour, depth of one ray.  `u` of lane's samples are
// recomputed by the callers from (sdf, z, z_min).
__device__ __forceinline__ void composite_ray(const RayView& v, int lane,
                                              const float* __restrict__ rgb_pt,
                                              float inv_tr, float tr,
                                              float& z_min, float& usum,
                                              float (&rgb)[3], float& depth) {
  z_min = first_crossing_depth(v, lane);
  const float z_cut = z_min + tr;
  float U = 0.f;
  for (int s0 = 0; s0 < v.n; s0 += 64) {
    const int s = s0 + lane;
    if (v.live(s)) {
      const float sd = v.sdf[v.p0 + s];
      const float b = sigmoidf(sd * inv_tr) * sigmoidf(-sd * inv_tr);
      U += v.z[s] < z_cut ? b : 0.f;
    }
  }
  U = wave_sum(U);
  usum = U;
  const float den = U + 1e-8f;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < v.n; s0 += 64) {
    const int s = s0 + lane;
    if (v.live(s)) {
      const float sd = v.sdf[v.p0 + s];
      const float b = sigmoidf(sd * inv_tr) * sigmoidf(-sd * inv_tr);
      const float w = (v.z[s] < z_cut ? b : 0.f) / den;
      const float* c = rgb_pt + (v.p0 + s) * 3;
      acc[0] += w * c[0];
      acc[1] += w * c[1];
      acc[2] += w * c[2];
      acc[3] += w * v.z[s];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) rgb[k] = wave_sum(acc[k]);
  depth = wave_sum(acc[3]);
}

struct RenderArgs {
  int n_rays, s_cap;
  int64_t p_cap;
  float trunc, inv_trunc, max_depth;
};

__global__ __launch_bounds__(kWaves * 64) void vox_render_fwd_kernel(
    RenderArgs a, const int* __restrict__ hit, const int* __restrict__ cnt,
    const int* __restrict__ offs, const float* __restrict__ s_depth,
    const float* __restrict__ sdf_pt, const float* __restrict__ rgb_pt,
    const float* __restrict__ target_d, const float* __restrict__ target_rgb,
    const int* __restrict__ meta, float* __restrict__ depth_out,
    float* __restrict__ rgb_out, float* __restrict__ zmin_out,
    float* __restrict__ weights_out, double* __restrict__ acc) {
  __shared__ float part[kWaves][4];   // this block's loss sums, per ray
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * kWaves + wave;
  if (lane < 4) part[wave][lane] = 0.f;
  const bool live = ray < a.n_rays;
  if (live && !hit[ray]) {
    if (lane == 0) {
      depth_out[ray] = 0.f;
      rgb_out[ray * 3] = rgb_out[ray * 3 + 1] = rgb_out[ray * 3 + 2] = 0.f;
      if (zmin_out) zmin_out[ray] = 0.f;
    }
    if (weights_out)
      for (int s = lane; s < a.s_cap; s += 64)
        weights_out[(int64_t)ray * a.s_cap + s] = 0.f;
  }
  if (live && hit[ray]) {
    RayView v;
    v.n = cnt[ray];
    v.s_max = meta[M_SMAX] < a.s_cap ? meta[M_SMAX] : a.s_cap;
    v.p0 = offs[ray];
    v.p_cap = a.p_cap;
    v.z = s_depth + (int64_t)ray * a.s_cap;
    v.sdf = sdf_pt;
    float z_min, U, rgb[3], depth;
    composite_ray(v, lane, rgb_pt, a.inv_trunc, a.trunc, z_min, U, rgb, depth);
    if (lane == 0) {
      depth_out[ray] = depth;
      rgb_out[ray * 3] = rgb[0];
      rgb_out[ray * 3 + 1] = rgb[1];
      rgb_out[ray * 3 + 2] = rgb[2];
      if (zmin_out) zmin_out[ray] = z_min;
    }
    if (weights_out) {
      const float den = U + 1e-8f, z_cut = z_min + a.trunc;
      for (int s = lane; s < a.s_cap; s += 64) {
        float w = 0.f;
        if (v.live(s)) {
          const float sd = v.sdf[v.p0 + s];
          const float b =
              sigmoidf(sd * a.inv_trunc) * sigmoidf(-sd * a.inv_trunc);
          w = (v.z[s] < z_cut ? b : 0.f) / den;
        }
        weights_out[(int64_t)ray * a.s_cap + s] = w;
      }
    }
    if (acc != nullptr) {
      // the four loss sums of this ray (get_loss_dict :103-143)
      const float td = target_d[ray];
      const float wv = (td > 0.01f && td < a.max_depth) ? 1.f : 0.f;
      const float lo = td - a.trunc, hi = td + a.trunc;
      float fs = 0.f, sd2 = 0.f;
      for (int s0 = 0; s0 < v.s_max; s0 += 64) {
        const int s = s0 + lane;
        if (s >= v.s_max) continue;
        const float z = v.z_at(s), sd = v.sdf_at(s);
        const bool front = z < lo, back = z > hi;
        if (front) fs += (sd - 1.f) * (sd - 1.f);
        if (!front && !back && td > 0.f) {
          const float e = (z + sd * a.trunc) - td;
          sd2 += e * e;
        }
      }
      fs = wave_sum(fs);
      sd2 = wave_sum(sd2);
      if (lane == 0) {
        float l_rgb = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          l_rgb += fabsf(rgb[k] * wv - target_rgb[ray * 3 + k] * wv);
        part[wave][0] = l_rgb;
        part[wave][1] = wv != 0.f ? fabsf(depth - td) : 0.f;
        part[wave][2] = fs;
        part[wave][3] = sd2;
      }
    }
  }
  if (acc == nullptr) return;   // uniform over the grid
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) t += (double)part[w][threadIdx.x];
    if (t != 0.0) atomicAdd(acc + threadIdx.x, t);
  }
}

// loss[0..3] = rgb, depth, sdf, fs (weighted), loss[4] = their sum in the
// order the algorithm adds them; scale = the factors of the backward
__global__ void vox_loss_finalize_kernel(const int* __restrict__ meta,
                                         int s_cap,
                                         const double* __restrict__ acc,
                                         float w_rgb, float w_depth,
                                         float w_sdf, float w_fs,
                                         float* __restrict__ loss,
                                         LossScale* __restrict__ scale) {
  if (threadIdx.x != 0) return;
  const int n_hit = meta[M_NHITRAYS];
  const int s_max = meta[M_SMAX] < s_cap ? meta[M_SMAX] : s_cap;
  LossScale sc = {0.f, 0.f, 0.f, 0.f};
  if (n_hit > 0 && s_max > 0) {
    const float n_fs = (float)meta[M_NFRONT], n_sdf = (float)meta[M_NMID];
    const float tot = n_fs + n_sdf;
    const float fs_w = 1.f - n_fs / tot, sdf_w = 1.f - n_sdf / tot;
    const float cells = (float)n_hit * (float)s_max;
    sc.c_rgb = w_rgb / (3.f * (float)n_hit);
    sc.c_depth = w_depth / (float)meta[M_NVALID];
    sc.c_fs = w_fs * fs_w / cells;
    sc.c_sdf = w_sdf * sdf_w / cells;
  }
  loss[0] = (float)acc[0] * sc.c_rgb;
  loss[1] = meta[M_NVALID] > 0 ? (float)acc[1] * sc.c_depth : 0.f;
  loss[2] = (float)acc[3] * sc.c_sdf;
  loss[3] = (float)acc[2] * sc.c_fs;
  loss[4] = ((loss[0] + loss[1]) + loss[2]) + loss[3];
  if (!(meta[M_NVALID] > 0)) sc.c_depth = 0.f;
  *scale = sc;
}

// d loss / d (per-point sdf, rgb), scaled by the upstream gradient *g_up
__global__ __launch_bounds__(kWaves * 64) void vox_render_bwd_kernel(
    RenderArgs a, const int* __restrict__ hit, const int* __restrict__ cnt,
    const int* __restrict__ offs, const float* __restrict__ s_depth,
    const float* __restrict__ sdf_pt, const float* __restrict__ rgb_pt,
    const float* __restrict__ target_d, const float* __restrict__ target_rgb,
    const int* __restrict__ meta, const LossScale* __restrict__ scale,
    const float* __restrict__ g_up, float* __restrict__ g_sdf,
    float* __restrict__ g_rgb) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * kWaves + wave;
  if (ray >= a.n_rays || !hit[ray]) return;
  RayView v;
  v.n = cnt[ray];
  v.s_max = meta[M_SMAX] < a.s_cap ? meta[M_SMAX] : a.s_cap;
  v.p0 = offs[ray];
  v.p_cap = a.p_cap;
  v.z = s_depth + (int64_t)ray * a.s_cap;
  v.sdf = sdf_pt;
  float z_min, U, rgb[3], depth;
  composite_ray(v, lane, rgb_pt, a.inv_trunc, a.trunc, z_min, U, rgb, depth);
  const LossScale sc = *scale;
  const float up = g_up ? *g_up : 1.f;
  const float td = target_d[ray];
  const float wv = (td > 0.01f && td < a.max_depth) ? 1.f : 0.f;
  auto sgn = [](float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); };
  float gr[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    gr[k] = up * sc.c_rgb * wv *
            sgn(rgb[k] * wv - target_rgb[ray * 3 + k] * wv);
  const float gd = wv != 0.f ? up * sc.c_depth * sgn(depth - td) : 0.f;
  const float den = U + 1e-8f, z_cut = z_min + a.trunc;
  const float lo = td - a.trunc, hi = td + a.trunc;
  // sum_k a_k w_k with a_k = d loss / d w_k
  float aw = 0.f;
  for (int s0 = 0; s0 < v.n; s0 += 64) {
    const int s = s0 + lane;
    if (v.live(s)) {
      const float sd = v.sdf[v.p0 + s];
      const float b = sigmoidf(sd * a.inv_trunc) * sigmoidf(-sd * a.inv_trunc);
      const float w = (v.z[s] < z_cut ? b : 0.f) / den;
      const float* c = rgb_pt + (v.p0 + s) * 3;
      aw += w * (gr[0] * c[0] + gr[1] * c[1] + gr[2] * c[2] + gd * v.z[s]);
    }
  }
  aw = wave_sum(aw);
  for (int s0 = 0; s0 < v.n; s0 += 64) {
    const int s = s0 + lane;
    if (!v.live(s)) continue;
    const int64_t p = v.p0 + s;
    const float sd = v.sdf[p], z = v.z[s];
    const float sp = sigmoidf(sd * a.inv_trunc),
                sm = sigmoidf(-sd * a.inv_trunc);
    const float b = sp * sm;
    const bool m = z < z_cut;
    const float w = (m ? b : 0.f) / den;
    const float* c = rgb_pt + p * 3;
    const float ak = gr[0] * c[0] + gr[1] * c[1] + gr[2] * c[2] + gd * z;
    float gs = m ? (ak - aw) / den * (b * (sm - sp)) * a.inv_trunc : 0.f;
    const bool front = z < lo, back = z > hi;
    if (front) gs += up * sc.c_fs * 2.f * (sd - 1.f);
    if (!front && !back && td > 0.f)
      gs += up * sc.c_sdf * 2.f * a.trunc * ((z + sd * a.trunc) - td);
    g_sdf[p] = gs;
    g_rgb[p * 3] = gr[0] * w;
    g_rgb[p * 3 + 1] = gr[1] * w;
    g_rgb[p * 3 + 2] = gr[2] * w;
  }
}

// g_rays_o = sum_k g_xyz_k, g_rays_d = sum_k z_k g_xyz_k   (xyz = o + d z)
__global__ __launch_bounds__(kWaves * 64) void vox_ray_grad_kernel(
    int n_rays, int s_cap, int64_t p_cap, const int* __restrict__ hit,
    const int* __restrict__ cnt, const int* __restrict__ offs,
    const float* __restrict__ s_depth, const float* __restrict__ g_xyz,
    float* __restrict__ g_o, float* __restrict__ g_d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * kWaves + wave;
  if (ray >= n_rays) return;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (hit[ray]) {
    const int n = cnt[ray];
    const int64_t p0 = offs[ray];
    for (int s = lane; s < n; s += 64) {
      if (p0 + s >= p_cap) break;
      const float z = s_depth[(int64_t)ray * s_cap + s];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float g = g_xyz[(p0 + s) * 3 + k];
        acc[k] += g;
        acc[3 + k] += z * g;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = wave_sum(acc[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      g_o[ray * 3 + k] = acc[k];
      g_d[ray * 3 + k] = acc[3 + k];
    }
  }
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int xrd_vox_meta_len(void) { return M_LEN; }

int xrd_vox_sample_rays(int n_rays, int n_max, int s_cap, int64_t p_cap,
                        int n_nodes, const float* centres,
                        const int32_t* children, float voxel_size,
                        float max_distance, float step_size, float trunc,
                        float max_depth, const float* rays_o,
                        const float* rays_d, const float* target_d,
                        const float* noise, int32_t* hit_idx, float* hit_min,
                        float* hit_max, float* probs, float* steps,
                        int32_t* hit, int32_t* rank, int32_t* hit_rays,
                        int32_t* s_idx, float* s_depth, int32_t* cnt,
                        int32_t* offs, float* xyz, int32_t* vox, int32_t* meta,
                        double* loss_acc, xrd_stream_t stream) {
  return xrd_vox_sample_rays_shard(
      n_rays, n_max, s_cap, p_cap, n_nodes, centres, children, voxel_size,
      max_distance, step_size, trunc, max_depth, rays_o, rays_d, target_d,
      noise, nullptr, hit_idx, hit_min, hit_max, probs, steps, hit, rank,
      hit_rays, s_idx, s_depth, cnt, offs, xyz, vox, meta, loss_acc, stream);
}

int xrd_vox_sample_rays_shard(
    int n_rays, int n_max, int s_cap, int64_t p_cap, int n_nodes,
    const float* centres, const int32_t* children, float voxel_size,
    float max_distance, float step_size, float trunc, float max_depth,
    const float* rays_o, const float* rays_d, const float* target_d,
    const float* noise, const uint8_t* ray_keep, int32_t* hit_idx,
    float* hit_min, float* hit_max, float* probs, float* steps, int32_t* hit,
    int32_t* rank, int32_t* hit_rays, int32_t* s_idx, float* s_depth,
    int32_t* cnt, int32_t* offs, float* xyz, int32_t* vox, int32_t* meta,
    double* loss_acc, xrd_stream_t stream) {
  if (n_rays < 0 || n_max < 1 || n_max > 64 || s_cap < 1 || p_cap < 1 ||
      !(step_size > 0.f))
    return XRD_ERR_ARG;
  if (s_cap > 1024) return XRD_ERR_UNSUPPORTED;  // sample rows live in LDS
  if (!meta || !loss_acc) return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n_rays > 0 &&
      (!rays_o || !rays_d || !target_d || !hit_idx || !hit_min || !hit_max ||
       !probs || !steps || !hit || !rank || !hit_rays || !s_idx || !s_depth ||
       !cnt || !offs || !xyz || !vox))
    return XRD_ERR_ARG;
  if (centres != nullptr && (n_nodes < 1 || !children)) return XRD_ERR_ARG;
  PipeArgs A;
  A.n_rays = n_rays; A.n_max = n_max; A.s_cap = s_cap; A.n_nodes = n_nodes;
  A.p_cap = p_cap; A.voxel_size = voxel_size; A.max_distance = max_distance;
  A.inv_step = 1.0f / step_size; A.trunc = trunc; A.max_depth = max_depth;
  A.centres = centres; A.rays_o = rays_o; A.rays_d = rays_d;
  A.target_d = target_d; A.noise = noise; A.children = children;
  A.ray_keep = ray_keep; A.hit_idx = hit_idx; A.hit = hit; A.rank = rank;
  A.hit_rays = hit_rays; A.s_idx = s_idx; A.cnt = cnt; A.offs = offs;
  A.vox = vox; A.meta = meta; A.hit_min = hit_min; A.hit_max = hit_max;
  A.probs = probs; A.steps = steps; A.s_depth = s_depth; A.xyz = xyz;
  A.loss_acc = loss_acc;
  const size_t lds =
      ((size_t)(n_rays + 3) / 4 * 4 + (size_t)kWaves * (64 + 2 * s_cap)) * 4;
  if (lds > 150 * 1024) return XRD_ERR_UNSUPPORTED;   // > ~30 k rays a batch
  static size_t lds_set = 0;
  if (lds > lds_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(vox_sample_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute");
    lds_set = lds;
  }
  int nb = (n_rays + kWaves - 1) / kWaves;
  nb = nb < 1 ? 1 : nb > kPipeBlocks ? kPipeBlocks : nb;
  const dim3 grid(nb), block(kWaves * 64);
  hipLaunchKernelGGL(vox_hits_kernel, grid, block, 0, st, A);
  hipLaunchKernelGGL(vox_sample_kernel, grid, block, lds, st, A);
  hipLaunchKernelGGL(vox_compact_kernel, grid, block, 0, st, A);
  return check_launch("xrd_vox_sample_rays");
}

int xrd_vox_render_fwd(int n_rays, int s_cap, int64_t p_cap, float trunc,
                       float max_depth, const int32_t* hit,
                       const int32_t* cnt, const int32_t* offs,
                       const float* s_depth, const float* sdf_pt,
                       const float* rgb_pt, const float* target_d,
                       const float* target_rgb, const int32_t* meta,
                       float* depth, float* rgb, float* z_min, float* weights,
                       double* loss_acc, float w_rgb, float w_depth,
                       float w_sdf, float w_fs, float* loss, float* scale,
                       xrd_stream_t stream) {
  if (n_rays < 0 || s_cap < 1 || p_cap < 1 || !(trunc > 0.f))
    return XRD_ERR_ARG;
  if (n_rays == 0) return XRD_OK;
  if (!hit || !cnt || !offs || !s_depth || !sdf_pt || !rgb_pt || !meta ||
      !depth || !rgb)
    return XRD_ERR_ARG;
  if (loss_acc && (!target_d || !target_rgb || !loss || !scale))
    return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const RenderArgs a = {n_rays, s_cap, p_cap, trunc, 1.0f / trunc, max_depth};
  hipLaunchKernelGGL(vox_render_fwd_kernel,
                     dim3((n_rays + kWaves - 1) / kWaves), dim3(kWaves * 64),
                     0, st, a, hit, cnt, offs, s_depth, sdf_pt, rgb_pt,
                     target_d, target_rgb, meta, depth, rgb, z_min, weights,
                     loss_acc);
  if (loss_acc)
    hipLaunchKernelGGL(vox_loss_finalize_kernel, dim3(1), dim3(64), 0, st,
                       meta, s_cap, loss_acc, w_rgb, w_depth, w_sdf, w_fs,
                       loss, reinterpret_cast<LossScale*>(scale));
  return check_launch("xrd_vox_render_fwd");
}

int xrd_vox_render_bwd(int n_rays, int s_cap, int64_t p_cap, float trunc,
                       float max_depth, const int32_t* hit,
                       const int32_t* cnt, const int32_t* offs,
                       const float* s_depth, const float* sdf_pt,
                       const float* rgb_pt, const float* target_d,
                       const float* target_rgb, const int32_t* meta,
                       const float* scale, const float* g_loss, float* g_sdf,
                       float* g_rgb, xrd_stream_t stream) {
  if (n_rays < 0 || s_cap < 1 || p_cap < 1 || !(trunc > 0.f))
    return XRD_ERR_ARG;
  if (n_rays == 0) return XRD_OK;
  if (!hit || !cnt || !offs || !s_depth || !sdf_pt || !rgb_pt || !target_d ||
      !target_rgb || !meta || !scale || !g_sdf || !g_rgb)
    return XRD_ERR_ARG;
  const RenderArgs a = {n_rays, s_cap, p_cap, trunc, 1.0f / trunc, max_depth};
  hipLaunchKernelGGL(vox_render_bwd_kernel,
                     dim3((n_rays + kWaves - 1) / kWaves), dim3(kWaves * 64),
                     0, (hipStream_t)stream, a, hit, cnt, offs, s_depth,
                     sdf_pt, rgb_pt, target_d, target_rgb, meta,
                     reinterpret_cast<const LossScale*>(scale), g_loss, g_sdf,
                     g_rgb);
  return check_launch("xrd_vox_render_bwd");
}

int xrd_vox_ray_grads(int n_rays, int s_cap, int64_t p_cap, const int32_t* hit,
                      const int32_t* cnt, const int32_t* offs,
                      const float* s_depth, const float* g_xyz,
                      float* g_rays_o, float* g_rays_d, xrd_stream_t stream) {
  if (n_rays < 0 || s_cap < 1 || p_cap < 1) return XRD_ERR_ARG;
  if (n_rays == 0) return XRD_OK;
  if (!hit || !cnt || !offs || !s_depth || !g_xyz || !g_rays_o || !g_rays_d)
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(vox_ray_grad_kernel, dim3((n_rays + kWaves - 1) / kWaves),
                     dim3(kWaves * 64), 0, (hipStream_t)stream, n_rays, s_cap,
                     p_cap, hit, cnt, offs, s_depth, g_xyz, g_rays_o,
                     g_rays_d);
  return check_launch("xrd_vox_ray_grads");
}

}  // extern "C"
