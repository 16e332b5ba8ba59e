// Multi-resolution hash-grid and OneBlob encodings for gfx950 — the two
// tiny-cuda-nn encodings Co-SLAM instantiates
// (slam/model_components/encodings_coslam.py:43-53, 68-75).  tiny-cuda-nn is
// not vendored in the reference; the arithmetic follows SURVEY.md App. C.1/C.2
// (oracle: oracle/tcnn_oracle.py, parity unpinned by the reference).
//
// HBM/cache-bound gathers: 16 levels x 8 corners x 8 B per point.  Lane
// mapping: 16 consecutive lanes = the 16 levels of ONE point, so the [N,32]
// output row (and dL/dy) is one coalesced 128-B line per point and the three
// coordinates are a broadcast load.
#include "common.h"

namespace xrd {
namespace {

struct HashLevel {
  float scale;
  uint32_t res;
  uint32_t size;    // entries in this level
  uint32_t offset;  // first entry
};
constexpr int kMaxLevels = 32;
struct HashMeta {
  HashLevel lv[kMaxLevels];
  int n_levels;
};

__device__ __forceinline__ uint32_t grid_index(const HashLevel& L, uint32_t cx,
                                               uint32_t cy, uint32_t cz) {
  // tcnn grid_index: dense strides while stride <= size, otherwise the
  // coherent prime hash; always modulo the level size
  uint64_t stride = 1;
  uint32_t idx = 0;
  const uint32_t c[3] = {cx, cy, cz};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (stride <= L.size) {
      idx += c[d] * (uint32_t)stride;
      stride *= L.res;
    }
  }
  if ((uint64_t)L.size < stride)
    idx = (cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u);
  return idx % L.size;
}

template <bool BWD>
__global__ __launch_bounds__(256) void hashgrid_kernel(
    HashMeta M, int64_t n, const float* __restrict__ x,
    const float* __restrict__ params, float* __restrict__ y,
    const float* __restrict__ dy, float* __restrict__ dparams,
    float* __restrict__ dx) {
  const int L = M.n_levels;  // <= 16 levels per 16-lane group per pass
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lvl_lane = threadIdx.x & 15;
  const int64_t pt = gid >> 4;
  const bool pvalid = pt < n;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (pvalid) {
    px = x[pt * 3 + 0];
    py = x[pt * 3 + 1];
    pz = x[pt * 3 + 2];
  }
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int lvl = lvl_lane; lvl < L; lvl += 16) {
    if (!pvalid) break;
    const HashLevel lv = M.lv[lvl];
    const float fx = fmaf(lv.scale, px, 0.5f), fy = fmaf(lv.scale, py, 0.5f),
                fz = fmaf(lv.scale, pz, 0.5f);
    const float ffx = floorf(fx), ffy = floorf(fy), ffz = floorf(fz);
    const uint32_t cx = (uint32_t)(int)ffx, cy = (uint32_t)(int)ffy,
                   cz = (uint32_t)(int)ffz;
    const float wx = fx - ffx, wy = fy - ffy, wz = fz - ffz;
    float2 acc = {0.f, 0.f};
    float2 g = {0.f, 0.f};
    if (BWD) g = *reinterpret_cast<const float2*>(dy + pt * (2 * L) + 2 * lvl);
    float dgx = 0.f, dgy = 0.f, dgz = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint32_t bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
      const float ax = bx ? wx : 1.f - wx, ay = by ? wy : 1.f - wy,
                  az = bz ? wz : 1.f - wz;
      const uint32_t idx = lv.offset + grid_index(lv, cx + bx, cy + by, cz + bz);
      const float2 v = *reinterpret_cast<const float2*>(params + 2 * (size_t)idx);
      if (!BWD) {
        const float w = ax * ay * az;
        acc.x = fmaf(w, v.x, acc.x);
        acc.y = fmaf(w, v.y, acc.y);
      } else {
        const float w = ax * ay * az;
        if (dparams != nullptr) {
          atomicAdd(dparams + 2 * (size_t)idx, w * g.x);
          atomicAdd(dparams + 2 * (size_t)idx + 1, w * g.y);
        }
        if (dx != nullptr) {
          const float dv = v.x * g.x + v.y * g.y;
          dgx += (bx ? 1.f : -1.f) * ay * az * dv;
          dgy += (by ? 1.f : -1.f) * ax * az * dv;
          dgz += (bz ? 1.f : -1.f) * ax * ay * dv;
        }
      }
    }
    if (!BWD) {
      *reinterpret_cast<float2*>(y + pt * (2 * L) + 2 * lvl) = acc;
    } else {
      gx = fmaf(lv.scale, dgx, gx);
      gy = fmaf(lv.scale, dgy, gy);
      gz = fmaf(lv.scale, dgz, gz);
    }
  }
  if (BWD && dx != nullptr) {
    // sum over the 16 level lanes of the point
    gx = row16_sum(gx);
    gy = row16_sum(gy);
    gz = row16_sum(gz);
    if (pvalid && lvl_lane == 0) {
      dx[pt * 3 + 0] = gx;
      dx[pt * 3 + 1] = gy;
      dx[pt * 3 + 2] = gz;
    }
  }
}

// ---- Co-SLAM smoothness term (joint_encoding.py:165-197) as kernels ---------------
// total variation of the hash features on a random R^3 lattice (R = 31 points
// a side, 0.1 m apart): the lattice points (f64 like the reference's bbox
// arithmetic), then — after a hashgrid forward over them — the loss and
// d loss / d features in one launch.  ~35 torch launches (coordinates, the
// arithmetic around them, three slice-subtract-square-sum chains and their
// autograd backward) become three.
__global__ __launch_bounds__(256) void tv_points_kernel(
    int R, double vs, double margin, double b0x, double b0y, double b0z,
    double vx, double vy, double vz, const double* __restrict__ r_off,
    const double* __restrict__ r_shift, float* __restrict__ pts) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= R * R * R) return;
  const int c[3] = {p / (R * R), (p / R) % R, p % R};
  const double b0[3] = {b0x, b0y, b0z}, vol[3] = {vx, vy, vz};
  const double grid_size = (double)R * vs;   // (sample_points - 1) * voxel
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double omax = vol[a] - grid_size - 2.0 * margin;
    const double off = r_off[a] * omax + margin;
    // pts = (coords + rand) * voxel + bb0 + offset;  (pts - bb0) / volume
    const double x = (((double)c[a] + r_shift[a]) * vs + b0[a]) + off;
    pts[p * 3 + a] = (float)((x - b0[a]) / vol[a]);
  }
}

// one 16-lane group per lattice point, lane = level (float2 per level)
__global__ __launch_bounds__(256) void tv_loss_kernel(
    int R, int L, float gscale, const float* __restrict__ feat,
    float* __restrict__ dfeat, double* __restrict__ loss) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lvl_lane = threadIdx.x & 15;
  const int p = (int)(gid >> 4);
  const int P = R * R * R;
  float part = 0.f;
  if (p < P) {
    const int c[3] = {p / (R * R), (p / R) % R, p % R};
    const int st[3] = {R * R, R, 1};
    for (int l = lvl_lane; l < L; l += 16) {
      const float2 f0 =
          *reinterpret_cast<const float2*>(feat + (size_t)p * 2 * L + 2 * l);
      float2 g = {0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (c[a] + 1 < R) {
          const float2 f1 = *reinterpret_cast<const float2*>(
              feat + (size_t)(p + st[a]) * 2 * L + 2 * l);
          const float dx = f1.x - f0.x, dy = f1.y - f0.y;
          part += dx * dx + dy * dy;
          g.x -= 2.f * dx;
          g.y -= 2.f * dy;
        }
        if (c[a] > 0) {
          const float2 f1 = *reinterpret_cast<const float2*>(
              feat + (size_t)(p - st[a]) * 2 * L + 2 * l);
          g.x += 2.f * (f0.x - f1.x);
          g.y += 2.f * (f0.y - f1.y);
        }
      }
      *reinterpret_cast<float2*>(dfeat + (size_t)p * 2 * L + 2 * l) =
          make_float2(g.x * gscale, g.y * gscale);
    }
  }
  __shared__ double sh[4];
  const double s = wave_sum((double)part);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    if (t != 0.0) atomicAdd(loss, t * (double)gscale);
  }
}

// ---- gradient scatter, LDS-privatised per (level, chunk, point slice) --------
// Random f32 global atomics run at ~5-10 G transactions/s on MI355X (measured:
// 1.3 ms for the 7 M corner updates of one Co-SLAM mapping batch).  The tables
// are small (2^16..2^19 entries per level), so each block takes one 8192-entry
// chunk of one level into LDS, re-derives the corner indices of the points of
// its slice for that level (a few integer ops) and keeps only the hits; the
// chunk is then added to the table with coalesced atomics (non-zero entries
// only).  LDS f32 atomics retire well under one lane-op per clock per CU
// (~0.4 measured in round 4), so the work is balanced over >= 32 blocks per
// level: levels with few chunks are split into more point slices.
constexpr int kChunk = 8192;       // entries (x2 floats = 64 KB of LDS)
constexpr int kBlocksPerLevel = 32;
struct ChunkMeta {
  HashLevel lv[kMaxLevels];
  uint32_t first_block[kMaxLevels + 1];  // prefix of blocks per level
  uint32_t slices[kMaxLevels];           // point slices of each level
  int n_levels;
};

// Round 4: the scatter with RUN MERGING (round 3 visited the points in a
// strided permutation, one point a thread: half of its 233 us were the
// permutation's 64-line gathers, half the LDS atomics).  The points of a Co-SLAM batch
// are consecutive samples of rays (and the lines of the smoothness lattice):
// neighbours in memory are neighbours in space.  A thread walks a run of
// ``run_len`` consecutive points, keeps the 8 corners of the current cell and
// their sums in registers, and touches LDS only when the run leaves the cell:
// fewer ds_add_f32 lane-operations (the half of the round-3 kernel that is
// bound by the ~0.4 lane-op / clock / CU this chip retires) and consecutive
// addresses per lane instead of the 64-line gathers of the permutation (the
// other half; profiles/r04_coslam_scatter_experiments.txt).  Correct for any
// run length and any point order (a run that crosses into another ray just
// flushes).  Measured, synthetic rays (102 727 points): 233 us -> 166 us with
// runs of 8 (4: 181, 16: 199, 43 = one thread a ray: 309 — too few threads),
// at 44 032 points runs of 4: 111 -> 91 us; in the Co-SLAM mapping iteration
// (sorted near-surface + uniform samples, + the smoothness lattice: 135 k
// points) 267 us -> 204 us with runs of 6 (4: 221, 5: 202, 8: 211, 12: 267).
__global__ __launch_bounds__(1024) void hash_chunk_scatter_runs_kernel(
    ChunkMeta M, int64_t n, int run_len, const float* __restrict__ x,
    const float* __restrict__ dy, int64_t point_stride, int64_t level_stride,
    float* __restrict__ dparams) {
  __shared__ float acc[2 * kChunk];
  int lvl = 0;
  while (lvl + 1 < M.n_levels && blockIdx.x >= M.first_block[lvl + 1]) ++lvl;
  const HashLevel lv = M.lv[lvl];
  const uint32_t n_slices = M.slices[lvl];
  const uint32_t rel = blockIdx.x - M.first_block[lvl];
  const uint32_t chunk = rel / n_slices, slice = rel - chunk * n_slices;
  const uint32_t lo = chunk * kChunk;
  const uint32_t cnt = min((uint32_t)kChunk, lv.size - lo);
  for (int i = threadIdx.x; i < 2 * (int)cnt; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const float* gy = dy + (int64_t)lvl * level_stride;
  const bool dense = (uint64_t)lv.res * lv.res * lv.res <= (uint64_t)lv.size;
  const uint32_t hmask = lv.size - 1;
  const int64_t n_runs = (n + run_len - 1) / run_len;
  for (int64_t r = (int64_t)slice * blockDim.x + threadIdx.x; r < n_runs;
       r += (int64_t)blockDim.x * n_slices) {
    const int64_t p0 = r * run_len;
    const int len = (int)min((int64_t)run_len, n - p0);
    uint32_t ccx = 0, ccy = 0, ccz = 0;
    bool have = false;  // (no sentinel cell: points left of the unit cube
                        // wrap to 0xffffffff like in the forward)
    uint32_t idx[8];
    float ax[8], ay[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      idx[c] = 0xffffffffu;
      ax[c] = 0.f;
      ay[c] = 0.f;
    }
    for (int s = 0; s < len; ++s) {
      const int64_t p = p0 + s;
      const float2 g = *reinterpret_cast<const float2*>(gy + p * point_stride);
      if (g.x == 0.f && g.y == 0.f) continue;
      const float px = x[p * 3 + 0], py = x[p * 3 + 1], pz = x[p * 3 + 2];
      const float fx = fmaf(lv.scale, px, 0.5f), fy = fmaf(lv.scale, py, 0.5f),
                  fz = fmaf(lv.scale, pz, 0.5f);
      const float ffx = floorf(fx), ffy = floorf(fy), ffz = floorf(fz);
      const uint32_t cx = (uint32_t)(int)ffx, cy = (uint32_t)(int)ffy,
                     cz = (uint32_t)(int)ffz;
      const float wx = fx - ffx, wy = fy - ffy, wz = fz - ffz;
      if (!have || cx != ccx || cy != ccy || cz != ccz) {
        have = true;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (idx[c] < cnt && (ax[c] != 0.f || ay[c] != 0.f)) {
            atomicAdd(&acc[2 * idx[c]], ax[c]);
            atomicAdd(&acc[2 * idx[c] + 1], ay[c]);
          }
          const uint32_t bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
          const uint32_t ux = cx + bx, uy = cy + by, uz = cz + bz;
          const uint32_t full = dense
              ? (ux + uy * lv.res + uz * lv.res * lv.res) % lv.size
              : ((ux * 1u) ^ (uy * 2654435761u) ^ (uz * 805459861u)) & hmask;
          idx[c] = full - lo;
          ax[c] = 0.f;
          ay[c] = 0.f;
        }
        ccx = cx;
        ccy = cy;
        ccz = cz;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
        const float w = (bx ? wx : 1.f - wx) * (by ? wy : 1.f - wy) *
                        (bz ? wz : 1.f - wz);
        ax[c] = fmaf(w, g.x, ax[c]);
        ay[c] = fmaf(w, g.y, ay[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (idx[c] < cnt && (ax[c] != 0.f || ay[c] != 0.f)) {
        atomicAdd(&acc[2 * idx[c]], ax[c]);
        atomicAdd(&acc[2 * idx[c] + 1], ay[c]);
      }
  }
  __syncthreads();
  float* out = dparams + 2 * ((size_t)lv.offset + lo);
  for (int i = threadIdx.x; i < 2 * (int)cnt; i += blockDim.x) {
    const float v = acc[i];
    if (v != 0.f) atomicAdd(out + i, v);
  }
}

__device__ __forceinline__ float quartic_cdf(float x, float inv_r) {
  const float u = x * inv_r, u2 = u * u, u4 = u2 * u2;
  return fminf(fmaxf((15.f / 16.f) * u * (1.f - (2.f / 3.f) * u2 +
                                          (1.f / 5.f) * u4) + 0.5f, 0.f), 1.f);
}
__device__ __forceinline__ float quartic_pdf(float x, float inv_r) {
  const float u = x * inv_r;
  if (fabsf(u) >= 1.f) return 0.f;
  const float t = 1.f - u * u;
  return (15.f / 16.f) * t * t * inv_r;
}

// one thread per (point, dim); writes/reads n_bins contiguous values
template <bool BWD>
__global__ __launch_bounds__(256) void oneblob_kernel(
    int64_t n, int dims, int n_bins, const float* __restrict__ x,
    float* __restrict__ y, const float* __restrict__ dy,
    float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dims) return;
  const float v = x[i];
  const float nb = (float)n_bins;
  const int64_t base = i * n_bins;  // [N][dims][n_bins]
  if (!BWD) {
    float left = quartic_cdf(-v, nb) + quartic_cdf(-v - 1.f, nb) +
                 quartic_cdf(-v + 1.f, nb);
    for (int k = 0; k < n_bins; ++k) {
      const float rb = (float)(k + 1) / nb - v;
      const float right = quartic_cdf(rb, nb) + quartic_cdf(rb - 1.f, nb) +
                          quartic_cdf(rb + 1.f, nb);
      y[base + k] = right - left;
      left = right;
    }
  } else {
    // d/dx cdf(t - x) = -pdf(t - x)
    float left = quartic_pdf(-v, nb) + quartic_pdf(-v - 1.f, nb) +
                 quartic_pdf(-v + 1.f, nb);
    float g = 0.f;
    for (int k = 0; k < n_bins; ++k) {
      const float rb = (float)(k + 1) / nb - v;
      const float right = quartic_pdf(rb, nb) + quartic_pdf(rb - 1.f, nb) +
                          quartic_pdf(rb + 1.f, nb);
      g += dy[base + k] * (-(right - left));
      left = right;
    }
    dx[i] = g;
  }
}

int fill_meta(HashMeta& M, int n_levels, const float* scales,
              const uint32_t* res, const uint32_t* sizes,
              const uint32_t* offsets) {
  if (n_levels < 1 || n_levels > kMaxLevels || !scales || !res || !sizes ||
      !offsets)
    return XRD_ERR_ARG;
  M.n_levels = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    if (sizes[l] == 0 || res[l] == 0) return XRD_ERR_ARG;
    M.lv[l] = HashLevel{scales[l], res[l], sizes[l], offsets[l]};
  }
  return XRD_OK;
}

}  // namespace

int launch_hash_chunk_scatter(int n_levels, const float* scales,
                              const uint32_t* res, const uint32_t* sizes,
                              const uint32_t* offsets, int64_t n_points,
                              const float* x, const float* dy,
                              int64_t point_stride, int64_t level_stride,
                              float* dparams, bool accumulate, void* stream) {
  HashMeta H;
  int rc = fill_meta(H, n_levels, scales, res, sizes, offsets);
  if (rc != XRD_OK) return rc;
  if (n_points < 0 || !dparams || (n_points > 0 && (!x || !dy)))
    return XRD_ERR_ARG;
  for (int l = 0; l < n_levels; ++l) {
    const uint64_t r = res[l], sz = sizes[l];
    if (r * r * r > sz && (sz & (sz - 1)) != 0) return XRD_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    const uint64_t total = (uint64_t)offsets[n_levels - 1] + sizes[n_levels - 1];
    rc = zero_floats(dparams, (size_t)total * 2, stream);
    if (rc != XRD_OK) return rc;
  }
  if (n_points == 0) return XRD_OK;
  ChunkMeta M;
  M.n_levels = n_levels;
  uint32_t blocks = 0;
  for (int l = 0; l < n_levels; ++l) {
    M.lv[l] = H.lv[l];
    M.first_block[l] = blocks;
    const uint32_t chunks = (sizes[l] + kChunk - 1) / kChunk;
    M.slices[l] = chunks >= kBlocksPerLevel ? 1 : kBlocksPerLevel / chunks;
    blocks += chunks * M.slices[l];
  }
  M.first_block[n_levels] = blocks;
  const int run_len = n_points >= 65536 ? 6 : 4;
  hipLaunchKernelGGL(hash_chunk_scatter_runs_kernel, dim3(blocks), dim3(1024),
                     0, st, M, n_points, run_len, x, dy, point_stride,
                     level_stride, dparams);
  return check_launch("hash_chunk_scatter_runs_kernel");
}

}  // namespace xrd

using namespace xrd;

extern "C" {

int xrd_hashgrid_levels(int n_levels, int base_resolution,
                        float per_level_scale, int log2_hashmap_size,
                        int dense, float* scales, uint32_t* res,
                        uint32_t* sizes, uint32_t* offsets,
                        uint32_t* total_entries) {
  if (n_levels < 1 || n_levels > kMaxLevels || !scales || !res || !sizes ||
      !offsets || !total_entries)
    return XRD_ERR_ARG;
  const float log2_pls = log2f(per_level_scale);
  uint64_t off = 0;
  const uint64_t cap = 1ull << log2_hashmap_size;
  for (int l = 0; l < n_levels; ++l) {
    const float scale = exp2f((float)l * log2_pls) * (float)base_resolution - 1.0f;
    const uint32_t r = (uint32_t)ceilf(scale) + 1u;
    uint64_t n = (uint64_t)r * r * r;
    n = (n + 7) / 8 * 8;
    if (!dense && n > cap) n = cap;
    if (n > 0xffffffffull || off + n > 0xffffffffull) return XRD_ERR_UNSUPPORTED;
    scales[l] = scale;
    res[l] = r;
    sizes[l] = (uint32_t)n;
    offsets[l] = (uint32_t)off;
    off += n;
  }
  *total_entries = (uint32_t)off;
  return XRD_OK;
}

int xrd_hashgrid_fwd(int n_levels, const float* scales, const uint32_t* res,
                     const uint32_t* sizes, const uint32_t* offsets,
                     int64_t n_points, const float* x, const float* params,
                     float* y, xrd_stream_t stream) {
  HashMeta M;
  int rc = fill_meta(M, n_levels, scales, res, sizes, offsets);
  if (rc != XRD_OK) return rc;
  if (n_points < 0 || (n_points > 0 && (!x || !params || !y))) return XRD_ERR_ARG;
  if (n_points == 0) return XRD_OK;
  const int64_t threads = n_points * 16;
  hipLaunchKernelGGL((hashgrid_kernel<false>), dim3((unsigned)((threads + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, M, n_points, x, params,
                     y, nullptr, nullptr, nullptr);
  return check_launch("xrd_hashgrid_fwd");
}

int xrd_hashgrid_bwd(int n_levels, const float* scales, const uint32_t* res,
                     const uint32_t* sizes, const uint32_t* offsets,
                     int64_t n_points, const float* x, const float* params,
                     const float* dy, float* dparams, float* dx,
                     xrd_stream_t stream) {
  HashMeta M;
  int rc = fill_meta(M, n_levels, scales, res, sizes, offsets);
  if (rc != XRD_OK) return rc;
  if (n_points < 0 || (n_points > 0 && (!x || !params || !dy))) return XRD_ERR_ARG;
  if (n_points == 0 || (!dparams && !dx)) return XRD_OK;
  if (dparams != nullptr) {
    rc = launch_hash_chunk_scatter(n_levels, scales, res, sizes, offsets,
                                   n_points, x, dy, 2 * n_levels, 2, dparams,
                                   /*accumulate=*/true, stream);
    if (rc != XRD_OK) return rc;
  }
  if (dx != nullptr) {
    const int64_t threads = n_points * 16;
    hipLaunchKernelGGL((hashgrid_kernel<true>),
                       dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, M, n_points, x, params, nullptr, dy,
                       nullptr, dx);
    rc = check_launch("xrd_hashgrid_bwd");
  }
  return rc;
}

int xrd_hashgrid_tv(int n_levels, const float* scales, const uint32_t* res,
                    const uint32_t* sizes, const uint32_t* offsets,
                    const float* params, int side, const double* bound6,
                    double voxel_size, double margin,
                    const double* rand_offset, const double* rand_shift,
                    float scale, float* points, float* feat, float* dfeat,
                    double* loss, xrd_stream_t stream) {
  HashMeta M;
  int rc = fill_meta(M, n_levels, scales, res, sizes, offsets);
  if (rc != XRD_OK) return rc;
  if (side < 2 || side > 128 || !bound6 || !(voxel_size > 0.0))
    return XRD_ERR_ARG;
  if (!params || !rand_offset || !rand_shift || !points || !feat || !dfeat ||
      !loss)
    return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int P = side * side * side;
  rc = zero_floats(reinterpret_cast<float*>(loss), 2, stream);
  if (rc != XRD_OK) return rc;
  hipLaunchKernelGGL(tv_points_kernel, dim3((P + 255) / 256), dim3(256), 0, st,
                     side, voxel_size, margin, bound6[0], bound6[2], bound6[4],
                     bound6[1] - bound6[0], bound6[3] - bound6[2],
                     bound6[5] - bound6[4], rand_offset, rand_shift, points);
  const int64_t threads = (int64_t)P * 16;
  hipLaunchKernelGGL((hashgrid_kernel<false>),
                     dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                     M, (int64_t)P, points, params, feat, nullptr, nullptr,
                     nullptr);
  // loss = scale * tv / (side + 1)^3 (joint_encoding.py:197: sample_points^3)
  const double n3 = (double)(side + 1) * (side + 1) * (side + 1);
  hipLaunchKernelGGL(tv_loss_kernel, dim3((unsigned)((threads + 255) / 256)),
                     dim3(256), 0, st, side, n_levels, (float)(scale / n3),
                     feat, dfeat, loss);
  return check_launch("xrd_hashgrid_tv");
}

int xrd_oneblob_fwd(int64_t n_points, int dims, int n_bins, const float* x,
                    float* y, xrd_stream_t stream) {
  if (n_points < 0 || dims < 1 || n_bins < 1 || (n_points > 0 && (!x || !y)))
    return XRD_ERR_ARG;
  if (n_points == 0) return XRD_OK;
  const int64_t t = n_points * dims;
  hipLaunchKernelGGL((oneblob_kernel<false>), dim3((unsigned)((t + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, n_points, dims, n_bins,
                     x, y, nullptr, nullptr);
  return check_launch("xrd_oneblob_fwd");
}

int xrd_oneblob_bwd(int64_t n_points, int dims, int n_bins, const float* x,
                    const float* dy, float* dx, xrd_stream_t stream) {
  if (n_points < 0 || dims < 1 || n_bins < 1 ||
      (n_points > 0 && (!x || !dy || !dx)))
    return XRD_ERR_ARG;
  if (n_points == 0) return XRD_OK;
  const int64_t t = n_points * dims;
  hipLaunchKernelGGL((oneblob_kernel<true>), dim3((unsigned)((t + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, n_points, dims, n_bins,
                     x, nullptr, dy, dx);
  return check_launch("xrd_oneblob_bwd");
}

}  // extern "C"
