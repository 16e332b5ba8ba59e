// SplaTAM rasteriser, tile binning on the device (the [ext] rasteriser's
// BinningState: prefix sum of tiles_touched, key duplication, 64-bit radix
// sort, per-tile ranges — call sites of the reference:
// slam/model_components/gaussian_cloud_splatam.py:63-69,267-268).  Round 1 did
// this with torch.cumsum + `.item()` + torch.sort + a gather, i.e. a host
// sync in every raster pass; round 2 / early round 3 with one library radix
// sort of (tile | depth) keys over the whole static-capacity list (rocPRIM
// picks its merge sort at ~1 M keys: 165 us a pass, a fifth of the pass).
//
// The keys are (tile, depth): the sort is a BUCKET step by tile followed by
// small independent sorts by depth.  One C call on the stream:
//   1. inclusive scan of tiles_touched (rocPRIM) — offsets of a Gaussian's
//      keys in the pre-sort order (the key-gradient reduction of gs_blend.hip
//      walks them);
//   2. count: one atomic per (Gaussian, tile) pair on the tile's counter;
//   3. one block scans the tile counters: list ranges, total pair count;
//   4. fill: a second atomic per pair hands out the slot inside the tile's
//      range; the slot order is arbitrary;
//   5. one block per tile sorts its range by (depth bits, Gaussian id)
//      (bitonic over keys held in registers: in-thread and in-wave exchanges,
//      LDS only for the top distances; 64-bit composite keys; ties fall in
//      Gaussian order, which is
//      what the stable library sort of keys emitted in Gaussian order gave)
//      and writes the Gaussian list and the map "sorted position -> pre-sort
//      pair index" (the blend backward files a pair's gradient row under its
//      pre-sort index: a Gaussian's rows are then contiguous).  A list longer than the LDS holds is rank-sorted
//      from global memory (slow, correct).
// The true number of (Gaussian, tile) pairs is left in a device scalar; the
// caller sizes the capacity from the previous pass's count, read back
// asynchronously, and is told when a pass did not fit (pairs whose position
// lies behind the capacity are dropped).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace xrd {
namespace {

constexpr int TILE = 16;

constexpr int SORT_THREADS = 256;
constexpr int SORT_LDS_KEYS = 4096;   // 32 KB of 64-bit keys a block

constexpr int BIN_PER_THREAD = 4;      // Gaussians a thread of count / fill
constexpr int BIN_LDS_TILES = 8192;    // tile counters a block keeps in LDS

// Random global atomics retire ~5-10 G/s on this part: 0.7 M
// pairs twice would cost ~200 us.  A block counts its 1024 Gaussians' pairs
// in an LDS histogram (Gaussians are stored in creation order: neighbours in
// memory are neighbours on screen, a block touches a few dozen tiles) and
// touches the global counters once per (block, tile).
template <bool LDS>
__global__ __launch_bounds__(256) void gs_tile_count_kernel(
    int n, int n_tiles, const int* __restrict__ rect, int grid_x,
    const int64_t* __restrict__ offsets, int64_t cap,
    int* __restrict__ tile_count) {
  __shared__ int hist[LDS ? BIN_LDS_TILES : 1];
  if (LDS) {
    for (int t = threadIdx.x; t < n_tiles; t += 256) hist[t] = 0;
    __syncthreads();
  }
  const int base = blockIdx.x * 256 * BIN_PER_THREAD;
#pragma unroll
  for (int k = 0; k < BIN_PER_THREAD; ++k) {
    const int i = base + k * 256 + threadIdx.x;
    if (i >= n) continue;
    const int x0 = rect[i * 4], y0 = rect[i * 4 + 1], x1 = rect[i * 4 + 2],
              y1 = rect[i * 4 + 3];
    // pairs are numbered in Gaussian order; the ones whose number does not
    // fit the capacity are dropped (the total is reported)
    int64_t pre = i == 0 ? 0 : offsets[i - 1];
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x, ++pre)
        if (pre < cap)
          atomicAdd((LDS ? hist : tile_count) + y * grid_x + x, 1);
  }
  if (LDS) {
    __syncthreads();
    for (int t = threadIdx.x; t < n_tiles; t += 256) {
      const int c = hist[t];
      if (c != 0) atomicAdd(tile_count + t, c);
    }
  }
}

// one block: exclusive scan of the tile counters -> tile_start[nt + 1],
// ranges; the counters become the fill cursors (zero)
__global__ __launch_bounds__(1024) void gs_tile_scan_kernel(
    int n_tiles, int64_t cap, int* __restrict__ tile_count,
    int* __restrict__ tile_start, int* __restrict__ ranges) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (n_tiles + 1023) / 1024;
  const int lo = tid * per, hi = min(n_tiles, lo + per);
  int sum = 0;
  for (int t = lo; t < hi; ++t) sum += tile_count[t];
  part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int64_t run = (int64_t)part[tid] - sum;   // exclusive prefix of the slice
  for (int t = lo; t < hi; ++t) {
    const int c = tile_count[t];
    tile_start[t] = (int)run;
    const int64_t b = run < cap ? run : cap;
    const int64_t e = run + c < cap ? run + c : cap;
    ranges[t * 2] = (int)b;
    ranges[t * 2 + 1] = (int)e;
    tile_count[t] = 0;
    run += c;
  }
  if (tid == 1023) tile_start[n_tiles] = part[1023];
}

template <bool LDS>
__global__ __launch_bounds__(256) void gs_tile_fill_kernel(
    int n, int n_tiles, const int* __restrict__ rect,
    const float* __restrict__ depths, int grid_x, int64_t cap,
    const int* __restrict__ tile_start, const int64_t* __restrict__ offsets,
    int* __restrict__ cursor, uint64_t* __restrict__ keys,
    int* __restrict__ gid_pre, int* __restrict__ live_pre,
    int64_t* __restrict__ total) {
  // LDS: hist[t] = this block's pairs of tile t, then the block's base slot
  // in the tile's range (one global atomic per touched tile), then a second
  // LDS pass hands out the slots
  __shared__ int hist[LDS ? BIN_LDS_TILES : 1];
  __shared__ int rank[LDS ? BIN_LDS_TILES : 1];
  const int base = blockIdx.x * 256 * BIN_PER_THREAD;
  if (LDS) {
    for (int t = threadIdx.x; t < n_tiles; t += 256) {
      hist[t] = 0;
      rank[t] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BIN_PER_THREAD; ++k) {
      const int i = base + k * 256 + threadIdx.x;
      if (i >= n) continue;
      const int x0 = rect[i * 4], y0 = rect[i * 4 + 1], x1 = rect[i * 4 + 2],
                y1 = rect[i * 4 + 3];
      int64_t pre = i == 0 ? 0 : offsets[i - 1];
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x, ++pre)
          if (pre < cap) atomicAdd(hist + y * grid_x + x, 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_tiles; t += 256) {
      const int c = hist[t];
      if (c != 0) hist[t] = atomicAdd(cursor + t, c);
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < BIN_PER_THREAD; ++k) {
    const int i = base + k * 256 + threadIdx.x;
    if (i >= n) continue;
    if (i == n - 1) total[0] = offsets[n - 1];
    const int x0 = rect[i * 4], y0 = rect[i * 4 + 1], x1 = rect[i * 4 + 2],
              y1 = rect[i * 4 + 3];
    if ((x1 - x0) * (y1 - y0) == 0) continue;
    // key = depth bits | PRE-SORT index of the pair (pairs are numbered in
    // Gaussian order, a Gaussian's pairs in rect order): ties of depth fall
    // in Gaussian order, and the index is what the gradient rows are filed
    // under (gs_blend.hip)
    const uint64_t dkey = (uint64_t)__float_as_uint(depths[i]) << 32;
    int64_t pre = i == 0 ? 0 : offsets[i - 1];
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x, ++pre) {
        if (pre >= cap) continue;          // dropped (not counted either)
        const int t = y * grid_x + x;
        const int slot = LDS ? hist[t] + atomicAdd(rank + t, 1)
                             : atomicAdd(cursor + t, 1);
        keys[tile_start[t] + slot] = dkey | (uint64_t)(uint32_t)pre;
        gid_pre[pre] = i;
        if (live_pre != nullptr) live_pre[pre] = i;
      }
  }
}

// sorted position `pos` holds the pair with pre-sort index `pre`
__device__ __forceinline__ void emit_sorted(int64_t pos, uint32_t pre,
                                            const int* __restrict__ gid_pre,
                                            int* __restrict__ point_list,
                                            int* __restrict__ key_pos) {
  point_list[pos] = gid_pre[pre];
  if (key_pos != nullptr) key_pos[pos] = (int)pre;
}

// Bitonic sort of 256 * E keys held E a thread (element index = tid * E + r).
// Compare-exchange distances below E stay inside a thread, distances below
// 64 E are lane exchanges inside a wave (no LDS traffic, no barrier), only
// the top two or three distances cross waves through LDS: a 1024-key list
// sorts with 3 barrier stages instead of 55.
template <int E>
__device__ __forceinline__ void bitonic_sort_regs(uint64_t (&v)[E], int tid,
                                                  uint64_t* __restrict__ s) {
  constexpr int P = SORT_THREADS * E;
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < E) {
#pragma unroll
        for (int jj = E / 2; jj >= 1; jj >>= 1) {
          if (j != jj) continue;
#pragma unroll
          for (int r = 0; r < E; ++r) {
            if (r & jj) continue;
            const bool up = ((tid * E + r) & k) == 0;
            const uint64_t a = v[r], b = v[r | jj];
            if ((a > b) == up) {
              v[r] = b;
              v[r | jj] = a;
            }
          }
        }
      } else if (j < 64 * E) {
        const int lx = j / E;
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const uint64_t o = __shfl_xor((unsigned long long)v[r], lx);
          const int e = tid * E + r;
          const bool keep_min = ((e & j) == 0) == ((e & k) == 0);
          v[r] = keep_min ? (o < v[r] ? o : v[r]) : (o > v[r] ? o : v[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < E; ++r) s[tid * E + r] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int e = tid * E + r;
          const uint64_t o = s[e ^ j];
          const bool keep_min = ((e & j) == 0) == ((e & k) == 0);
          v[r] = keep_min ? (o < v[r] ? o : v[r]) : (o > v[r] ? o : v[r]);
        }
        __syncthreads();
      }
    }
  }
}

template <int E>
__device__ __forceinline__ void sort_tile(
    int tid, int r0, int L, uint64_t* __restrict__ s,
    const uint64_t* __restrict__ keys, const int* __restrict__ gid_pre,
    int* __restrict__ point_list, int* __restrict__ key_pos) {
  uint64_t v[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = tid * E + r;
    v[r] = i < L ? keys[r0 + i] : ~0ull;
  }
  bitonic_sort_regs<E>(v, tid, s);
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = tid * E + r;
    if (i < L)
      emit_sorted((int64_t)r0 + i, (uint32_t)v[r], gid_pre, point_list,
                  key_pos);
  }
}

__global__ __launch_bounds__(SORT_THREADS) void gs_tile_sort_kernel(
    const int* __restrict__ ranges, const uint64_t* __restrict__ keys,
    const int* __restrict__ gid_pre, int* __restrict__ point_list,
    int* __restrict__ key_pos) {
  __shared__ uint64_t s[SORT_LDS_KEYS];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int r0 = ranges[t * 2], r1 = ranges[t * 2 + 1];
  const int L = r1 - r0;
  if (L <= 0) return;
  if (L > SORT_LDS_KEYS) {
    // rank sort from global memory: position = number of smaller keys (keys
    // are distinct: the pair index is part of them)
    for (int i = tid; i < L; i += SORT_THREADS) {
      const uint64_t k = keys[r0 + i];
      int rank = 0;
      for (int j = 0; j < L; ++j) rank += keys[r0 + j] < k ? 1 : 0;
      emit_sorted((int64_t)r0 + rank, (uint32_t)k, gid_pre, point_list,
                  key_pos);
    }
    return;
  }
  if (L <= SORT_THREADS * 4)
    sort_tile<4>(tid, r0, L, s, keys, gid_pre, point_list, key_pos);
  else if (L <= SORT_THREADS * 8)
    sort_tile<8>(tid, r0, L, s, keys, gid_pre, point_list, key_pos);
  else
    sort_tile<16>(tid, r0, L, s, keys, gid_pre, point_list, key_pos);
}

struct BinLayout {
  size_t offsets, keys, gid_pre, tile_count, tile_start, temp, total;
};

struct ToI64 {
  __host__ __device__ int64_t operator()(int v) const { return (int64_t)v; }
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

BinLayout layout(int n, int64_t cap, int n_tiles) {
  BinLayout L;
  size_t scan = 0;
  (void)rocprim::inclusive_scan(
      nullptr, scan,
      rocprim::make_transform_iterator((const int*)nullptr, ToI64()),
      (int64_t*)nullptr, (size_t)n, rocprim::plus<int64_t>());
  size_t at = 0;
  L.offsets = at;
  at += align256((size_t)n * sizeof(int64_t));
  L.keys = at;
  at += align256((size_t)cap * sizeof(uint64_t));
  L.gid_pre = at;
  at += align256((size_t)cap * sizeof(int));
  L.tile_count = at;
  at += align256((size_t)(n_tiles + 1) * sizeof(int));
  L.tile_start = at;
  at += align256((size_t)(n_tiles + 1) * sizeof(int));
  L.temp = at;
  at += align256(scan);
  L.total = at;
  return L;
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int64_t xrd_gs_bin_ws_bytes(int n, int64_t key_capacity, int image_width,
                            int image_height) {
  if (n < 1 || key_capacity < 1 || image_width < 1 || image_height < 1)
    return 0;
  const int nt = ((image_width + TILE - 1) / TILE) *
                 ((image_height + TILE - 1) / TILE);
  return (int64_t)layout(n, key_capacity, nt).total;
}

static int bin_impl(int n, int image_width, int image_height,
                    const int32_t* rect, const int32_t* tiles_touched,
                    const float* depths, int64_t key_capacity,
                    void* workspace, int32_t* point_list, int32_t* ranges,
                    int64_t* n_keys, int32_t* key_pos, int64_t* offsets_out,
                    xrd_stream_t stream) {
  if (n < 0 || image_width < 1 || image_height < 1 || key_capacity < 1 ||
      key_capacity > 0x7fffffff)
    return XRD_ERR_ARG;
  if (!ranges || !n_keys) return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int gx = (image_width + TILE - 1) / TILE;
  const int nt = gx * ((image_height + TILE - 1) / TILE);
  if (n == 0) {
    // zero-fill as kernels (common.h: memset nodes in captured graphs)
    int rc = zero_floats(reinterpret_cast<float*>(ranges), (size_t)nt * 2, st);
    if (rc == XRD_OK)
      rc = zero_floats(reinterpret_cast<float*>(n_keys), 2, st);
    return rc;
  }
  if (!rect || !tiles_touched || !depths || !workspace || !point_list)
    return XRD_ERR_ARG;
  const BinLayout L = layout(n, key_capacity, nt);
  char* ws = static_cast<char*>(workspace);
  int64_t* offsets = offsets_out ? offsets_out
                                 : reinterpret_cast<int64_t*>(ws + L.offsets);
  uint64_t* keys = reinterpret_cast<uint64_t*>(ws + L.keys);
  int* gid_pre = reinterpret_cast<int*>(ws + L.gid_pre);
  int* tile_count = reinterpret_cast<int*>(ws + L.tile_count);
  int* tile_start = reinterpret_cast<int*>(ws + L.tile_start);
  int rc = zero_floats(reinterpret_cast<float*>(tile_count), (size_t)nt + 1,
                       st);
  if (rc != XRD_OK) return rc;
  size_t temp_bytes = L.total - L.temp;
  if (rocprim::inclusive_scan(
          ws + L.temp, temp_bytes,
          rocprim::make_transform_iterator(tiles_touched, ToI64()), offsets,
          (size_t)n, rocprim::plus<int64_t>(), st) != hipSuccess)
    return check_launch("rocprim inclusive_scan");
  const unsigned nb =
      (unsigned)((n + 256 * BIN_PER_THREAD - 1) / (256 * BIN_PER_THREAD));
  if (nt <= BIN_LDS_TILES)
    hipLaunchKernelGGL(gs_tile_count_kernel<true>, dim3(nb), dim3(256), 0, st,
                       n, nt, rect, gx, offsets, key_capacity, tile_count);
  else
    hipLaunchKernelGGL(gs_tile_count_kernel<false>, dim3(nb), dim3(256), 0,
                       st, n, nt, rect, gx, offsets, key_capacity, tile_count);
  hipLaunchKernelGGL(gs_tile_scan_kernel, dim3(1), dim3(1024), 0, st, nt,
                     key_capacity, tile_count, tile_start, ranges);
  if (nt <= BIN_LDS_TILES)
    hipLaunchKernelGGL(gs_tile_fill_kernel<true>, dim3(nb), dim3(256), 0, st,
                       n, nt, rect, depths, gx, key_capacity, tile_start,
                       offsets, tile_count, keys, gid_pre,
                       key_pos ? key_pos + key_capacity : nullptr, n_keys);
  else
    hipLaunchKernelGGL(gs_tile_fill_kernel<false>, dim3(nb), dim3(256), 0, st,
                       n, nt, rect, depths, gx, key_capacity, tile_start,
                       offsets, tile_count, keys, gid_pre,
                       key_pos ? key_pos + key_capacity : nullptr, n_keys);
  hipLaunchKernelGGL(gs_tile_sort_kernel, dim3(nt), dim3(SORT_THREADS), 0, st,
                     ranges, keys, gid_pre, point_list, key_pos);
  return check_launch("xrd_gs_bin");
}

int xrd_gs_bin(int n, int image_width, int image_height, const int32_t* rect,
               const int32_t* tiles_touched, const float* depths,
               int64_t key_capacity, void* workspace, int32_t* point_list,
               int32_t* ranges, int64_t* n_keys, xrd_stream_t stream) {
  return bin_impl(n, image_width, image_height, rect, tiles_touched, depths,
                  key_capacity, workspace, point_list, ranges, n_keys,
                  nullptr, nullptr, stream);
}

int xrd_gs_bin2(int n, int image_width, int image_height, const int32_t* rect,
                const int32_t* tiles_touched, const float* depths,
                int64_t key_capacity, void* workspace, int32_t* point_list,
                int32_t* ranges, int64_t* n_keys, int32_t* key_pos,
                int64_t* offsets, xrd_stream_t stream) {
  if (n > 0 && (!key_pos || !offsets)) return XRD_ERR_ARG;
  return bin_impl(n, image_width, image_height, rect, tiles_touched, depths,
                  key_capacity, workspace, point_list, ranges, n_keys, key_pos,
                  offsets, stream);
}

}  // extern "C"
