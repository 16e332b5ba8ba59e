// SplaTAM rasteriser, tile binning on the device (the [ext] rasteriser's
// BinningState: prefix sum of tiles_touched, key duplication, 64-bit radix
// sort, per-tile ranges — call sites of the reference:
// slam/model_components/gaussian_cloud_splatam.py:63-69,267-268).  Round 1 did
// this with torch.cumsum + `.item()` + torch.sort + a gather, i.e. a host
// sync in every raster pass.  Here the pass is one C call on the stream:
//   inclusive scan (rocPRIM) -> duplicate keys into a STATIC-capacity array
//   (unused slots carry the all-ones key) -> radix sort of the capacity
//   (rocPRIM, tile bits + 32 depth bits only) -> ranges.
// The true number of (Gaussian, tile) pairs is left in a device scalar; the
// caller sizes the capacity from the previous pass's count, read back
// asynchronously, and is told when a pass did not fit.
// The sort carries the PRE-SORT index of a key (a Gaussian's keys are
// contiguous before the sort): the sorted list of Gaussian ids and the
// inverse map "pre-sort key -> sorted position" follow from one small launch.
// The inverse map lets the blend backward write one gradient row per key
// and a per-Gaussian launch sum its rows without atomics (gs_raster.hip).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace xrd {
namespace {

constexpr int TILE = 16;
constexpr uint64_t kEmptyKey = ~0ull;

__global__ __launch_bounds__(256) void gs_fill_keys_kernel(
    int64_t cap, uint64_t* __restrict__ keys, int* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) {
    keys[i] = kEmptyKey;
    vals[i] = 0;
  }
}

// keys of Gaussian i go to [offsets[i-1], offsets[i]); pairs beyond the
// capacity are dropped (the total is reported)
__global__ __launch_bounds__(256) void gs_duplicate_cap_kernel(
    int n, const int* __restrict__ rect, const int64_t* __restrict__ offsets,
    const float* __restrict__ depths, int grid_x, int64_t cap,
    uint64_t* __restrict__ keys, int* __restrict__ values,
    int* __restrict__ gid_pre, int64_t* __restrict__ total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == n - 1) *total = offsets[n - 1];
  const int x0 = rect[i * 4], y0 = rect[i * 4 + 1], x1 = rect[i * 4 + 2],
            y1 = rect[i * 4 + 3];
  if ((x1 - x0) * (y1 - y0) == 0) return;
  int64_t off = (i == 0) ? 0 : offsets[i - 1];
  const uint32_t dbits = __float_as_uint(depths[i]);
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      if (off < cap) {
        keys[off] = ((uint64_t)(uint32_t)(y * grid_x + x) << 32) |
                    (uint64_t)dbits;
        values[off] = (int)off;   // pre-sort index
        gid_pre[off] = i;
      }
      ++off;
    }
}

__global__ __launch_bounds__(256) void gs_ranges_cap_kernel(
    int64_t cap, int n_tiles, const uint64_t* __restrict__ keys,
    int* __restrict__ ranges) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  const uint32_t tile = (uint32_t)(keys[i] >> 32);
  if (tile >= (uint32_t)n_tiles) return;   // unused slot
  if (i == 0 || (uint32_t)(keys[i - 1] >> 32) != tile)
    ranges[tile * 2] = (int)i;
  if (i == cap - 1 || (uint32_t)(keys[i + 1] >> 32) != tile)
    ranges[tile * 2 + 1] = (int)(i + 1);
}

// sorted position p holds pre-sort key vals[p]: Gaussian id list + inverse map
__global__ __launch_bounds__(256) void gs_unpermute_kernel(
    int64_t cap, const int64_t* __restrict__ total,
    const int* __restrict__ vals_sorted, const int* __restrict__ gid_pre,
    int* __restrict__ point_list, int* __restrict__ key_pos) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= cap) return;
  const int64_t live = total[0] < cap ? total[0] : cap;
  if (p >= live) {
    point_list[p] = 0;
    return;
  }
  const int pre = vals_sorted[p];
  point_list[p] = gid_pre[pre];
  if (key_pos) key_pos[pre] = (int)p;
}

struct BinLayout {
  size_t offsets, keys_in, vals_in, keys_out, vals_out, gid_pre, temp, total;
};

struct ToI64 {
  __host__ __device__ int64_t operator()(int v) const { return (int64_t)v; }
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int tile_bits(int n_tiles) {
  int b = 1;
  while ((1 << b) < n_tiles + 1) ++b;   // +1: the unused-slot key sorts last
  return b;
}

BinLayout layout(int n, int64_t cap, int end_bit) {
  BinLayout L;
  size_t scan = 0, sort = 0;
  (void)rocprim::inclusive_scan(
      nullptr, scan,
      rocprim::make_transform_iterator((const int*)nullptr, ToI64()),
      (int64_t*)nullptr, (size_t)n, rocprim::plus<int64_t>());
  (void)rocprim::radix_sort_pairs(
      nullptr, sort, (const uint64_t*)nullptr, (uint64_t*)nullptr,
      (const int*)nullptr, (int*)nullptr, (size_t)cap, 0u,
      (unsigned)end_bit);
  size_t at = 0;
  L.offsets = at;
  at += align256((size_t)n * sizeof(int64_t));
  L.keys_in = at;
  at += align256((size_t)cap * sizeof(uint64_t));
  L.vals_in = at;
  at += align256((size_t)cap * sizeof(int));
  L.keys_out = at;
  at += align256((size_t)cap * sizeof(uint64_t));
  L.vals_out = at;
  at += align256((size_t)cap * sizeof(int));
  L.gid_pre = at;
  at += align256((size_t)cap * sizeof(int));
  L.temp = at;
  at += align256(scan > sort ? scan : sort);
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
  return (int64_t)layout(n, key_capacity, 32 + tile_bits(nt)).total;
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
  // zero-fill as kernels (common.h: memset nodes in captured graphs)
  int rc = zero_floats(reinterpret_cast<float*>(ranges), (size_t)nt * 2, st);
  if (rc == XRD_OK) rc = zero_floats(reinterpret_cast<float*>(n_keys), 2, st);
  if (rc != XRD_OK) return rc;
  if (n == 0) return XRD_OK;
  if (!rect || !tiles_touched || !depths || !workspace || !point_list)
    return XRD_ERR_ARG;
  const int end_bit = 32 + tile_bits(nt);
  const BinLayout L = layout(n, key_capacity, end_bit);
  char* ws = static_cast<char*>(workspace);
  int64_t* offsets = offsets_out ? offsets_out
                                 : reinterpret_cast<int64_t*>(ws + L.offsets);
  uint64_t* keys_in = reinterpret_cast<uint64_t*>(ws + L.keys_in);
  int* vals_in = reinterpret_cast<int*>(ws + L.vals_in);
  uint64_t* keys_out = reinterpret_cast<uint64_t*>(ws + L.keys_out);
  int* vals_out = reinterpret_cast<int*>(ws + L.vals_out);
  int* gid_pre = reinterpret_cast<int*>(ws + L.gid_pre);
  size_t temp_bytes = L.total - L.temp;
  if (rocprim::inclusive_scan(
          ws + L.temp, temp_bytes,
          rocprim::make_transform_iterator(tiles_touched, ToI64()), offsets,
          (size_t)n, rocprim::plus<int64_t>(), st) != hipSuccess)
    return check_launch("rocprim inclusive_scan");
  const unsigned cb = (unsigned)((key_capacity + 255) / 256);
  hipLaunchKernelGGL(gs_fill_keys_kernel, dim3(cb), dim3(256), 0, st,
                     key_capacity, keys_in, vals_in);
  hipLaunchKernelGGL(gs_duplicate_cap_kernel, dim3((n + 255) / 256), dim3(256),
                     0, st, n, rect, offsets, depths, gx, key_capacity, keys_in,
                     vals_in, gid_pre, n_keys);
  temp_bytes = L.total - L.temp;
  if (rocprim::radix_sort_pairs(ws + L.temp, temp_bytes, keys_in, keys_out,
                                vals_in, vals_out, (size_t)key_capacity, 0u,
                                (unsigned)end_bit, st) != hipSuccess)
    return check_launch("rocprim radix_sort_pairs");
  hipLaunchKernelGGL(gs_unpermute_kernel, dim3(cb), dim3(256), 0, st,
                     key_capacity, n_keys, vals_out, gid_pre, point_list,
                     key_pos);
  hipLaunchKernelGGL(gs_ranges_cap_kernel, dim3(cb), dim3(256), 0, st,
                     key_capacity, nt, keys_out, ranges);
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
