// Map maintenance between iterations as kernels: the
// row surgery and selection steps the reference runs as chains of boolean-mask
// indexing (one compaction + one size read-back per tensor), numpy on the host
// or torch.unique.  All of it is HBM-bound byte / index work.
//
//   xrd_compact_rows          stable compaction of up to 24 row arrays by ONE
//                             keep mask (SplaTAM remove_points /
//                             get_pointcloud(mask), gaussian_cloud_splatam.py:
//                             84-111,355-399; Vox-Fusion voxel rows)
//   xrd_voxel_first_flags     first occurrence of every distinct voxel row
//                             (the octree creates a node for the first
//                             occurrence only: sparse_voxel.py:333-340 +
//                             octree.cpp insert order)
//   xrd_point_dynamic_radius  Point-SLAM's per-pixel add / query radii from the
//                             colour-gradient magnitude (point_slam.py:326-354:
//                             rgb2gray, Sobel, clip, two np.interp) in f64
//   xrd_point_sensor_points   o + d * depth
//   xrd_point_insert          Point-SLAM add_neural_points' selection +
//                             placement (neural_point_cloud.py:109-221)
//   xrd_point_frustum_mask    Point-SLAM get_mask_from_c2w (point_slam.py:356-420)
//
// Every arithmetic expression keeps the reference's operation order with no
// contraction, so results are the torch / numpy formulation's bit for bit
// (tests/test_map_ops_hip.py).
#include <climits>

#include "common.h"

#pragma clang fp contract(off)

namespace xrd {
namespace {

// ------------------------------------------------------------ row compaction
constexpr int kMaxArrays = 24;
constexpr int kChunk = 1024;     // rows per block

struct RowArrays {
  const uint32_t* src[kMaxArrays];
  uint32_t* dst[kMaxArrays];
  int words[kMaxArrays];
  int n_arrays;
};

__global__ __launch_bounds__(kChunk) void compact_count_kernel(
    int64_t n, const uint8_t* __restrict__ keep, int* __restrict__ block_count) {
  __shared__ int wsum[kChunk / 64];
  const int64_t i = (int64_t)blockIdx.x * kChunk + threadIdx.x;
  const bool k = i < n && keep[i] != 0;
  const unsigned long long b = __ballot(k);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kChunk / 64; ++w) s += wsum[w];
    block_count[blockIdx.x] = s;
  }
}

// exclusive scan of the block counts in place (one block); total -> count
__global__ __launch_bounds__(1024) void compact_scan_kernel(
    int n_blocks, int* __restrict__ block_count, int* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n_blocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n_blocks ? block_count[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(inc, o);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int carry = carry_s;
    if (i < n_blocks) block_count[i] = carry + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = carry_s;
}

// a block moves the kept rows of its chunk: ranks from ballots, the kept local
// row numbers listed in LDS, then every array copied word by word with
// consecutive threads on consecutive destination words
__global__ __launch_bounds__(kChunk) void compact_move_kernel(
    int64_t n, const uint8_t* __restrict__ keep,
    const int* __restrict__ block_off, RowArrays a) {
  __shared__ int wsum[kChunk / 64];
  __shared__ int rows[kChunk];
  const int64_t base = (int64_t)blockIdx.x * kChunk;
  const int64_t i = base + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool k = i < n && keep[i] != 0;
  const unsigned long long b = __ballot(k);
  if (lane == 0) wsum[wave] = __popcll(b);
  __syncthreads();
  int woff = 0, total = 0;
  for (int w = 0; w < kChunk / 64; ++w) {
    const int c = wsum[w];
    if (w < wave) woff += c;
    total += c;
  }
  if (k) rows[woff + __popcll(b & ((1ull << lane) - 1ull))] = threadIdx.x;
  __syncthreads();
  const int64_t off = block_off[blockIdx.x];
  for (int j = 0; j < a.n_arrays; ++j) {
    const int w = a.words[j];
    const uint32_t* __restrict__ src = a.src[j] + base * w;
    uint32_t* __restrict__ dst = a.dst[j] + off * w;
    const int words = total * w;
    if (w == 1) {
      for (int e = threadIdx.x; e < words; e += kChunk) dst[e] = src[rows[e]];
    } else {
      for (int e = threadIdx.x; e < words; e += kChunk) {
        const int r = e / w, c = e - r * w;
        dst[e] = src[(int64_t)rows[r] * w + c];
      }
    }
  }
}

// ------------------------------------------- first occurrence of voxel rows
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kCoordBias = 1 << 20;      // 21 bits a coordinate

__device__ __forceinline__ unsigned long long voxel_key(
    const int32_t* __restrict__ v, int64_t i, int* __restrict__ err) {
  const int x = v[i * 3] + kCoordBias, y = v[i * 3 + 1] + kCoordBias,
            z = v[i * 3 + 2] + kCoordBias;
  if (((unsigned)x | (unsigned)y | (unsigned)z) >> 21) *err = 1;
  return ((unsigned long long)(x & 0x1fffff) << 42) |
         ((unsigned long long)(y & 0x1fffff) << 21) |
         (unsigned long long)(z & 0x1fffff);
}
__device__ __forceinline__ uint32_t key_slot(unsigned long long k,
                                             uint32_t mask) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k & mask;
}

__global__ void voxel_table_reset_kernel(int64_t size,
                                         unsigned long long* __restrict__ keys,
                                         int* __restrict__ first_row,
                                         int* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < size) {
    keys[i] = kEmptyKey;
    first_row[i] = INT_MAX;
  }
  if (i == 0) *err = 0;
}

// neighbouring depth pixels mostly fall into the same voxel: only the first
// row of a run of equal keys inside a wave goes to the table (rows of a run
// after the first can never be a first occurrence)
__device__ __forceinline__ bool run_head(unsigned long long key, bool live) {
  const unsigned long long prev = __shfl_up(key, 1);
  const int prev_live = __shfl_up((int)live, 1);
  return live && ((threadIdx.x & 63) == 0 || !prev_live || prev != key);
}

__global__ __launch_bounds__(256) void voxel_insert_kernel(
    int64_t n, const int32_t* __restrict__ voxels, uint32_t mask,
    unsigned long long* __restrict__ keys, int* __restrict__ first_row,
    int* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const unsigned long long key = live ? voxel_key(voxels, i, err) : 0ull;
  if (!run_head(key, live)) return;
  uint32_t h = key_slot(key, mask);
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    const unsigned long long prev = atomicCAS(keys + h, kEmptyKey, key);
    if (prev == kEmptyKey || prev == key) {
      atomicMin(first_row + h, (int)i);
      return;
    }
    h = (h + 1) & mask;
  }
  *err = 2;     // table full
}

__global__ __launch_bounds__(256) void voxel_flag_kernel(
    int64_t n, const int32_t* __restrict__ voxels, uint32_t mask,
    const unsigned long long* __restrict__ keys,
    const int* __restrict__ first_row, uint8_t* __restrict__ first,
    int* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  int dummy = 0;
  const unsigned long long key = live ? voxel_key(voxels, i, &dummy) : 0ull;
  const bool head = run_head(key, live);
  if (!live) return;
  uint8_t f = 0;
  if (head) {
    uint32_t h = key_slot(key, mask);
    for (uint32_t probe = 0; probe <= mask; ++probe) {
      const unsigned long long k = keys[h];
      if (k == key) {
        f = first_row[h] == (int)i;
        break;
      }
      if (k == kEmptyKey) {
        *err = 3;   // cannot happen: every head was inserted
        break;
      }
      h = (h + 1) & mask;
    }
  }
  first[i] = f;
}

// -------------------------------------------------- Point-SLAM dynamic radii
struct RadiusCfg {
  double thresh;        // clip limit = last knot
  double knot1;         // 0.01
  double add_max, add_min, add_slope0, add_slope1;
  double qry_max, qry_min, qry_slope0, qry_slope1;
};

__device__ __forceinline__ int reflect(int i, int n) {   // np.pad 'symmetric'
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
// skimage's luma weights applied like numpy applies them to a float32 image:
// float32 products and sums, left to right
__device__ __forceinline__ double luma(const float* __restrict__ rgb, int y,
                                       int x, int W) {
  const float* p = rgb + ((int64_t)y * W + x) * 3;
  const float a = p[0] * 0.2125f;
  const float b = p[1] * 0.7154f;
  const float c = p[2] * 0.0721f;
  const float ab = a + b;
  return (double)(ab + c);
}
// np.interp over the knots (0, knot1, thresh) for 0 <= x <= thresh
__device__ __forceinline__ double interp3(double x, double knot1,
                                          double thresh, double y0, double y2,
                                          double s0, double s1) {
  if (x == thresh) return y2;                 // last knot: its value
  if (x < knot1) {
    if (x == 0.0) return y0;
    const double t = s0 * (x - 0.0);
    return t + y0;
  }
  if (x == knot1) return y0;                  // (y1 == y0)
  const double t = s1 * (x - knot1);
  return t + y0;
}

__global__ __launch_bounds__(256) void point_radius_kernel(
    int H, int W, const float* __restrict__ rgb, RadiusCfg c,
    double* __restrict__ r_add, double* __restrict__ r_query) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  double g[3][3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
      g[dy][dx] = luma(rgb, reflect(y + dy - 1, H), reflect(x + dx - 1, W), W);
  // sobel_v (difference along x, smoothed along y) and sobel_h
  const double v0 = g[0][0] - g[0][2], v1 = g[1][0] - g[1][2],
               v2 = g[2][0] - g[2][2];
  const double sv = ((v0 + 2.0 * v1) + v2) / 4.0;
  const double h0 = g[0][0] - g[2][0], h1 = g[0][1] - g[2][1],
               h2 = g[0][2] - g[2][2];
  const double sh = ((h0 + 2.0 * h1) + h2) / 4.0;
  const double sv2 = sv * sv, sh2 = sh * sh;
  double mag = sqrt(sv2 + sh2);
  mag = mag < 0.0 ? 0.0 : (mag > c.thresh ? c.thresh : mag);
  const int64_t i = (int64_t)y * W + x;
  r_add[i] = interp3(mag, c.knot1, c.thresh, c.add_max, c.add_min,
                     c.add_slope0, c.add_slope1);
  r_query[i] = interp3(mag, c.knot1, c.thresh, c.qry_max, c.qry_min,
                       c.qry_slope0, c.qry_slope1);
}

// ------------------------------------------------------ Point-SLAM insertion
__global__ void point_sensor_points_kernel(int64_t n,
                                           const float* __restrict__ o,
                                           const float* __restrict__ d,
                                           const float* __restrict__ depth,
                                           float* __restrict__ pts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 3) return;
  const float t = d[i] * depth[i / 3];
  pts[i] = o[i] + t;
}

struct InsertCfg {
  int n_add;
  int fix_interval;
  float near_end, far_end;
};
constexpr int kMaxAdd = 16;

// ONE block walks the rays in chunks of 1024 with a running output offset
// (<= ~2e4 rays per call): keep = depth > 0 and no neural point within the add
// radius; kept rays append their sensor point, their colour * 255 and n_add
// points along the ray, in ray order
__global__ __launch_bounds__(1024) void point_insert_kernel(
    int n, const float* __restrict__ o, const float* __restrict__ d,
    const float* __restrict__ depth, const float* __restrict__ color,
    const float* __restrict__ pts_gt, const int* __restrict__ n_within,
    const float* __restrict__ lin, InsertCfg c, float* __restrict__ out_pos,
    float* __restrict__ out_rgb, float* __restrict__ out_pts,
    int* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    bool k = false;
    float dep = 0.f;
    if (i < n) {
      dep = depth[i];
      k = dep > 0.f && (n_within == nullptr || n_within[i] == 0);
    }
    const unsigned long long b = __ballot(k);
    if (lane == 0) wsum[wave] = __popcll(b);
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      const int cw = wsum[w];
      if (w < wave) woff += cw;
      total += cw;
    }
    const int carry = carry_s;
    if (k) {
      const int64_t r = carry + woff + __popcll(b & ((1ull << lane) - 1ull));
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        out_pos[r * 3 + a] = pts_gt[(int64_t)i * 3 + a];
        out_rgb[r * 3 + a] = color[(int64_t)i * 3 + a] * 255.f;
      }
      for (int j = 0; j < c.n_add; ++j) {
        float z;
        if (c.fix_interval) {
          z = dep + lin[j];
        } else {
          const float t = lin[j];
          const float near = (c.near_end * dep) * (1.f - t);
          const float far = (c.far_end * dep) * t;
          z = near + far;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float s = d[(int64_t)i * 3 + a] * z;
          out_pts[(r * c.n_add + j) * 3 + a] = o[(int64_t)i * 3 + a] + s;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = carry_s;
}

// ---------------------------------------------------- Point-SLAM frustum mask
struct MaskCam {
  int H, W, edge;
  double fx, fy, cx, cy;
};

__device__ __forceinline__ float depth_tap(const float* __restrict__ img,
                                           float uu, float vv, int H, int W) {
  const bool ok = uu >= 0.f && uu <= (float)(W - 1) && vv >= 0.f &&
                  vv <= (float)(H - 1);
  return ok ? img[(int64_t)(int)vv * W + (int)uu] : 0.f;
}

// pass 1: projected pixel, bilinear depth (zero border), its maximum
__global__ __launch_bounds__(256) void point_mask_depth_kernel(
    int64_t n, const float* __restrict__ pts, const double* __restrict__ w2c,
    const float* __restrict__ depth, MaskCam cam, float* __restrict__ uvzd,
    double* __restrict__ zs, int* __restrict__ dmax_bits) {
  __shared__ float red[4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float dval = 0.f;
  if (i < n) {
    const double x = pts[i * 3], y = pts[i * 3 + 1], z3 = pts[i * 3 + 2];
    double pc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double a = x * w2c[r * 4], b = y * w2c[r * 4 + 1],
                   c = z3 * w2c[r * 4 + 2];
      pc[r] = ((a + b) + c) + w2c[r * 4 + 3];
    }
    const double un = cam.fx * (-pc[0]) + cam.cx * pc[2];
    const double vn = cam.fy * pc[1] + cam.cy * pc[2];
    const double z = pc[2] + 1e-5;
    const float u = (float)(un / z), v = (float)(vn / z);
    const float u0 = floorf(u), v0 = floorf(v);
    const float fu = u - u0, fv = v - v0;
    const float gu = 1.f - fu, gv = 1.f - fv;
    const float t00 = depth_tap(depth, u0, v0, cam.H, cam.W);
    const float t10 = depth_tap(depth, u0 + 1.f, v0, cam.H, cam.W);
    const float t01 = depth_tap(depth, u0, v0 + 1.f, cam.H, cam.W);
    const float t11 = depth_tap(depth, u0 + 1.f, v0 + 1.f, cam.H, cam.W);
    const float a = (t00 * gu) * gv, b = (t10 * fu) * gv, c = (t01 * gu) * fv,
                e = (t11 * fu) * fv;
    dval = ((a + b) + c) + e;
    uvzd[i * 3] = u;
    uvzd[i * 3 + 1] = v;
    uvzd[i * 3 + 2] = dval;
    zs[i] = z;
  }
  float m = dval > 0.f ? dval : 0.f;      // NaN-free maximum of the block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f) atomicMax(dmax_bits, __float_as_int(m));
  }
}

// pass 2: inside the image minus the edge, in front of the camera, not behind
// the measured depth + 0.5 m (pixels without depth take the maximum)
__global__ __launch_bounds__(256) void point_mask_final_kernel(
    int64_t n, const float* __restrict__ uvzd, const double* __restrict__ zs,
    MaskCam cam, const int* __restrict__ dmax_bits,
    uint8_t* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float u = uvzd[i * 3], v = uvzd[i * 3 + 1];
  float dval = uvzd[i * 3 + 2];
  const double z = zs[i];
  if (dval == 0.f) dval = __int_as_float(*dmax_bits);
  const float edge = (float)cam.edge;
  const bool inside = u < (float)(cam.W - cam.edge) && u > edge &&
                      v < (float)(cam.H - cam.edge) && v > edge;
  const float near = dval + 0.5f;
  mask[i] = inside && (-z >= 0.0) && ((float)(-z) <= near);
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int64_t xrd_compact_ws_ints(int64_t n) { return (n + kChunk - 1) / kChunk + 1; }

int xrd_compact_rows(int64_t n, const uint8_t* keep, int n_arrays,
                     const void* const* src, void* const* dst,
                     const int32_t* row_words, int32_t* ws, int32_t* count,
                     xrd_stream_t stream) {
  if (n < 0 || n_arrays < 0 || n_arrays > kMaxArrays || !count)
    return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return zero_floats((float*)count, 1, stream);
  if (!keep || !ws || (n_arrays && (!src || !dst || !row_words)))
    return XRD_ERR_ARG;
  if (n > (int64_t)INT_MAX) return XRD_ERR_ARG;
  RowArrays a = {};
  a.n_arrays = n_arrays;
  for (int j = 0; j < n_arrays; ++j) {
    if (!src[j] || !dst[j] || row_words[j] < 1) return XRD_ERR_ARG;
    a.src[j] = (const uint32_t*)src[j];
    a.dst[j] = (uint32_t*)dst[j];
    a.words[j] = row_words[j];
  }
  const int blocks = (int)((n + kChunk - 1) / kChunk);
  hipLaunchKernelGGL(compact_count_kernel, dim3(blocks), dim3(kChunk), 0, st,
                     n, keep, ws);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, blocks,
                     ws, count);
  if (n_arrays)
    hipLaunchKernelGGL(compact_move_kernel, dim3(blocks), dim3(kChunk), 0, st,
                       n, keep, ws, a);
  return check_launch("xrd_compact_rows");
}

int xrd_voxel_first_flags(int64_t n, const int32_t* voxels, uint8_t* first,
                          uint64_t* table_keys, int32_t* table_rows,
                          int64_t table_size, int32_t* err,
                          xrd_stream_t stream) {
  if (n < 0 || table_size < 2 || (table_size & (table_size - 1)) ||
      table_size > ((int64_t)1 << 31) || n > (int64_t)INT_MAX)
    return XRD_ERR_ARG;
  if (!table_keys || !table_rows || !err) return XRD_ERR_ARG;
  if (n && (!voxels || !first)) return XRD_ERR_ARG;
  if (table_size < 2 * n) return XRD_ERR_ARG;   // load factor <= 1/2
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(voxel_table_reset_kernel,
                     dim3((unsigned)((table_size + 255) / 256)), dim3(256), 0,
                     st, table_size, (unsigned long long*)table_keys,
                     table_rows, err);
  if (n) {
    const dim3 grid((unsigned)((n + 255) / 256));
    const uint32_t mask = (uint32_t)(table_size - 1);
    hipLaunchKernelGGL(voxel_insert_kernel, grid, dim3(256), 0, st, n, voxels,
                       mask, (unsigned long long*)table_keys, table_rows, err);
    hipLaunchKernelGGL(voxel_flag_kernel, grid, dim3(256), 0, st, n, voxels,
                       mask, (const unsigned long long*)table_keys, table_rows,
                       first, err);
  }
  return check_launch("xrd_voxel_first_flags");
}

int xrd_point_dynamic_radius(int H, int W, const float* rgb, double thresh,
                             double add_max, double add_min,
                             double query_ratio, double* r_add,
                             double* r_query, xrd_stream_t stream) {
  if (H < 1 || W < 1 || !rgb || !r_add || !r_query || !(thresh > 0.01))
    return XRD_ERR_ARG;
  RadiusCfg c;
  c.thresh = thresh;
  c.knot1 = 0.01;
  c.add_max = add_max;
  c.add_min = add_min;
  c.qry_max = query_ratio * add_max;
  c.qry_min = query_ratio * add_min;
  // np.interp's slopes: (y[i+1] - y[i]) / (x[i+1] - x[i])
  c.add_slope0 = (c.add_max - c.add_max) / (c.knot1 - 0.0);
  c.add_slope1 = (c.add_min - c.add_max) / (c.thresh - c.knot1);
  c.qry_slope0 = (c.qry_max - c.qry_max) / (c.knot1 - 0.0);
  c.qry_slope1 = (c.qry_min - c.qry_max) / (c.thresh - c.knot1);
  const dim3 grid((W + 63) / 64, (H + 3) / 4);
  hipLaunchKernelGGL(point_radius_kernel, grid, dim3(256), 0,
                     (hipStream_t)stream, H, W, rgb, c, r_add, r_query);
  return check_launch("xrd_point_dynamic_radius");
}

int xrd_point_sensor_points(int64_t n, const float* rays_o, const float* rays_d,
                            const float* depth, float* pts,
                            xrd_stream_t stream) {
  if (n < 0) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!rays_o || !rays_d || !depth || !pts) return XRD_ERR_ARG;
  hipLaunchKernelGGL(point_sensor_points_kernel,
                     dim3((unsigned)((n * 3 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, n, rays_o, rays_d, depth, pts);
  return check_launch("xrd_point_sensor_points");
}

int xrd_point_insert(int n, const float* rays_o, const float* rays_d,
                     const float* depth, const float* color,
                     const float* pts_gt, const int32_t* n_within,
                     const float* lin, int n_add, int fix_interval,
                     float near_end, float far_end, float* out_pos,
                     float* out_rgb, float* out_pts, int32_t* count,
                     xrd_stream_t stream) {
  if (n < 0 || n_add < 1 || n_add > kMaxAdd || !count) return XRD_ERR_ARG;
  if (n && (!rays_o || !rays_d || !depth || !color || !pts_gt || !lin ||
            !out_pos || !out_rgb || !out_pts))
    return XRD_ERR_ARG;
  const InsertCfg c = {n_add, fix_interval, near_end, far_end};
  hipLaunchKernelGGL(point_insert_kernel, dim3(1), dim3(1024), 0,
                     (hipStream_t)stream, n, rays_o, rays_d, depth, color,
                     pts_gt, n_within, lin, c, out_pos, out_rgb, out_pts,
                     count);
  return check_launch("xrd_point_insert");
}

int xrd_point_frustum_mask(int64_t n, const float* points, const double* w2c,
                           const float* depth, int H, int W, double fx,
                           double fy, double cx, double cy, int edge,
                           float* ws_f, double* ws_d, int32_t* ws_i,
                           uint8_t* mask, xrd_stream_t stream) {
  if (n < 0 || H < 1 || W < 1 || !ws_i) return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int rc = zero_floats((float*)ws_i, 1, stream);
  if (rc != XRD_OK || n == 0) return rc;
  if (!points || !w2c || !depth || !ws_f || !ws_d || !mask)
    return XRD_ERR_ARG;
  const MaskCam cam = {H, W, edge, fx, fy, cx, cy};
  const dim3 grid((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(point_mask_depth_kernel, grid, dim3(256), 0, st, n,
                     points, w2c, depth, cam, ws_f, ws_d, ws_i);
  hipLaunchKernelGGL(point_mask_final_kernel, grid, dim3(256), 0, st, n, ws_f,
                     ws_d, cam, ws_i, mask);
  return check_launch("xrd_point_frustum_mask");
}

}  // extern "C"
