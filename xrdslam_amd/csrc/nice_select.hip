// NICE-SLAM frustum feature selection on the device (gfx950): which cells of a
// feature grid the mapper optimises for the current frame.
//
// Reference behaviour restated (never copied): ConvOnet.pre_precessing /
// get_mask_from_c2w (slam/models/conv_onet.py:94-130,
// slam/model_components/utils.py:298-375): the lattice points of the grid are
// projected into the current depth image (cv2.remap, bilinear, zero border);
// a point is selected when it projects inside the image and lies in front of
// the camera no deeper than the sampled depth + 0.5 m (pixels without depth
// count as the image's largest sampled depth), or when it lies within 0.5 m of
// the camera centre.  The reference does this with numpy + cv2 on the host
// and a boolean-mask gather / scatter of the whole grid per iteration; round
// 2 did it with ~25 torch launches per grid, a 4x4 inverse through rocSOLVER
// and a nonzero() host sync per grid.  Here: two launches per mapping call
// for ALL grids — (1) sampled depth of every lattice point + their maximum,
// (2) the byte mask the render backward reads and the list of selected cells
// + its length the fused Adam reads (wave-aggregated append; the list's order
// is irrelevant: every cell's Adam state is reset per mapping call).
#include <hip/hip_runtime.h>

#include "common.h"

namespace xrd {
namespace {

constexpr int kMaxSelGrids = 4;

struct SelGrids {
  int n_grids;
  int dims[kMaxSelGrids][3];           // Z, Y, X
  const float* axis[kMaxSelGrids][3];  // x[X], y[Y], z[Z] lattice coordinates
  float* sampled[kMaxSelGrids];        // [Z*Y*X] bilinear depth of the point
  uint8_t* mask[kMaxSelGrids];         // [Z*Y*X]
  int32_t* cells[kMaxSelGrids];        // capacity Z*Y*X
  int32_t* count[kMaxSelGrids];        // [1]
};

struct SelCam {
  int H, W;
  float fx, fy, cx, cy;
};

// world -> camera of a 4x4 pose with bottom row (0,0,0,1), in double
__device__ __forceinline__ void affine_inverse(const float* __restrict__ c2w,
                                               double (&R)[3][3],
                                               double (&t)[3]) {
  double a[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a[i][j] = (double)c2w[i * 4 + j];
  const double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1];
  const double c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2];
  const double c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
  const double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
  const double id = 1.0 / det;
  R[0][0] = c00 * id;
  R[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id;
  R[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id;
  R[1][0] = c01 * id;
  R[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id;
  R[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id;
  R[2][0] = c02 * id;
  R[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id;
  R[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id;
  for (int i = 0; i < 3; ++i)
    t[i] = -(R[i][0] * (double)c2w[3] + R[i][1] * (double)c2w[7] +
             R[i][2] * (double)c2w[11]);
}

struct Proj {
  float u, v, negz;  // pixel coordinates, -z (depth along the view axis)
  bool near;         // within 0.5 m of the camera centre
};

// lattice point i of a grid (numpy meshgrid 'ij' order over x, y, z: the
// reference's flat index, utils.py:316-325) -> its projection
__device__ __forceinline__ Proj project(const SelGrids& g, int gi, int64_t i,
                                        const double (&R)[3][3],
                                        const double (&t)[3],
                                        const float* __restrict__ c2w,
                                        const SelCam& cam, int64_t& cell) {
  const int Z = g.dims[gi][0], Y = g.dims[gi][1];
  const int iz = (int)(i % Z), iy = (int)((i / Z) % Y),
            ix = (int)(i / ((int64_t)Z * Y));
  cell = ((int64_t)iz * Y + iy) * g.dims[gi][2] + ix;  // [Z][Y][X] storage
  const double p[3] = {(double)g.axis[gi][0][ix], (double)g.axis[gi][1][iy],
                       (double)g.axis[gi][2][iz]};
  double c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
    c[a] = p[0] * R[a][0] + p[1] * R[a][1] + p[2] * R[a][2] + t[a];
  const double un = (double)cam.fx * (-c[0]) + (double)cam.cx * c[2];
  const double vn = (double)cam.fy * c[1] + (double)cam.cy * c[2];
  const double z = c[2] + 1e-5;
  Proj o;
  o.u = (float)(un / z);
  o.v = (float)(vn / z);
  o.negz = (float)(-z);
  double d2 = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double d = p[a] - (double)c2w[a * 4 + 3];
    d2 += d * d;
  }
  o.near = d2 < 0.25;
  return o;
}

__device__ __forceinline__ float tap(const float* __restrict__ img, int H,
                                     int W, float uu, float vv) {
  if (!(uu >= 0.f && uu <= (float)(W - 1) && vv >= 0.f &&
        vv <= (float)(H - 1)))
    return 0.f;
  return img[(int64_t)(int)vv * W + (int)uu];
}

// bilinear sample with a zero border (cv2.remap INTER_LINEAR, BORDER_CONSTANT)
__device__ __forceinline__ float bilinear(const float* __restrict__ img, int H,
                                          int W, float u, float v) {
  const float u0 = floorf(u), v0 = floorf(v);
  const float fu = u - u0, fv = v - v0;
  const float a = tap(img, H, W, u0, v0), b = tap(img, H, W, u0 + 1.f, v0),
              c = tap(img, H, W, u0, v0 + 1.f),
              d = tap(img, H, W, u0 + 1.f, v0 + 1.f);
  const float gu = 1.f - fu, gv = 1.f - fv;
  float s = __fmul_rn(__fmul_rn(a, gu), gv);
  s = __fadd_rn(s, __fmul_rn(__fmul_rn(b, fu), gv));
  s = __fadd_rn(s, __fmul_rn(__fmul_rn(c, gu), fv));
  s = __fadd_rn(s, __fmul_rn(__fmul_rn(d, fu), fv));
  return s;
}

__global__ __launch_bounds__(256) void frustum_depth_kernel(
    SelGrids g, SelCam cam, const float* __restrict__ c2w,
    const float* __restrict__ depth, int* __restrict__ dmax_bits) {
  __shared__ double sR[3][3], st[3];
  __shared__ int smax;
  const int gi = blockIdx.y;
  if (threadIdx.x == 0) {
    double R[3][3], t[3];
    affine_inverse(c2w, R, t);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) sR[i][j] = R[i][j];
      st[i] = t[i];
    }
    smax = 0;
  }
  __syncthreads();
  double R[3][3], t[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[i][j] = sR[i][j];
    t[i] = st[i];
  }
  const int64_t n =
      (int64_t)g.dims[gi][0] * g.dims[gi][1] * g.dims[gi][2];
  int local = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    int64_t cell;
    const Proj pr = project(g, gi, i, R, t, c2w, cam, cell);
    float s = bilinear(depth, cam.H, cam.W, pr.u, pr.v);
    if (!(s == s)) s = 0.f;  // NaN coordinates sample nothing
    g.sampled[gi][i] = s;
    // sampled depths are >= 0: their float order is their integer order
    local = max(local, __float_as_int(fmaxf(s, 0.f)));
  }
  atomicMax(&smax, local);
  __syncthreads();
  if (threadIdx.x == 0 && smax > 0) atomicMax(dmax_bits + gi, smax);
}

__global__ __launch_bounds__(256) void frustum_select_kernel(
    SelGrids g, SelCam cam, const float* __restrict__ c2w,
    const int* __restrict__ dmax_bits) {
  __shared__ double sR[3][3], st[3];
  const int gi = blockIdx.y;
  if (threadIdx.x == 0) {
    double R[3][3], t[3];
    affine_inverse(c2w, R, t);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) sR[i][j] = R[i][j];
      st[i] = t[i];
    }
  }
  __syncthreads();
  double R[3][3], t[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[i][j] = sR[i][j];
    t[i] = st[i];
  }
  const float dmax = __int_as_float(dmax_bits[gi]);
  const int64_t n =
      (int64_t)g.dims[gi][0] * g.dims[gi][1] * g.dims[gi][2];
  const int lane = threadIdx.x & 63;
  // whole waves iterate together (the append is wave-aggregated)
  const int64_t n_up = (n + 63) / 64 * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_up;
       i += (int64_t)gridDim.x * 256) {
    bool sel = false;
    int64_t cell = 0;
    if (i < n) {
      const Proj pr = project(g, gi, i, R, t, c2w, cam, cell);
      float d = g.sampled[gi][i];
      if (d == 0.f) d = dmax;
      sel = pr.u < (float)cam.W && pr.u > 0.f && pr.v < (float)cam.H &&
            pr.v > 0.f && pr.negz >= 0.f && pr.negz <= d + 0.5f;
      sel = sel || pr.near;
      g.mask[gi][cell] = sel ? 1 : 0;
    }
    const uint64_t b = __ballot(sel);
    if (b != 0) {
      int base = 0;
      if (lane == 0) base = atomicAdd(g.count[gi], __popcll(b));
      base = __shfl(base, 0);
      if (sel)
        g.cells[gi][base + __popcll(b & ((1ull << lane) - 1ull))] = (int)cell;
    }
  }
}

__global__ void frustum_reset_kernel(SelGrids g, int* dmax_bits) {
  if ((int)threadIdx.x < g.n_grids) {
    g.count[threadIdx.x][0] = 0;
    dmax_bits[threadIdx.x] = 0;
  }
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" int xrd_nice_frustum_cells(
    int n_grids, const int32_t* dims_zyx, const float* const* axes,
    const float* c2w, const float* depth, int H, int W, float fx, float fy,
    float cx, float cy, float* const* sampled, uint8_t* const* mask,
    int32_t* const* cells, int32_t* const* count, int32_t* ws,
    xrd_stream_t stream) {
  if (n_grids < 0 || n_grids > kMaxSelGrids || H < 1 || W < 1)
    return XRD_ERR_ARG;
  if (n_grids == 0) return XRD_OK;
  if (!dims_zyx || !axes || !c2w || !depth || !sampled || !mask || !cells ||
      !count || !ws)
    return XRD_ERR_ARG;
  SelGrids g = {};
  g.n_grids = n_grids;
  int64_t largest = 0;
  for (int i = 0; i < n_grids; ++i) {
    for (int a = 0; a < 3; ++a) {
      g.dims[i][a] = dims_zyx[i * 3 + a];
      g.axis[i][a] = axes[i * 3 + a];
      if (g.dims[i][a] < 1 || g.axis[i][a] == nullptr) return XRD_ERR_ARG;
    }
    g.sampled[i] = sampled[i];
    g.mask[i] = mask[i];
    g.cells[i] = cells[i];
    g.count[i] = count[i];
    if (!g.sampled[i] || !g.mask[i] || !g.cells[i] || !g.count[i])
      return XRD_ERR_ARG;
    const int64_t n = (int64_t)g.dims[i][0] * g.dims[i][1] * g.dims[i][2];
    if (n > largest) largest = n;
  }
  const SelCam cam = {H, W, fx, fy, cx, cy};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(frustum_reset_kernel, dim3(1), dim3(64), 0, st, g, ws);
  int64_t blocks = (largest + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const dim3 grid((unsigned)blocks, (unsigned)n_grids);
  hipLaunchKernelGGL(frustum_depth_kernel, grid, dim3(256), 0, st, g, cam, c2w,
                     depth, ws);
  hipLaunchKernelGGL(frustum_select_kernel, grid, dim3(256), 0, st, g, cam,
                     c2w, ws);
  return check_launch("xrd_nice_frustum_cells");
}
