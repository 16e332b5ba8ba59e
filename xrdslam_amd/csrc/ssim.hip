// Fused SSIM map for SplaTAM's mapping loss (0.2 * (1 - SSIM), reference:
// slam/models/gaussian_splatting.py:214-216 -> calc_ssim / _ssim,
// slam/model_components/slam_external_splatam.py:59-96): 11x11 Gaussian window
// (sigma 1.5), zero padding, per channel.  The reference runs five depth-wise
// convolutions forward and their autograd backward (MIOpen picks ~0.2 ms
// kernels per convolution at 3x480x640); here one launch computes the five
// windowed moments from an LDS tile and the SSIM value, and keeps the three
// partial derivatives the backward needs; one more launch blurs them back.
// The window is an outer product: a block filters its tile's rows once
// (26 x 16 row sums per quantity, kept in LDS) and every pixel then sums 11
// of them down its column — 22 instead of 121 taps a pixel and quantity, in
// the SAME association (row sums first, then the column) as the direct form,
// so the maps are bit-identical to round 4's.
#include "common.h"

namespace xrd {
namespace {

constexpr int kT = 16;            // pixels per block side
constexpr int kR = 5;             // window radius (11 taps)
constexpr int kE = kT + 2 * kR;   // tile edge with halo
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

struct Win {
  float g[11];
};

__device__ __forceinline__ float load0(const float* __restrict__ img, int H,
                                       int W, int y, int x) {
  return (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.f;
}

__global__ __launch_bounds__(kT* kT) void ssim_fwd_kernel(
    Win win, int H, int W, const float* __restrict__ img1,
    const float* __restrict__ img2, float* __restrict__ ssim_map,
    float* __restrict__ d_mu1, float* __restrict__ d_e11,
    float* __restrict__ d_e12) {
  __shared__ float s1[kE][kE + 1], s2[kE][kE + 1];
  __shared__ float hs[5][kE][kT + 1];
  const size_t plane = (size_t)blockIdx.z * H * W;
  const float* a = img1 + plane;
  const float* b = img2 + plane;
  const int x0 = blockIdx.x * kT - kR, y0 = blockIdx.y * kT - kR;
  for (int i = threadIdx.y * kT + threadIdx.x; i < kE * kE; i += kT * kT) {
    const int ty = i / kE, tx = i - ty * kE;
    s1[ty][tx] = load0(a, H, W, y0 + ty, x0 + tx);
    s2[ty][tx] = load0(b, H, W, y0 + ty, x0 + tx);
  }
  __syncthreads();
  // row sums of the five quantities for the tile's 26 rows x 16 columns
  for (int i = threadIdx.y * kT + threadIdx.x; i < kE * kT; i += kT * kT) {
    const int ty = i / kT, tx = i - ty * kT;
    float r1 = 0.f, r2 = 0.f, r11 = 0.f, r22 = 0.f, r12 = 0.f;
#pragma unroll
    for (int dx = 0; dx < 11; ++dx) {
      const float u = s1[ty][tx + dx];
      const float v = s2[ty][tx + dx];
      const float w = win.g[dx];
      r1 = fmaf(w, u, r1);
      r2 = fmaf(w, v, r2);
      r11 = fmaf(w, u * u, r11);
      r22 = fmaf(w, v * v, r22);
      r12 = fmaf(w, u * v, r12);
    }
    hs[0][ty][tx] = r1;
    hs[1][ty][tx] = r2;
    hs[2][ty][tx] = r11;
    hs[3][ty][tx] = r22;
    hs[4][ty][tx] = r12;
  }
  __syncthreads();
  const int x = blockIdx.x * kT + threadIdx.x, y = blockIdx.y * kT + threadIdx.y;
  if (x >= W || y >= H) return;
  float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
  for (int dy = 0; dy < 11; ++dy) {
    const float wy = win.g[dy];
    mu1 = fmaf(wy, hs[0][threadIdx.y + dy][threadIdx.x], mu1);
    mu2 = fmaf(wy, hs[1][threadIdx.y + dy][threadIdx.x], mu2);
    e11 = fmaf(wy, hs[2][threadIdx.y + dy][threadIdx.x], e11);
    e22 = fmaf(wy, hs[3][threadIdx.y + dy][threadIdx.x], e22);
    e12 = fmaf(wy, hs[4][threadIdx.y + dy][threadIdx.x], e12);
  }
  const float s1sq = e11 - mu1 * mu1, s2sq = e22 - mu2 * mu2;
  const float s12 = e12 - mu1 * mu2;
  const float A1 = 2.f * mu1 * mu2 + kC1, A2 = 2.f * s12 + kC2;
  const float B1 = mu1 * mu1 + mu2 * mu2 + kC1, B2 = s1sq + s2sq + kC2;
  const float ssim = A1 * A2 / (B1 * B2);
  const size_t o = plane + (size_t)y * W + x;
  ssim_map[o] = ssim;
  if (d_mu1 != nullptr) {
    // partials w.r.t. the window moments (mu1, E[x^2], E[xy]) of image 1
    const float ds1 = -ssim / B2;               // d/d sigma1^2 = d/d e11
    const float ds12 = 2.f * A1 / (B1 * B2);    // d/d sigma12  = d/d e12
    const float dmu = 2.f * mu2 * A2 / (B1 * B2) - ssim * 2.f * mu1 / B1;
    d_mu1[o] = dmu - 2.f * mu1 * ds1 - mu2 * ds12;
    d_e11[o] = ds1;
    d_e12[o] = ds12;
  }
}

__global__ __launch_bounds__(kT* kT) void ssim_bwd_kernel(
    Win win, int H, int W, const float* __restrict__ img1,
    const float* __restrict__ img2, const float* __restrict__ g_map,
    const float* __restrict__ d_mu1, const float* __restrict__ d_e11,
    const float* __restrict__ d_e12, float* __restrict__ g_img1) {
  __shared__ float t0[kE][kE + 1], t1[kE][kE + 1], t2[kE][kE + 1];
  __shared__ float hs[3][kE][kT + 1];
  const size_t plane = (size_t)blockIdx.z * H * W;
  const int x0 = blockIdx.x * kT - kR, y0 = blockIdx.y * kT - kR;
  for (int i = threadIdx.y * kT + threadIdx.x; i < kE * kE; i += kT * kT) {
    const int ty = i / kE, tx = i - ty * kE;
    const int yy = y0 + ty, xx = x0 + tx;
    float g = 0.f, a = 0.f, b = 0.f, c = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const size_t o = plane + (size_t)yy * W + xx;
      g = g_map[o];
      a = d_mu1[o];
      b = d_e11[o];
      c = d_e12[o];
    }
    t0[ty][tx] = g * a;
    t1[ty][tx] = g * b;
    t2[ty][tx] = g * c;
  }
  __syncthreads();
  for (int i = threadIdx.y * kT + threadIdx.x; i < kE * kT; i += kT * kT) {
    const int ty = i / kT, tx = i - ty * kT;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
    for (int dx = 0; dx < 11; ++dx) {
      const float w = win.g[dx];
      r0 = fmaf(w, t0[ty][tx + dx], r0);
      r1 = fmaf(w, t1[ty][tx + dx], r1);
      r2 = fmaf(w, t2[ty][tx + dx], r2);
    }
    hs[0][ty][tx] = r0;
    hs[1][ty][tx] = r1;
    hs[2][ty][tx] = r2;
  }
  __syncthreads();
  const int x = blockIdx.x * kT + threadIdx.x, y = blockIdx.y * kT + threadIdx.y;
  if (x >= W || y >= H) return;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
  for (int dy = 0; dy < 11; ++dy) {
    const float wy = win.g[dy];
    b0 = fmaf(wy, hs[0][threadIdx.y + dy][threadIdx.x], b0);
    b1 = fmaf(wy, hs[1][threadIdx.y + dy][threadIdx.x], b1);
    b2 = fmaf(wy, hs[2][threadIdx.y + dy][threadIdx.x], b2);
  }
  const size_t o = plane + (size_t)y * W + x;
  g_img1[o] = b0 + 2.f * img1[o] * b1 + img2[o] * b2;
}

Win make_window() {
  // gaussian(11, 1.5) of slam_external_splatam.py:41-46, normalised in float
  Win w;
  double g[11], s = 0.0;
  for (int i = 0; i < 11; ++i) {
    g[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
    s += g[i];
  }
  for (int i = 0; i < 11; ++i) w.g[i] = (float)(g[i] / s);
  return w;
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int xrd_ssim_fwd(int channels, int height, int width, const float* img1,
                 const float* img2, float* ssim_map, float* d_mu1, float* d_e11,
                 float* d_e12, xrd_stream_t stream) {
  if (channels < 1 || height < 1 || width < 1 || !img1 || !img2 || !ssim_map)
    return XRD_ERR_ARG;
  const bool save = d_mu1 || d_e11 || d_e12;
  if (save && (!d_mu1 || !d_e11 || !d_e12)) return XRD_ERR_ARG;
  const dim3 grid((width + kT - 1) / kT, (height + kT - 1) / kT, channels);
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(kT, kT), 0, (hipStream_t)stream,
                     make_window(), height, width, img1, img2, ssim_map, d_mu1,
                     d_e11, d_e12);
  return check_launch("xrd_ssim_fwd");
}

int xrd_ssim_bwd(int channels, int height, int width, const float* img1,
                 const float* img2, const float* g_map, const float* d_mu1,
                 const float* d_e11, const float* d_e12, float* g_img1,
                 xrd_stream_t stream) {
  if (channels < 1 || height < 1 || width < 1 || !img1 || !img2 || !g_map ||
      !d_mu1 || !d_e11 || !d_e12 || !g_img1)
    return XRD_ERR_ARG;
  const dim3 grid((width + kT - 1) / kT, (height + kT - 1) / kT, channels);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(kT, kT), 0, (hipStream_t)stream,
                     make_window(), height, width, img1, img2, g_map, d_mu1,
                     d_e11, d_e12, g_img1);
  return check_launch("xrd_ssim_bwd");
}

}  // extern "C"
