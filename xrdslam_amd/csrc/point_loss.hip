// Point-SLAM mapping: compositing + loss + their backward as one launch
// (raw2outputs_nerf_color2, slam/model_components/utils.py:247-294, and the
// mapping branch of get_loss_dict, slam/models/conv_onet_pointslam.py:
// 190-204).  One thread per ray, S <= 16 samples in registers:
//   alpha_s = sigmoid(coef occ_s)   (occ_s = -100 where the sample has no
//                                    neighbours, conv_onet_pointslam.py:441)
//   w_s = alpha_s prod_{j<s} (1 - alpha_j + 1e-10),  W = sum w + 1e-10
//   depth = sum w z / W,  colour = sum w rgb / W
//   a ray counts when target_d > 0, >= min_valid of its samples have
//   neighbours, its depth is not NaN (and the batch mask, if given, keeps it)
//   loss = sum |target_d - depth| + w_color sum |target_rgb - colour|
// The loss is a plain sum with unit weight, so the gradient w.r.t. raw
// ([rgb, occ] per sample) is final when the kernel returns; the autograd
// wrapper scales it by the incoming gradient.  Restated from the reference's
// formulas, not copied; parity: tests/test_pointslam_hip.py.
#include "common.h"

namespace xrd {
namespace {

constexpr int kMaxS = 16;

__global__ __launch_bounds__(256) void point_map_loss_kernel(
    int n, int S, const float* __restrict__ raw,
    const uint8_t* __restrict__ point_mask, const float* __restrict__ z_vals,
    const float* __restrict__ target_d, const float* __restrict__ target_rgb,
    const uint8_t* __restrict__ ray_valid, float coef, float w_color,
    int min_valid, float* __restrict__ loss, float* __restrict__ g_raw) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  float l_geo = 0.f, l_rgb = 0.f;
  if (ray < n) {
    float alpha[kMaxS], T[kMaxS], w[kMaxS], z[kMaxS];
    int cnt = 0;
    float run = 1.f, wsum = 0.f;
    for (int s = 0; s < S; ++s) {
      const bool has = point_mask[ray * S + s] != 0;
      cnt += has;
      const float occ = has ? raw[(ray * S + s) * 4 + 3] : -100.f;
      alpha[s] = 1.f / (1.f + expf(-(coef * occ)));
      T[s] = run;
      w[s] = alpha[s] * run;
      run = run * (1.f - alpha[s] + 1e-10f);
      wsum += w[s];
      z[s] = z_vals[ray * S + s];
    }
    const float W = wsum + 1e-10f;
    float A = 0.f;
    for (int s = 0; s < S; ++s) A += w[s] * z[s];
    const float depth = A / W;
    const float td = target_d[ray];
    bool m = td > 0.f && cnt >= min_valid && !(depth != depth);
    if (ray_valid != nullptr) m = m && ray_valid[ray] != 0;
    float col[3] = {0.f, 0.f, 0.f}, g_col[3] = {0.f, 0.f, 0.f};
    const bool color = target_rgb != nullptr;
    if (color) {
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c) col[c] += w[s] * raw[(ray * S + s) * 4 + c];
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] /= W;
    }
    float g_depth = 0.f;
    if (m) {
      const float e = td - depth;
      l_geo = fabsf(e);
      g_depth = e > 0.f ? -1.f : (e < 0.f ? 1.f : 0.f);
      if (color) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float ec = target_rgb[ray * 3 + c] - col[c];
          l_rgb += fabsf(ec);
          g_col[c] = w_color * (ec > 0.f ? -1.f : (ec < 0.f ? 1.f : 0.f));
        }
        l_rgb *= w_color;
      }
    }
    // backward: g_w_s, then the transmittance chain from the last sample
    float tail = 0.f;   // sum_{k > s} g_w_k w_k
    for (int s = S - 1; s >= 0; --s) {
      float g_w = g_depth * (z[s] - depth) / W;
      f32x4 out = {0.f, 0.f, 0.f, 0.f};
      if (color) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          g_w += g_col[c] * (raw[(ray * S + s) * 4 + c] - col[c]) / W;
          out[c] = g_col[c] * w[s] / W;
        }
      }
      const float g_alpha =
          g_w * T[s] - tail / (1.f - alpha[s] + 1e-10f);
      tail += g_w * w[s];
      out[3] = g_alpha * coef * alpha[s] * (1.f - alpha[s]);
      *reinterpret_cast<f32x4*>(g_raw + (ray * S + s) * 4) = out;
    }
  }
  // block sums -> two atomics per block
  __shared__ float red[2][4];
  const float sg = wave_sum(l_geo), sr = wave_sum(l_rgb);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    red[0][wave] = sg;
    red[1][wave] = sr;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(loss, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    atomicAdd(loss + 1, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" int xrd_point_map_loss(int n_rays, int n_samples, const float* raw,
                                  const uint8_t* point_mask,
                                  const float* z_vals, const float* target_d,
                                  const float* target_rgb,
                                  const uint8_t* ray_valid, float sigmoid_coef,
                                  float w_color, int min_valid_points,
                                  float* loss, float* g_raw,
                                  xrd_stream_t stream) {
  if (n_rays < 0 || n_samples < 1 || n_samples > kMaxS) return XRD_ERR_ARG;
  if (!loss) return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int rc = zero_floats(loss, 2, st);
  if (rc != XRD_OK) return rc;
  if (n_rays == 0) return XRD_OK;
  if (!raw || !point_mask || !z_vals || !target_d || !g_raw)
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(point_map_loss_kernel, dim3((n_rays + 255) / 256),
                     dim3(256), 0, st, n_rays, n_samples, raw, point_mask,
                     z_vals, target_d, target_rgb, ray_valid, sigmoid_coef,
                     w_color, min_valid_points, loss, g_raw);
  return check_launch("xrd_point_map_loss");
}
