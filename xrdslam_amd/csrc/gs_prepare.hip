// SplaTAM per-iteration glue as kernels (gfx950): what the reference builds
// with ~40 small torch ops before and after every raster pass.
//
//   xrd_gs_prepare_fwd / _bwd — transform_to_frame + the two render-variable
//     dictionaries (slam/model_components/slam_helpers_splatam.py:263-292,
//     205-260; call site gaussian_cloud_splatam.py:47-62): camera-frame
//     centres p' = R p + t of w2c = inverse(c2w) (rigid inverse evaluated in
//     the kernel), normalised rotations, sigmoid opacities, exp scales tiled
//     to 3, and the depth / silhouette "colours" (z, 1, z^2) in the first
//     frame's camera — one launch each way for all N Gaussians instead of two
//     GEMMs with K = 4, two normalisations, sigmoids, exps, tiles, stacks.
//     Backward: gradients to the Gaussians and / or to c2w (12 block-reduced
//     sums, chained through the inverse by one thread of the last block).
//   xrd_gs_loss_fwd / _bwd — GaussianSplatting.get_loss_dict
//     (slam/models/gaussian_splatting.py:102-160) without boolean-mask
//     indexing (a host sync per iteration in torch): masked L1 depth + L1
//     colour, tracking sums / mapping means, as a statistics launch and a
//     gradient launch.
#include <hip/hip_runtime.h>

#include "common.h"

namespace xrd {
namespace {

struct Rigid {
  float R[3][3], t[3];   // w2c = [R | t]
  float Rf[3], tf;       // third row of the first frame's w2c: z = Rf . p' + tf
};

// w2c of a rigid c2w: [Rc^T | -Rc^T tc]
__device__ __forceinline__ void rigid_inverse(const float* __restrict__ c2w,
                                              float (&R)[3][3],
                                              float (&t)[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i][j] = c2w[j * 4 + i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
    t[i] = -(R[i][0] * c2w[3] + R[i][1] * c2w[7] + R[i][2] * c2w[11]);
}

__device__ __forceinline__ void load_rigid(const float* __restrict__ pose,
                                           int pose_is_c2w,
                                           const float* __restrict__ first_w2c,
                                           Rigid& g) {
  if (pose_is_c2w) {
    rigid_inverse(pose, g.R, g.t);
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) g.R[i][j] = pose[i * 4 + j];
      g.t[i] = pose[i * 4 + 3];
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) g.Rf[j] = first_w2c[8 + j];
  g.tf = first_w2c[11];
}

__global__ __launch_bounds__(256) void gs_prepare_fwd_kernel(
    int n, const float* __restrict__ means, const float* __restrict__ urot,
    const float* __restrict__ logit, const float* __restrict__ lscale,
    const float* __restrict__ pose, int pose_is_c2w,
    const float* __restrict__ first_w2c, float* __restrict__ pts,
    float* __restrict__ rot, float* __restrict__ opac,
    float* __restrict__ scales, float* __restrict__ dscol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Rigid g;
  load_rigid(pose, pose_is_c2w, first_w2c, g);
  const float p[3] = {means[i * 3], means[i * 3 + 1], means[i * 3 + 2]};
  float q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // the association of (w2c @ [p,1]) summed k = 0..3
    q[a] = ((g.R[a][0] * p[0] + g.R[a][1] * p[1]) + g.R[a][2] * p[2]) + g.t[a];
    pts[i * 3 + a] = q[a];
  }
  const f32x4 u = *reinterpret_cast<const f32x4*>(urot + i * 4);
  const float nrm = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
  const float inv = 1.f / fmaxf(nrm, 1e-12f);   // F.normalize
  *reinterpret_cast<f32x4*>(rot + i * 4) = u * inv;
  const float l = logit[i];
  opac[i] = 1.f / (1.f + expf(-l));
  const float s = expf(lscale[i]);
  scales[i * 3] = s;
  scales[i * 3 + 1] = s;
  scales[i * 3 + 2] = s;
  const float z =
      ((g.Rf[0] * q[0] + g.Rf[1] * q[1]) + g.Rf[2] * q[2]) + g.tf;
  dscol[i * 3] = z;
  dscol[i * 3 + 1] = 1.f;
  dscol[i * 3 + 2] = z * z;
}

// acc [16] floats (zero on entry): 12 sums of the pose gradient + ticket;
// g_pose [16] written by the last block
__global__ __launch_bounds__(256) void gs_prepare_bwd_kernel(
    int n, const float* __restrict__ means, const float* __restrict__ urot,
    const float* __restrict__ logit, const float* __restrict__ lscale,
    const float* __restrict__ pose, int pose_is_c2w,
    const float* __restrict__ first_w2c, const float* __restrict__ g_pts,
    const float* __restrict__ g_rot, const float* __restrict__ g_opac,
    const float* __restrict__ g_scales, const float* __restrict__ g_dscol,
    float* __restrict__ g_means, float* __restrict__ g_urot,
    float* __restrict__ g_logit, float* __restrict__ g_lscale,
    float* __restrict__ acc, float* __restrict__ g_pose) {
  __shared__ float red[4][12];
  Rigid g;
  load_rigid(pose, pose_is_c2w, first_w2c, g);
  // d loss / d w2c, rows a: gq[a] * p (R part), gq[a] (t part), summed over
  // the thread's Gaussians (grid-stride: the pose sums cost 12 same-address
  // atomics and a ticket per BLOCK, ~25 ns each — one block a CU)
  float v[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) v[k] = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    float gq[3], p[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) p[a] = means[i * 3 + a];
    float q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
      q[a] = ((g.R[a][0] * p[0] + g.R[a][1] * p[1]) + g.R[a][2] * p[2]) +
             g.t[a];
    const float z = ((g.Rf[0] * q[0] + g.Rf[1] * q[1]) + g.Rf[2] * q[2]) + g.tf;
    float gz = 0.f;
    if (g_dscol != nullptr)
      gz = g_dscol[i * 3] + 2.f * z * g_dscol[i * 3 + 2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
      gq[a] = (g_pts ? g_pts[i * 3 + a] : 0.f) + g.Rf[a] * gz;
    if (g_means != nullptr) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
        g_means[i * 3 + a] =
            g.R[0][a] * gq[0] + g.R[1][a] * gq[1] + g.R[2][a] * gq[2];
    }
    if (g_urot != nullptr) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(urot + i * 4);
      const float nrm =
          sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
      const float inv = 1.f / fmaxf(nrm, 1e-12f);
      f32x4 gr = {0.f, 0.f, 0.f, 0.f};
      if (g_rot) gr = *reinterpret_cast<const f32x4*>(g_rot + i * 4);
      const f32x4 r = u * inv;
      const float d = r[0] * gr[0] + r[1] * gr[1] + r[2] * gr[2] + r[3] * gr[3];
      // below the eps the normalisation is a constant scale
      *reinterpret_cast<f32x4*>(g_urot + i * 4) =
          nrm > 1e-12f ? (gr - r * d) * inv : gr * inv;
    }
    if (g_logit != nullptr) {
      const float o = 1.f / (1.f + expf(-logit[i]));
      g_logit[i] = (g_opac ? g_opac[i] : 0.f) * o * (1.f - o);
    }
    if (g_lscale != nullptr) {
      const float s = expf(lscale[i]);
      g_lscale[i] = g_scales ? (g_scales[i * 3] + g_scales[i * 3 + 1] +
                                g_scales[i * 3 + 2]) * s
                             : 0.f;
    }
    if (g_pose != nullptr) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) v[a * 4 + b] += gq[a] * p[b];
        v[a * 4 + 3] += gq[a];
      }
    }
  }
  if (g_pose == nullptr) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const float s = wave_sum(v[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 12)
    atomicAdd(acc + threadIdx.x, (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                     (red[2][threadIdx.x] + red[3][threadIdx.x]));
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* ticket = reinterpret_cast<unsigned*>(acc + 12);
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      __threadfence();
      float gW[3][4];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 4; ++b)
          gW[a][b] = __builtin_nontemporal_load(acc + a * 4 + b);
      for (int k = 0; k < 16; ++k) g_pose[k] = 0.f;
      if (!pose_is_c2w) {
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 4; ++b) g_pose[a * 4 + b] = gW[a][b];
      } else {
        // W = Rc^T, s = -Rc^T tc:  gRc = gW^T - tc (x) gs,  gtc = -Rc gs
        const float tc[3] = {pose[3], pose[7], pose[11]};
        for (int a = 0; a < 3; ++a) {
          for (int b = 0; b < 3; ++b)
            g_pose[a * 4 + b] = gW[b][a] - tc[a] * gW[b][3];
          g_pose[a * 4 + 3] = -(pose[a * 4 + 0] * gW[0][3] +
                                pose[a * 4 + 1] * gW[1][3] +
                                pose[a * 4 + 2] * gW[2][3]);
        }
      }
      for (int k = 0; k < 13; ++k) acc[k] = 0.f;   // reusable without a fill
    }
  }
}

// ---- losses -------------------------------------------------------------
// stats [8] doubles: 0 depth error sum, 1 depth count, 2 colour error sum,
// 3 colour count (elements).  Renders are [3,H,W]; the targets are the
// frame's device images as the reference hands them over: depth [H,W] and
// colour [H,W,3] (the reference permutes it per iteration).
struct LossArgs {
  int H, W, is_mapping, use_sil;
  float sil_thres, w_depth, w_rgb, rgb_l1_scale;
};

__device__ __forceinline__ bool loss_mask(const LossArgs& a, float td, float d,
                                          float sil, float d2) {
  const float unc = d2 - d * d;
  bool m = td > 0.f && !(d != d) && !(unc != unc);
  if (!a.is_mapping && a.use_sil) m = m && sil > a.sil_thres;
  return m;
}

__global__ __launch_bounds__(256) void gs_loss_stats_kernel(
    LossArgs a, const float* __restrict__ rgb, const float* __restrict__ dsil,
    const float* __restrict__ tgt_d, const float* __restrict__ tgt_rgb,
    double* __restrict__ stats) {
  __shared__ double red[4][4];
  const int HW = a.H * a.W;
  double sd = 0.0, nd = 0.0, sc = 0.0, nc = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW;
       i += gridDim.x * blockDim.x) {
    const float d = dsil[i], sil = dsil[HW + i], d2 = dsil[2 * HW + i];
    const bool m = loss_mask(a, tgt_d[i], d, sil, d2);
    if (m) {
      sd += (double)fabsf(tgt_d[i] - d);
      nd += 1.0;
    }
    // colour: tracking with the silhouette mask sums the masked pixels,
    // tracking without it and mapping use every pixel
    const bool mc = (!a.is_mapping && a.use_sil) ? m : true;
    if (mc) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        sc += (double)fabsf(tgt_rgb[i * 3 + ch] - rgb[ch * HW + i]);
      nc += 3.0;
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double v[4] = {sd, nd, sc, nc};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double s = wave_sum(v[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 4)
    atomicAdd(stats + threadIdx.x,
              (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                  (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

__global__ void gs_loss_finalize_kernel(LossArgs a,
                                        const double* __restrict__ stats,
                                        float* __restrict__ loss_depth,
                                        float* __restrict__ loss_rgb) {
  if (threadIdx.x != 0) return;
  // tracking: sums; mapping: means (the colour mean is scaled by
  // rgb_l1_scale = 0.8: the 0.2 (1 - SSIM) part is added by the caller)
  const double dep = a.is_mapping ? stats[0] / stats[1] : stats[0];
  const double col = a.is_mapping ? stats[2] / stats[3] : stats[2];
  loss_depth[0] = (float)dep * a.w_depth;
  loss_rgb[0] = (float)col * a.rgb_l1_scale * a.w_rgb;
}

__global__ __launch_bounds__(256) void gs_loss_grad_kernel(
    LossArgs a, const float* __restrict__ rgb, const float* __restrict__ dsil,
    const float* __restrict__ tgt_d, const float* __restrict__ tgt_rgb,
    const double* __restrict__ stats, const float* __restrict__ g_depth_up,
    const float* __restrict__ g_rgb_up, float* __restrict__ g_rgb,
    float* __restrict__ g_dsil) {
  const int HW = a.H * a.W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW) return;
  const float gd_up = g_depth_up ? g_depth_up[0] : 0.f;
  const float gc_up = g_rgb_up ? g_rgb_up[0] : 0.f;
  const float kd = a.w_depth * gd_up /
                   (a.is_mapping ? (float)stats[1] : 1.f);
  const float kc = a.w_rgb * a.rgb_l1_scale * gc_up /
                   (a.is_mapping ? (float)stats[3] : 1.f);
  const float d = dsil[i], sil = dsil[HW + i], d2 = dsil[2 * HW + i];
  const bool m = loss_mask(a, tgt_d[i], d, sil, d2);
  const float e = tgt_d[i] - d;
  g_dsil[i] = m ? (e > 0.f ? -kd : (e < 0.f ? kd : 0.f)) : 0.f;
  g_dsil[HW + i] = 0.f;
  g_dsil[2 * HW + i] = 0.f;
  const bool mc = (!a.is_mapping && a.use_sil) ? m : true;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float ec = tgt_rgb[i * 3 + ch] - rgb[ch * HW + i];
    g_rgb[ch * HW + i] = mc ? (ec > 0.f ? -kc : (ec < 0.f ? kc : 0.f)) : 0.f;
  }
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int xrd_gs_prepare_fwd(int n, const float* means3D, const float* unnorm_rot,
                       const float* logit_opacities, const float* log_scales,
                       const float* pose, int pose_is_c2w,
                       const float* first_w2c, float* pts, float* rotations,
                       float* opacities, float* scales, float* ds_colors,
                       xrd_stream_t stream) {
  if (n < 0) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!means3D || !unnorm_rot || !logit_opacities || !log_scales || !pose ||
      !first_w2c || !pts || !rotations || !opacities || !scales || !ds_colors)
    return XRD_ERR_ARG;
  hipLaunchKernelGGL(gs_prepare_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, n, means3D, unnorm_rot,
                     logit_opacities, log_scales, pose, pose_is_c2w, first_w2c,
                     pts, rotations, opacities, scales, ds_colors);
  return check_launch("xrd_gs_prepare_fwd");
}

int xrd_gs_prepare_bwd(int n, const float* means3D, const float* unnorm_rot,
                       const float* logit_opacities, const float* log_scales,
                       const float* pose, int pose_is_c2w,
                       const float* first_w2c, const float* g_pts,
                       const float* g_rotations, const float* g_opacities,
                       const float* g_scales, const float* g_ds_colors,
                       float* g_means3D, float* g_unnorm_rot,
                       float* g_logit_opacities, float* g_log_scales,
                       float* acc, float* g_pose, xrd_stream_t stream) {
  if (n < 0) return XRD_ERR_ARG;
  if (n == 0) return XRD_OK;
  if (!means3D || !unnorm_rot || !logit_opacities || !log_scales || !pose ||
      !first_w2c)
    return XRD_ERR_ARG;
  if (g_pose != nullptr && acc == nullptr) return XRD_ERR_ARG;
  int blocks = (n + 255) / 256;
  if (g_pose != nullptr && blocks > 256) blocks = 256;   // see the kernel
  hipLaunchKernelGGL(gs_prepare_bwd_kernel, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, n, means3D, unnorm_rot,
                     logit_opacities, log_scales, pose, pose_is_c2w, first_w2c,
                     g_pts, g_rotations, g_opacities, g_scales, g_ds_colors,
                     g_means3D, g_unnorm_rot, g_logit_opacities, g_log_scales,
                     acc, g_pose);
  return check_launch("xrd_gs_prepare_bwd");
}

int xrd_gs_loss_fwd(int H, int W, int is_mapping, int use_sil, float sil_thres,
                    float w_depth, float w_rgb, float rgb_l1_scale,
                    const float* rgb, const float* depth_sil,
                    const float* target_d, const float* target_rgb,
                    double* stats, float* loss_depth, float* loss_rgb,
                    xrd_stream_t stream) {
  if (H < 1 || W < 1) return XRD_ERR_ARG;
  if (!rgb || !depth_sil || !target_d || !target_rgb || !stats ||
      !loss_depth || !loss_rgb)
    return XRD_ERR_ARG;
  const LossArgs a = {H, W, is_mapping, use_sil, sil_thres, w_depth, w_rgb,
                      rgb_l1_scale};
  hipStream_t st = (hipStream_t)stream;
  int rc = zero_floats(reinterpret_cast<float*>(stats), 8, stream);
  if (rc != XRD_OK) return rc;
  const int HW = H * W;
  int blocks = (HW + 255) / 256;
  if (blocks > 128) blocks = 128;   // 4 same-address atomics a block
  hipLaunchKernelGGL(gs_loss_stats_kernel, dim3(blocks), dim3(256), 0, st, a,
                     rgb, depth_sil, target_d, target_rgb, stats);
  hipLaunchKernelGGL(gs_loss_finalize_kernel, dim3(1), dim3(64), 0, st, a,
                     stats, loss_depth, loss_rgb);
  return check_launch("xrd_gs_loss_fwd");
}

int xrd_gs_loss_bwd(int H, int W, int is_mapping, int use_sil, float sil_thres,
                    float w_depth, float w_rgb, float rgb_l1_scale,
                    const float* rgb, const float* depth_sil,
                    const float* target_d, const float* target_rgb,
                    const double* stats, const float* g_loss_depth,
                    const float* g_loss_rgb, float* g_rgb, float* g_depth_sil,
                    xrd_stream_t stream) {
  if (H < 1 || W < 1) return XRD_ERR_ARG;
  if (!rgb || !depth_sil || !target_d || !target_rgb || !stats || !g_rgb ||
      !g_depth_sil)
    return XRD_ERR_ARG;
  const LossArgs a = {H, W, is_mapping, use_sil, sil_thres, w_depth, w_rgb,
                      rgb_l1_scale};
  const int HW = H * W;
  hipLaunchKernelGGL(gs_loss_grad_kernel, dim3((HW + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, a, rgb, depth_sil, target_d,
                     target_rgb, stats, g_loss_depth, g_loss_rgb, g_rgb,
                     g_depth_sil);
  return check_launch("xrd_gs_loss_bwd");
}

}  // extern "C"
