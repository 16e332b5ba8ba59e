// Fused Adam over selected cells of a feature grid (HBM-bound, 16 B/lane).
//
// Replaces, for NICE-SLAM frustum feature selection, the per-iteration
//   val[mask] = val_grad  (conv_onet.py:105-114)   +  torch.optim.Adam.step()
//   + val[mask] = val_grad.clone()  (conv_onet.py:94-103)
// by an in-place update of exactly the selected cells.  Arithmetic follows
// torch.optim.Adam (single-tensor path, no amsgrad / weight decay):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#include "common.h"

namespace xrd {
namespace {

__global__ __launch_bounds__(256) void adam_cells_kernel(
    float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, const int32_t* __restrict__ cell_idx,
    int64_t n_cells, int vec_per_cell, float beta1, float beta2, float eps,
    float step_size, float inv_bc2_sqrt, int zero_grad) {
  const int64_t total = n_cells * vec_per_cell;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = i / vec_per_cell;
    const int sub = (int)(i - cell * vec_per_cell);
    const int64_t c = cell_idx ? (int64_t)cell_idx[cell] : cell;
    const int64_t off = (c * vec_per_cell + sub) * 4;
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + off);
    const int64_t soff = i * 4;  // moments are compact: [n_cells][cell_floats]
    f32x4 mv = *reinterpret_cast<const f32x4*>(m + soff);
    f32x4 vv = *reinterpret_cast<const f32x4*>(v + soff);
    f32x4 pv = *reinterpret_cast<const f32x4*>(p + off);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // torch: exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
      mv[k] = mv[k] + (gv[k] - mv[k]) * (1.f - beta1);
      vv[k] = vv[k] * beta2 + (1.f - beta2) * gv[k] * gv[k];
      const float denom = sqrtf(vv[k]) * inv_bc2_sqrt + eps;
      pv[k] = pv[k] - step_size * (mv[k] / denom);
    }
    *reinterpret_cast<f32x4*>(m + soff) = mv;
    *reinterpret_cast<f32x4*>(v + soff) = vv;
    *reinterpret_cast<f32x4*>(p + off) = pv;
    if (zero_grad)
      *reinterpret_cast<f32x4*>(g + off) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

}  // namespace
}  // namespace xrd

extern "C" int xrd_adam_cells(float* param, float* g, float* m, float* v,
                              const int32_t* cell_idx, int64_t n_cells,
                              int cell_floats, float lr, float beta1,
                              float beta2, float eps, int step, int zero_grad,
                              xrd_stream_t stream) {
  if (!param || !g || !m || !v || n_cells < 0 || cell_floats <= 0 ||
      (cell_floats & 3) || step < 1)
    return XRD_ERR_ARG;
  if (n_cells == 0) return XRD_OK;
  // bias corrections in double like torch's python scalars, then to f32
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const int vec = cell_floats / 4;
  const int64_t total = n_cells * vec;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(xrd::adam_cells_kernel, dim3((unsigned)blocks), dim3(256),
                     0, (hipStream_t)stream, param, g, m, v, cell_idx, n_cells,
                     vec, beta1, beta2, eps, step_size, inv_bc2_sqrt,
                     zero_grad);
  return xrd::check_launch("xrd_adam_cells");
}
