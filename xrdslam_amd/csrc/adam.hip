// Fused Adam over selected cells of a feature grid (HBM-bound, 16 B/lane).
//
// Replaces, for NICE-SLAM frustum feature selection, the per-iteration
//   val[mask] = val_grad  (conv_onet.py:105-114)   +  torch.optim.Adam.step()
//   + val[mask] = val_grad.clone()  (conv_onet.py:94-103)
// by an in-place update of exactly the selected cells.  Arithmetic follows
// torch.optim.Adam (single-tensor path, no amsgrad / weight decay):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// The step count t comes either from the host (step >= 1) or from a device
// counter (step_dev, for hipGraph replay: the same launch is replayed with a
// counter that a previous node increments); bias corrections are evaluated in
// double by one lane per block.
#include "common.h"

namespace xrd {
namespace {

__device__ __forceinline__ void adam_cells_body(
    float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, const int32_t* __restrict__ cell_idx,
    int64_t n_cells, int vec_per_cell, float lr, float beta1, float beta2,
    float eps, int step_host, int32_t* __restrict__ step_dev, int tick,
    const int32_t* __restrict__ n_cells_dev, int zero_grad) {
  __shared__ float s_coef[2];
  // tick != 0: step_dev = {steps taken so far, ticket}; this launch is step
  // t = step_dev[0] + 1 and the LAST block to finish stores t (every block
  // has read the counter by then) — no separate increment launch
  const int t_now = step_dev ? step_dev[0] + (tick ? 1 : 0) : step_host;
  // (the two f64 pow() of the bias corrections on two lanes side by side: they
  // are a serial ~2 us each in front of a launch that updates 170 KB)
  if (threadIdx.x < 2) {
    const bool first = threadIdx.x == 0;
    const double bc =
        1.0 - pow(first ? (double)beta1 : (double)beta2, (double)t_now);
    s_coef[threadIdx.x] =
        first ? (float)((double)lr / bc) : (float)(1.0 / sqrt(bc));
  }
  __syncthreads();
  const float step_size = s_coef[0], inv_bc2_sqrt = s_coef[1];
  // persistent hipGraphs: the launch covers the list's capacity, the number of
  // valid entries is read at execution time
  if (n_cells_dev) n_cells = min(n_cells, (int64_t)n_cells_dev[0]);
  const int64_t total = n_cells * vec_per_cell;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = i / vec_per_cell;
    const int sub = (int)(i - cell * vec_per_cell);
    const int64_t c = cell_idx ? (int64_t)cell_idx[cell] : cell;
    const int64_t off = (c * vec_per_cell + sub) * 4;
    const int64_t soff = i * 4;  // moments are compact: [n_cells][cell_floats]
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + off);
    f32x4 mv = *reinterpret_cast<const f32x4*>(m + soff);
    f32x4 vv = *reinterpret_cast<const f32x4*>(v + soff);
    f32x4 pv = *reinterpret_cast<const f32x4*>(p + off);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // torch: exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
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
  if (tick) {
    __syncthreads();
    if (threadIdx.x == 0 &&
        atomicAdd(step_dev + 1, 1) == (int)gridDim.x - 1) {
      step_dev[1] = 0;
      step_dev[0] = t_now;
    }
  }
}

__global__ __launch_bounds__(512) void adam_cells_kernel(
    float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, const int32_t* __restrict__ cell_idx,
    int64_t n_cells, int vec_per_cell, float lr, float beta1, float beta2,
    float eps, int step_host, int32_t* __restrict__ step_dev, int tick,
    const int32_t* __restrict__ n_cells_dev, int zero_grad) {
  adam_cells_body(p, g, m, v, cell_idx, n_cells, vec_per_cell, lr, beta1,
                  beta2, eps, step_host, step_dev, tick, n_cells_dev,
                  zero_grad);
}

// several grids in one launch (blockIdx.y = the grid): a mapping iteration of
// the colour stage steps three feature grids — three launches of ~9 us each,
// every one a short dependent chain (count -> cell list -> moments), were 10 %
// of the iteration
struct AdamSets {
  xrd_adam_cells_set s[XRD_ADAM_MAX_SETS];
};
__global__ __launch_bounds__(512) void adam_cells_multi_kernel(
    AdamSets a, int vec_per_cell, float beta1, float beta2, float eps,
    int zero_grad) {
  const xrd_adam_cells_set& s = a.s[blockIdx.y];
  adam_cells_body(s.param, s.grad, s.m, s.v, s.cell_idx, s.n_cells,
                  vec_per_cell, s.lr, beta1, beta2, eps, 0, s.step_ticket, 2,
                  s.n_cells_dev, zero_grad);
}

}  // namespace
}  // namespace xrd

static int adam_launch(float* param, float* g, float* m, float* v,
                       const int32_t* cell_idx, int64_t n_cells,
                       int cell_floats, float lr, float beta1, float beta2,
                       float eps, int step, int32_t* step_dev, int tick,
                       const int32_t* n_cells_dev, int zero_grad,
                       xrd_stream_t stream) {
  if (n_cells < 0 || cell_floats <= 0 || (cell_floats & 3) ||
      (step < 1 && !step_dev))
    return XRD_ERR_ARG;
  // an empty selection is a no-op like torch's Adam over an empty val[mask]
  // (the compact moments are then zero-sized: NULL data pointers are fine)
  if (n_cells == 0) return XRD_OK;
  if (!param || !g || !m || !v) return XRD_ERR_ARG;
  const int vec = cell_floats / 4;
  const int64_t total = n_cells * vec;
  // one block a CU: the step-counter ticket is one same-address atomic a
  // block, ~25 ns each (4096 blocks measured 2.5x the time of the update)
  int64_t blocks = (total + 511) / 512;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(xrd::adam_cells_kernel, dim3((unsigned)blocks), dim3(512),
                     0, (hipStream_t)stream, param, g, m, v, cell_idx, n_cells,
                     vec, lr, beta1, beta2, eps, step, step_dev, tick,
                     n_cells_dev, zero_grad);
  return xrd::check_launch("xrd_adam_cells");
}

extern "C" int xrd_adam_cells(float* param, float* g, float* m, float* v,
                              const int32_t* cell_idx, int64_t n_cells,
                              int cell_floats, float lr, float beta1,
                              float beta2, float eps, int step, int zero_grad,
                              xrd_stream_t stream) {
  return adam_launch(param, g, m, v, cell_idx, n_cells, cell_floats, lr, beta1,
                     beta2, eps, step, nullptr, 0, nullptr, zero_grad, stream);
}

extern "C" int xrd_adam_cells_devstep(float* param, float* g, float* m,
                                      float* v, const int32_t* cell_idx,
                                      int64_t n_cells, int cell_floats,
                                      float lr, float beta1, float beta2,
                                      float eps, const int32_t* step_dev,
                                      int zero_grad, xrd_stream_t stream) {
  if (!step_dev) return XRD_ERR_ARG;
  return adam_launch(param, g, m, v, cell_idx, n_cells, cell_floats, lr, beta1,
                     beta2, eps, 0, const_cast<int32_t*>(step_dev), 0, nullptr,
                     zero_grad, stream);
}

extern "C" int xrd_adam_cells_devcount(float* param, float* g, float* m,
                                       float* v, const int32_t* cell_idx,
                                       int64_t capacity, int cell_floats,
                                       float lr, float beta1, float beta2,
                                       float eps, const int32_t* step_dev,
                                       const int32_t* n_cells_dev,
                                       int zero_grad, xrd_stream_t stream) {
  if (!step_dev || !n_cells_dev || !cell_idx) return XRD_ERR_ARG;
  return adam_launch(param, g, m, v, cell_idx, capacity, cell_floats, lr,
                     beta1, beta2, eps, 0, const_cast<int32_t*>(step_dev), 0,
                     n_cells_dev, zero_grad, stream);
}

extern "C" int xrd_adam_cells_tick(float* param, float* g, float* m, float* v,
                                   const int32_t* cell_idx, int64_t n_cells,
                                   int cell_floats, float lr, float beta1,
                                   float beta2, float eps,
                                   int32_t* step_ticket,
                                   const int32_t* n_cells_dev, int zero_grad,
                                   xrd_stream_t stream) {
  if (!step_ticket || (n_cells_dev && !cell_idx)) return XRD_ERR_ARG;
  return adam_launch(param, g, m, v, cell_idx, n_cells, cell_floats, lr, beta1,
                     beta2, eps, 0, step_ticket, 2, n_cells_dev, zero_grad,
                     stream);
}

extern "C" int xrd_adam_cells_multi(int n_sets, const xrd_adam_cells_set* sets,
                                    int cell_floats, float beta1, float beta2,
                                    float eps, int zero_grad,
                                    xrd_stream_t stream) {
  if (n_sets < 0 || n_sets > XRD_ADAM_MAX_SETS || (n_sets && !sets) ||
      cell_floats <= 0 || (cell_floats & 3))
    return XRD_ERR_ARG;
  xrd::AdamSets a = {};
  int n = 0;
  int64_t most = 0;
  for (int i = 0; i < n_sets; ++i) {
    const xrd_adam_cells_set& s = sets[i];
    if (s.n_cells < 0 || !s.step_ticket || (s.n_cells_dev && !s.cell_idx))
      return XRD_ERR_ARG;
    if (s.n_cells == 0) continue;   // empty selection: a no-op like torch's
    if (!s.param || !s.grad || !s.m || !s.v) return XRD_ERR_ARG;
    a.s[n++] = s;
    most = s.n_cells > most ? s.n_cells : most;
  }
  if (n == 0) return XRD_OK;
  const int vec = cell_floats / 4;
  int64_t blocks = (most * vec + 511) / 512;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(xrd::adam_cells_multi_kernel,
                     dim3((unsigned)blocks, (unsigned)n), dim3(512), 0,
                     (hipStream_t)stream, a, vec, beta1, beta2, eps,
                     zero_grad);
  return xrd::check_launch("xrd_adam_cells_multi");
}

