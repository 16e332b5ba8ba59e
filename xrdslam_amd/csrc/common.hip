// Error plumbing shared by all entry points.
#include "common.h"

namespace xrd {
thread_local const char* g_last_error = "";

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_error = hipGetErrorString(e);
    (void)what;
    return XRD_ERR_LAUNCH;
  }
  return XRD_OK;
}

namespace {
__global__ __launch_bounds__(256) void zero_kernel(float4* p4, size_t n4,
                                                   float* tail, int ntail) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t k = i; k < n4; k += stride) p4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < (size_t)ntail) tail[i] = 0.f;
}
}  // namespace

int zero_floats(float* p, size_t n, void* stream) {
  if (n == 0) return XRD_OK;
  if (p == nullptr) return XRD_ERR_ARG;
  // align the vector part to 16 bytes
  size_t head = ((16 - ((uintptr_t)p & 15)) & 15) / 4;
  if (head > n) head = n;
  const size_t n4 = (n - head) / 4;
  const size_t tail0 = head + 4 * n4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  hipStream_t st = (hipStream_t)stream;
  if (head)
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, st, nullptr, 0, p,
                       (int)head);
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                     reinterpret_cast<float4*>(p + head), n4, p + tail0,
                     (int)(n - tail0));
  return check_launch("zero_floats");
}
}  // namespace xrd

extern "C" {
int xrd_abi_version(void) { return 1; }
const char* xrd_last_error(void) { return xrd::g_last_error; }
}
