// Do VALU instructions overlap with MFMAs of OTHER waves of the same SIMD on
// gfx950?  One block of 12 waves a CU (3 a SIMD, like nice_map_fused_kernel);
// every wave runs REPS x (8 MFMAs on 8 independent accumulators + K
// independent v_fma_f32).  If the f32 MFMA had a pipe of its own, time would
// be max(MFMA, VALU); measured (profiles/r06_mfma_valu_overlap.txt) it is
// their SUM for v_mfma_f32_16x16x4_f32 and close to the max for
// v_mfma_f32_16x16x32_bf16.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int MODE>   // MODE 0: f32 MFMA, 1: bf16 MFMA, 2: no MFMA
__global__ __launch_bounds__(768) void k(float* out, int reps, float seed) {
  f32x4 acc[8];
  float v[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  const float a = seed * 0.5f, b = seed * 0.25f;
  bf16x8 ab, bb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      if (MODE == 1)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < K / 8; ++j) {
        const int t = (i * (K / 8) + j) & 15;
        v[t] = __builtin_fmaf(v[t], 1.0001f, 0.5f);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int MODE>
float run(float* out, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<K, MODE>), dim3(256), dim3(768), 0, 0, out, reps, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<K, MODE>), dim3(256), dim3(768), 0, 0, out, reps, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 768 * 4);
  const int reps = 2000;   // 16000 MFMAs a wave, 48000 a SIMD
  printf("12 waves a CU (3 a SIMD), %d x (8 MFMA + K v_fma_f32) a wave; us\n", reps);
  printf("K VALU per 8 MFMA |  f32 16x16x4 | bf16 16x16x32 | VALU only\n");
#define ROW(K) printf("%17d | %12.1f | %13.1f | %9.1f\n", K, run<K, 0>(out, reps), run<K, 1>(out, reps), run<K, 2>(out, reps));
  ROW(0) ROW(8) ROW(16) ROW(32) ROW(64) ROW(128)
  printf("(f32: 48000 MFMAs a SIMD x 32 cycles = 640 us at 2.4 GHz)\n");
  return 0;
}
