// Stand-alone check of the transposed 12-value wave reduction of csrc/gs_blend.hip
// (v_permlane32_swap / v_permlane16_swap through inline asm + DPP): prints the
// reference sums and what every lane holds.  An earlier version of this file
// used __builtin_amdgcn_permlane{16,32}_swap and showed that this toolchain
// returns the FIRST register in both elements of the result pair.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/t tools/permlane_merge_check.hip && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// six / three independent swaps back to back; the s_nop cover the VALU ->
// permlane-swap and permlane-swap -> VALU wait states (inline asm is opaque
// to the hazard recogniser)
__device__ __forceinline__ void swap32x6(float (&v)[12]) {
  asm volatile(
      "s_nop 1\n"
      "v_permlane32_swap_b32 %0, %1\n"
      "v_permlane32_swap_b32 %2, %3\n"
      "v_permlane32_swap_b32 %4, %5\n"
      "v_permlane32_swap_b32 %6, %7\n"
      "v_permlane32_swap_b32 %8, %9\n"
      "v_permlane32_swap_b32 %10, %11\n"
      "s_nop 1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]),
        "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
        "+v"(v[10]), "+v"(v[11]));
}
__device__ __forceinline__ void swap16x3(float (&u)[6]) {
  asm volatile(
      "s_nop 1\n"
      "v_permlane16_swap_b32 %0, %1\n"
      "v_permlane16_swap_b32 %2, %3\n"
      "v_permlane16_swap_b32 %4, %5\n"
      "s_nop 1"
      : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]),
        "+v"(u[5]));
}
__global__ void k(float* out, float* o32, float* o16, float* oror, float* ohm) {
  int lane = threadIdx.x;
  float v[12];
  for (int i = 0; i < 12; ++i) v[i] = (float)((lane + 1) * (i + 1) % 17);
  o32[lane] = 0; o16[lane] = 0;
  oror[lane] = dpp_get<0x128>((float)lane);
  ohm[lane] = dpp_get<0x141>((float)lane);
  float u[6], t[3];
  swap32x6(v);
  for (int kk = 0; kk < 6; ++kk) u[kk] = v[2 * kk] + v[2 * kk + 1];
  swap16x3(u);
  for (int kk = 0; kk < 3; ++kk) t[kk] = u[2 * kk] + u[2 * kk + 1];
  const bool b3 = lane & 8, b2 = lane & 4;
  const float w0 = (b3 ? t[1] : t[0]) + dpp_get<0x128>(b3 ? t[0] : t[1]);
  const float w1 = t[2] + dpp_get<0x128>(t[2]);
  float z = (b2 ? w1 : w0) + dpp_get<0x141>(b2 ? w0 : w1);
  z += dpp_get<0xB1>(z);
  z += dpp_get<0x4E>(z);
  out[lane] = z;
}
int main() {
  float *d; hipMalloc(&d, 5 * 64 * 4);
  k<<<1, 64>>>(d, d + 64, d + 128, d + 192, d + 256);
  float h[5 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  float ref[12] = {0};
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 12; ++i) ref[i] += (float)((l + 1) * (i + 1) % 17);
  printf("ref:"); for (int i = 0; i < 12; ++i) printf(" %g", ref[i]); printf("\n");
  printf("out:"); for (int l = 0; l < 64; ++l) printf(" %g", h[l]); printf("\n");
  printf("m32:"); for (int l = 0; l < 64; ++l) printf(" %g", h[64 + l]); printf("\n");
  printf("m16:"); for (int l = 0; l < 64; ++l) printf(" %g", h[128 + l]); printf("\n");
  printf("ror8:"); for (int l = 0; l < 64; ++l) printf(" %g", h[192 + l]); printf("\n");
  printf("hm:"); for (int l = 0; l < 64; ++l) printf(" %g", h[256 + l]); printf("\n");
}
