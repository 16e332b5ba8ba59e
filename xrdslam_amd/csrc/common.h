// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xrdslam_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) * B(4x16), exact f32 fma chain.
// lane l: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D reg r: row=(l>>4)*4+r col=l&15
#define XRD_MFMA4(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace xrd {

extern thread_local const char* g_last_error;
int check_launch(const char* what);

// Hash-table gradient scatter (encodings.hip): a block takes one 8192-entry
// chunk of one level and a slice of the points, accumulates the contributions
// that fall into its chunk in LDS and adds the chunk to dparams with coalesced
// atomics (accumulate == false: dparams is zeroed first).  dy: per point and level one float2 at
// dy[p * point_stride + l * level_stride].
// zero-fill as a kernel node: hipMemsetAsync nodes inside a captured hipGraph
// were observed to race with neighbouring kernel nodes on replay (ROCm 7.2)
int zero_floats(float* p, size_t n, void* stream);

int launch_hash_chunk_scatter(int n_levels, const float* scales,
                              const uint32_t* res, const uint32_t* sizes,
                              const uint32_t* offsets, int64_t n_points,
                              const float* x, const float* dy,
                              int64_t point_stride, int64_t level_stride,
                              float* dparams, bool accumulate, void* stream);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Wave reduction on DPP (full-rate VALU, no LDS crossbar): quad swaps, row
// half-mirror / mirror, then row broadcasts; the TOTAL is returned to every
// lane through a scalar read of lane 63.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define XRD_DPP_ADD(CTRL, ROWMASK)                                            \
  v += __builtin_bit_cast(                                                    \
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, \
                                         ROWMASK, 0xf, false))
  XRD_DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
  XRD_DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
  XRD_DPP_ADD(0x141, 0xf);  // row_half_mirror
  XRD_DPP_ADD(0x140, 0xf);  // row_mirror: every lane holds its row's sum
  XRD_DPP_ADD(0x142, 0xa);  // row_bcast:15 -> rows 1 and 3
  XRD_DPP_ADD(0x143, 0xc);  // row_bcast:31 -> rows 2 and 3
#undef XRD_DPP_ADD
  return __builtin_bit_cast(
      float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// sum over the 4 lane groups (l>>4) that hold the same point (l&15)
__device__ __forceinline__ float group4_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
// sum over the 16 lanes of a row (same l>>4) on DPP (full-rate VALU, no LDS
// crossbar round trips): every lane of the row gets the row's sum
__device__ __forceinline__ float row16_sum_dpp(float v) {
#define XRD_DPP_ADD(CTRL)                                                     \
  v += __builtin_bit_cast(                                                    \
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, \
                                         0xf, 0xf, false))
  XRD_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
  XRD_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
  XRD_DPP_ADD(0x141);  // row_half_mirror
  XRD_DPP_ADD(0x140);  // row_mirror
#undef XRD_DPP_ADD
  return v;
}
// sum over the 16 lanes of a row (same l>>4)
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// LDS fragment reads of an MFMA contraction as volatile asm: left to itself
// the compiler sinks every fragment read next to the MFMAs that consume it
// (read - wait - MFMAs: one exposed LDS latency per group); pinned, the reads
// of K-step s+1 are in flight while the MFMAs of step s issue.  lds_landed()
// is the matching wait; lds_tie() makes a value's consumers depend on it.
template <int OFF>
__device__ __forceinline__ float lds_async(uint32_t addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// LDS reads return in order: PENDING = reads issued after the ones waited for
template <int PENDING>
__device__ __forceinline__ void lds_landed() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PENDING) : "memory");
}
__device__ __forceinline__ void lds_tie(float& v) {
  asm volatile("" : "+v"(v));
}
__device__ __forceinline__ uint32_t lds_addr(const float* p) {
  return (uint32_t)(uintptr_t)(
      const __attribute__((address_space(3))) float*)p;
}

// one LDS-DMA load of 1 KB: lane l's 16 bytes at base + voff(l) land at
// lds_base + 16 l, no VGPR round trip.  Inline asm on purpose: the compiler
// does not track the load, so it puts no vmcnt(0) in front of LDS reads of
// OTHER buffers (with the builtin every ds_read behind an outstanding LDS-DMA
// load waits for it); lds_dma_landed() is the wait.  base, lds_base:
// wave-uniform.
__device__ __forceinline__ void lds_dma16(const float* base, uint32_t voff,
                                          uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2"
               :
               : "s"(lds_base), "v"(voff), "s"(base)
               : "memory");
}
__device__ __forceinline__ void lds_dma_landed() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// this wave's share of a block-wide copy src[0, N) -> LDS byte address dst:
// 1 KB chunks dealt over the NW waves, the partial last chunk by a lane mask.
// Visible to the block after lds_dma_landed() + a barrier.
template <int NW, int N>
__device__ __forceinline__ void lds_dma_issue(const float* __restrict__ src,
                                              uint32_t dst, int wave,
                                              int lane) {
  static_assert(N % 4 == 0, "16-byte lanes");
  constexpr int FULL = N / 256, TAIL = N % 256;
  const uint32_t voff = (uint32_t)lane * 16u;
#pragma unroll
  for (int k = 0; k < (FULL + NW - 1) / NW; ++k) {
    const int c = wave + NW * k;   // wave-uniform
    if (c < FULL) lds_dma16(src + c * 256, voff, dst + (uint32_t)c * 1024u);
  }
  if (TAIL != 0 && wave == FULL % NW && lane * 4 < TAIL)
    lds_dma16(src + FULL * 256, voff, dst + (uint32_t)FULL * 1024u);
}

// make LDS writes of this wave visible to its own later LDS reads
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// sin/cos with a 3-constant Cody-Waite reduction (fma): < 1e-7 absolute error
// for |x| < 1e5 (checked against float64 in tests/test_host_math.py).  The
// Fourier-feature arguments p.B reach ~1e3 rad (B ~ N(0, 25^2),
// decoder_nice.py:20-38), so the fast __sinf is not usable, while ocml's sinf
// carries a Payne-Hanek path that costs ~10x the registers/instructions.
__device__ __forceinline__ void sincos_cw(float x, float& sn, float& cs) {
  const float k = rintf(x * 0.636619772367581f);
  float r = fmaf(k, -1.57079601e+00f, x);
  r = fmaf(k, -3.13916473e-07f, r);
  r = fmaf(k, -5.39030253e-15f, r);
  const float s = r * r;
  float ps = fmaf(s, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(s, ps, -1.6666654611e-1f);
  const float sr = fmaf(r * s, ps, r);
  float pc = fmaf(s, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(s, pc, 4.166664568298827e-2f);
  const float cr = fmaf(s * s, pc, fmaf(s, -0.5f, 1.0f));
  const int n = (int)k & 3;
  const float a = (n & 1) ? cr : sr;
  const float b = (n & 1) ? sr : cr;
  sn = (n & 2) ? -a : a;
  cs = ((n + 1) & 2) ? -b : b;
}
// sin_cw(x) in four stages of ~7 VALU instructions each (the same operations
// in the same order as sincos_cw: bit-identical), so that a caller can place
// one MFMA between two stages — a K-step of the Fourier embedding is one sine
// and 4 MFMAs (nice_device.h: mlp_fwd_ra)
struct SinStages {
  float k, r, s, ps, pc, sr, cr;
};
__device__ __forceinline__ void sin_stage_a(float x, SinStages& t) {
  t.k = rintf(x * 0.636619772367581f);
  float r = fmaf(t.k, -1.57079601e+00f, x);
  r = fmaf(t.k, -3.13916473e-07f, r);
  t.r = fmaf(t.k, -5.39030253e-15f, r);
}
__device__ __forceinline__ void sin_stage_b(SinStages& t) {
  t.s = t.r * t.r;
  float ps = fmaf(t.s, -1.9515295891e-4f, 8.3321608736e-3f);
  t.ps = fmaf(t.s, ps, -1.6666654611e-1f);
  float pc = fmaf(t.s, 2.443315711809948e-5f, -1.388731625493765e-3f);
  t.pc = fmaf(t.s, pc, 4.166664568298827e-2f);
}
__device__ __forceinline__ void sin_stage_c(SinStages& t) {
  t.sr = fmaf(t.r * t.s, t.ps, t.r);
  t.cr = fmaf(t.s * t.s, t.pc, fmaf(t.s, -0.5f, 1.0f));
}
__device__ __forceinline__ float sin_stage_d(const SinStages& t) {
  const int n = (int)t.k & 3;
  const float a = (n & 1) ? t.cr : t.sr;
  return (n & 2) ? -a : a;
}
__device__ __forceinline__ float sin_cw(float x) {
  float s, c;
  sincos_cw(x, s, c);
  return s;
}
__device__ __forceinline__ float cos_cw(float x) {
  float s, c;
  sincos_cw(x, s, c);
  return c;
}

}  // namespace xrd
