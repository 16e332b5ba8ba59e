// Vox-Fusion decoder weight gradients for gfx950: the five contractions over
// the sample points
//   dW0 [128,16]  = gh1^T x        dW1 [128,128] = gh2^T h1
//   dWo [129,128] = [gs|gf]^T h2   dWc [128,144] = ghc^T [f|x]
//   dW4 [3,128]   = g3^T hc        + the five bias gradients (column sums)
// on the operands xrd_vox_points_fwd / _bwd leave in HBM (autograd of
// slam/model_components/decoder_voxfusion.py:123-149).  K (the points) is the
// long dimension — 1e4..1e6 against 128 x 128 outputs — which library GEMMs
// handle badly (rocBLAS: 4.8 ms for the five at 590 000 capacity rows, plus
// ~1 ms per column sum), and the LIVE point count is only known on the device.
//
// A block of 8 waves owns chunks of 64 points (persistent, stride = grid) and
// keeps its share of all five products in MFMA accumulators across its chunks
// (v_mfma_f32_16x16x4_f32: A = 16 output features x 4 points of G^T, B = 4
// points x 16 input features): wave w owns output rows 16w..16w+15 of every
// 128-row product (1 + 8 + 8 + 9 accumulator tiles).  Per layer the chunk's G
// and A rows are staged in LDS (row stride 144 floats: the four point groups
// of a fragment read fall on distinct banks).  The one-row products (gs, the
// three colour rows) and the bias sums run on the VALU from the same staged
// rows.  Rows beyond the live count are staged as zeros.  Each block writes
// its partial to a [blocks, 54276] workspace; vox_dw_reduce sums the live
// blocks into the flat gradient (state_dict order).
#include "common.h"
#include "vox_layout.h"

namespace xrd {
namespace {

constexpr int DW_WAVES = 8;
constexpr int DW_CHUNK = 64;      // points per stage
constexpr int DW_STRIDE = 144;    // LDS row stride (floats)
constexpr int DW_BLOCKS = 256;    // persistent blocks: one per CU
constexpr int DW_LEN = 54276;     // = VoxFlat::LEN
// flat offsets (state_dict order)
constexpr int F_W0 = 0, F_B0 = 2048, F_W1 = 2176, F_B1 = 18560,
              F_WO = 18688, F_BO = 35200, F_WC = 35329, F_BC = 53761,
              F_W4 = 53889, F_B4 = 54273;
static_assert(F_B4 + 3 == DW_LEN, "flat layout");

// rows [p0, p0+64) x width floats of src ([P, width] row-major) -> LDS rows of
// stride DW_STRIDE at column col0; rows >= n are zeros
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, int col0,
                                           const float* __restrict__ src,
                                           int width, int64_t p0, int64_t n) {
  const int w4 = width >> 2;
  for (int i = threadIdx.x; i < DW_CHUNK * w4; i += DW_WAVES * 64) {
    const int r = i / w4, c = (i - r * w4) << 2;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p0 + r < n)
      v = *reinterpret_cast<const f32x4*>(src + (p0 + r) * width + c);
    *reinterpret_cast<f32x4*>(dst + r * DW_STRIDE + col0 + c) = v;
  }
}

// acc[it] += G^T(tile of this wave) x A(input tile it) over the 64 staged rows
template <int NT>
__device__ __forceinline__ void contract(const float* __restrict__ g,
                                         const float* __restrict__ a, int wave,
                                         int lane, f32x4* acc) {
  const int k = lane >> 4, j = lane & 15;
#pragma unroll 2
  for (int ks = 0; ks < DW_CHUNK / 4; ++ks) {
    const int row = (4 * ks + k) * DW_STRIDE;
    const float ga = g[row + 16 * wave + j];
#pragma unroll
    for (int it = 0; it < NT; ++it)
      acc[it] = XRD_MFMA4(ga, a[row + 16 * it + j], acc[it]);
  }
}

// column sums of the staged G rows: thread (col = t & 127, grp = t >> 7) adds
// its 16 rows
__device__ __forceinline__ float colsum16(const float* __restrict__ g) {
  const int col = threadIdx.x & 127, grp = threadIdx.x >> 7;
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += g[(grp * 16 + r) * DW_STRIDE + col];
  return s;
}

__global__ __launch_bounds__(DW_WAVES * 64) void vox_dw_kernel(
    int64_t p_cap, const int* __restrict__ n_dev,
    const float* __restrict__ sx, const float* __restrict__ sh1,
    const float* __restrict__ sh2, const float* __restrict__ sf,
    const float* __restrict__ shc, const float* __restrict__ gc3,
    const float* __restrict__ ghc, const float* __restrict__ gf,
    const float* __restrict__ gh2, const float* __restrict__ gh1,
    float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* G = reinterpret_cast<float*>(smem_raw);        // [64][144]
  float* A = G + DW_CHUNK * DW_STRIDE;                   // [64][144]
  float* S = A + DW_CHUNK * DW_STRIDE;                   // [64][4] gc3 rows
  int64_t n = p_cap;
  if (n_dev != nullptr) n = *n_dev < n ? (*n_dev > 0 ? *n_dev : 0) : n;
  const int64_t nchunks = (n + DW_CHUNK - 1) / DW_CHUNK;
  if ((int64_t)blockIdx.x >= nchunks) return;   // uniform per block
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = threadIdx.x & 127, grp = threadIdx.x >> 7;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 a0[1] = {z4}, a1[8], ao[8], ac[9];
#pragma unroll
  for (int i = 0; i < 8; ++i) a1[i] = ao[i] = z4;
#pragma unroll
  for (int i = 0; i < 9; ++i) ac[i] = z4;
  float b0 = 0.f, b1 = 0.f, bo = 0.f, bc = 0.f;     // bias column sums
  float r_gs = 0.f, r4[3] = {0.f, 0.f, 0.f};       // one-row products
  float bs = 0.f;                                   // sums of gc3 columns
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t p0 = ch * DW_CHUNK;
    // ---- layer 0: gh1^T x ------------------------------------------------
    __syncthreads();
    stage_rows(G, 0, gh1, 128, p0, n);
    stage_rows(A, 0, sx, 16, p0, n);
    if (threadIdx.x < DW_CHUNK) {
      f32x4 v = z4;
      if (p0 + threadIdx.x < n)
        v = *reinterpret_cast<const f32x4*>(gc3 + (p0 + threadIdx.x) * 4);
      *reinterpret_cast<f32x4*>(S + threadIdx.x * 4) = v;
    }
    __syncthreads();
    contract<1>(G, A, wave, lane, a0);
    b0 += colsum16(G);
    if (threadIdx.x < 4)
      for (int r = 0; r < DW_CHUNK; ++r) bs += S[r * 4 + threadIdx.x];
    // ---- layer 1: gh2^T h1 -------------------------------------------------
    __syncthreads();
    stage_rows(G, 0, gh2, 128, p0, n);
    stage_rows(A, 0, sh1, 128, p0, n);
    __syncthreads();
    contract<8>(G, A, wave, lane, a1);
    b1 += colsum16(G);
    // ---- sdf_out: [gs | gf]^T h2 ---------------------------------------------
    __syncthreads();
    stage_rows(G, 0, gf, 128, p0, n);
    stage_rows(A, 0, sh2, 128, p0, n);
    __syncthreads();
    contract<8>(G, A, wave, lane, ao);
    bo += colsum16(G);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      r_gs = fmaf(S[(grp * 16 + r) * 4 + 3],
                  A[(grp * 16 + r) * DW_STRIDE + col], r_gs);
    // ---- colour layer 0: ghc^T [f | x] ------------------------------------------
    __syncthreads();
    stage_rows(G, 0, ghc, 128, p0, n);
    stage_rows(A, 0, sf, 128, p0, n);
    stage_rows(A, 128, sx, 16, p0, n);
    __syncthreads();
    contract<9>(G, A, wave, lane, ac);
    bc += colsum16(G);
    // ---- colour layer 1: g3^T hc (three rows, VALU) ---------------------------
    __syncthreads();
    stage_rows(A, 0, shc, 128, p0, n);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float h = A[(grp * 16 + r) * DW_STRIDE + col];
#pragma unroll
      for (int o = 0; o < 3; ++o)
        r4[o] = fmaf(S[(grp * 16 + r) * 4 + o], h, r4[o]);
    }
  }
  // ---- this block's partial ------------------------------------------------------
  float* out = partial + (int64_t)blockIdx.x * DW_LEN;
  const int q = lane >> 4, j = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 16 * wave + 4 * q + r;
    out[F_W0 + o * 16 + j] = a0[0][r];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      out[F_W1 + o * 128 + 16 * it + j] = a1[it][r];
      out[F_WO + (1 + o) * 128 + 16 * it + j] = ao[it][r];
    }
#pragma unroll
    for (int it = 0; it < 9; ++it)
      out[F_WC + o * 144 + 16 * it + j] = ac[it][r];
  }
  // VALU sums: 4 row groups -> one value per column, through LDS
  __syncthreads();
  float* R = G;   // [8 quantities][4 groups][128]
  const float vals[8] = {b0, b1, bo, bc, r_gs, r4[0], r4[1], r4[2]};
#pragma unroll
  for (int k = 0; k < 8; ++k) R[(k * 4 + grp) * 128 + col] = vals[k];
  if (threadIdx.x < 4) S[threadIdx.x] = bs;
  __syncthreads();
  if (threadIdx.x < 128) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      t[k] = R[(k * 4 + 0) * 128 + col] + R[(k * 4 + 1) * 128 + col] +
             R[(k * 4 + 2) * 128 + col] + R[(k * 4 + 3) * 128 + col];
    out[F_B0 + col] = t[0];
    out[F_B1 + col] = t[1];
    out[F_BO + 1 + col] = t[2];
    out[F_BC + col] = t[3];
    out[F_WO + col] = t[4];            // row 0 of sdf_out: the sdf itself
    out[F_W4 + col] = t[5];
    out[F_W4 + 128 + col] = t[6];
    out[F_W4 + 256 + col] = t[7];
  }
  if (threadIdx.x < 3) out[F_B4 + threadIdx.x] = S[threadIdx.x];
  if (threadIdx.x == 3) out[F_BO] = S[3];
}

// flat[i] = sum over the live blocks of partial[b][i]: a block takes 64
// columns x 4 groups of partials (independent chains), combined through LDS
__global__ __launch_bounds__(256) void vox_dw_reduce_kernel(
    int64_t p_cap, const int* __restrict__ n_dev, int n_blocks,
    const float* __restrict__ partial, float* __restrict__ flat) {
  __shared__ float red[4][64];
  int64_t n = p_cap;
  if (n_dev != nullptr) n = *n_dev < n ? (*n_dev > 0 ? *n_dev : 0) : n;
  const int64_t nchunks = (n + DW_CHUNK - 1) / DW_CHUNK;
  const int live = nchunks < n_blocks ? (int)nchunks : n_blocks;
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < DW_LEN) {
    int b = grp;
    for (; b + 12 < live; b += 16) {
      s0 += partial[(int64_t)b * DW_LEN + i];
      s1 += partial[(int64_t)(b + 4) * DW_LEN + i];
      s2 += partial[(int64_t)(b + 8) * DW_LEN + i];
      s3 += partial[(int64_t)(b + 12) * DW_LEN + i];
    }
    for (; b < live; b += 4) s0 += partial[(int64_t)b * DW_LEN + i];
  }
  red[grp][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && i < DW_LEN)
    flat[i] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int64_t xrd_vox_dw_ws_floats(void) { return (int64_t)DW_BLOCKS * DW_LEN; }

int xrd_vox_dw(int64_t n_points, const int32_t* n_points_dev,
               const float* save_x, const float* save_h1, const float* save_h2,
               const float* save_f, const float* save_hc, const float* g_c3,
               const float* g_hc, const float* g_f, const float* g_h2,
               const float* g_h1, float* workspace, float* g_flat,
               xrd_stream_t stream) {
  static_assert(DW_LEN == VoxFlat::LEN, "flat decoder length");
  if (n_points < 0) return XRD_ERR_ARG;
  if (!g_flat || !workspace) return XRD_ERR_ARG;
  if (n_points > 0 && (!save_x || !save_h1 || !save_h2 || !save_f ||
                       !save_hc || !g_c3 || !g_hc || !g_f || !g_h2 || !g_h1))
    return XRD_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(2 * DW_CHUNK * DW_STRIDE + DW_CHUNK * 4) *
                     sizeof(float);
  static bool ready = false;
  if (!ready) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(vox_dw_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute");
    ready = true;
  }
  int nb = DW_BLOCKS;
  if (n_points > 0) {
    const int64_t nchunks = (n_points + DW_CHUNK - 1) / DW_CHUNK;
    if (nchunks < nb) nb = (int)nchunks;
    hipLaunchKernelGGL(vox_dw_kernel, dim3(nb), dim3(DW_WAVES * 64), lds, st,
                       n_points, n_points_dev, save_x, save_h1, save_h2,
                       save_f, save_hc, g_c3, g_hc, g_f, g_h2, g_h1,
                       workspace);
  }
  hipLaunchKernelGGL(vox_dw_reduce_kernel, dim3((DW_LEN + 63) / 64),
                     dim3(256), 0, st, n_points, n_points_dev, nb, workspace,
                     g_flat);
  return check_launch("xrd_vox_dw");
}

}  // extern "C"
