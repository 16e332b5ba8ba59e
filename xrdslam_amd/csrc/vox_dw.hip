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
// A block of 4 waves owns chunks of 64 points (persistent, stride = grid) and
// keeps its share of all five products in MFMA accumulators across its chunks
// (v_mfma_f32_16x16x4_f32: A = 16 output features x 4 points of G^T, B = 4
// points x 16 input features): wave w owns output rows 32w..32w+31 of every
// 128-row product (2 x (1 + 8 + 8 + 9) accumulator tiles = 208 registers: one
// wave per SIMD with the whole register file; with 8 thinner waves the
// compiler ran out of registers at 256 and serialised every LDS read with the
// two MFMAs that consume it).  Per layer the chunk's G and A rows are staged
// in LDS (row stride 144 floats: the four point groups of a fragment read fall
// on distinct banks); the stage's rows travel global -> registers -> LDS with
// the loads of the NEXT stage issued before the current contraction, and
// inside a contraction the fragments of K-step s+1 are read before the MFMAs
// of step s.  The one-row products (gs, the three colour rows) and the bias
// sums run on the VALU from the same staged rows.  Rows beyond the live count
// are staged as zeros.  Each block writes its partial to a [blocks, 54276]
// workspace; vox_dw_reduce sums the live blocks into the flat gradient
// (state_dict order).
#include "common.h"
#include "vox_layout.h"

namespace xrd {
namespace {

constexpr int DW_WAVES = 4;
constexpr int DW_THREADS = DW_WAVES * 64;
constexpr int DW_CHUNK = 64;      // points per stage
constexpr int DW_STRIDE = 144;    // LDS row stride (floats)
constexpr int DW_BLOCKS = 256;    // persistent blocks: one per CU
constexpr int DW_LEN = 54276;     // = VoxFlat::LEN
// flat offsets (state_dict order)
constexpr int F_W0 = 0, F_B0 = 2048, F_W1 = 2176, F_B1 = 18560,
              F_WO = 18688, F_BO = 35200, F_WC = 35329, F_BC = 53761,
              F_W4 = 53889, F_B4 = 54273;
static_assert(F_B4 + 3 == DW_LEN, "flat layout");

// fragments of one K-step: the wave's two G tiles and the NT input tiles
template <int NT>
struct Frag {
  float g[2], a[NT];
  template <int KOFF, int I>
  __device__ __forceinline__ void load_a(uint32_t aaddr) {
    if constexpr (I < NT) {
      a[I] = lds_async<KOFF + 64 * I>(aaddr);
      load_a<KOFF, I + 1>(aaddr);
    }
  }
  // KOFF: byte offset of the K-step relative to the address registers
  template <int KOFF>
  __device__ __forceinline__ void load(uint32_t gaddr, uint32_t aaddr) {
    g[0] = lds_async<KOFF>(gaddr);
    g[1] = lds_async<KOFF + 64>(gaddr);
    load_a<KOFF, 0>(aaddr);
  }
  template <int PENDING>
  __device__ __forceinline__ void landed() {
    lds_landed<PENDING>();
    lds_tie(g[0]);
    lds_tie(g[1]);
#pragma unroll
    for (int it = 0; it < NT; ++it) lds_tie(a[it]);
  }
  __device__ __forceinline__ void mma(f32x4 (*acc)[NT]) const {
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      acc[0][it] = XRD_MFMA4(g[0], a[it], acc[0][it]);
      acc[1][it] = XRD_MFMA4(g[1], a[it], acc[1][it]);
    }
  }
};
// acc[t][it] += G^T(tile 2 wave + t) x A(input tile it) over the 64 staged rows
template <int NT>
__device__ __forceinline__ void contract(const float* __restrict__ g,
                                         const float* __restrict__ a, int wave,
                                         int lane, f32x4 (*acc)[NT]) {
  constexpr int KSTEP = 4 * DW_STRIDE * 4;      // bytes between K-steps
  const int k = lane >> 4, j = lane & 15;
  uint32_t gaddr = lds_addr(g + k * DW_STRIDE + 32 * wave + j);
  uint32_t aaddr = lds_addr(a + k * DW_STRIDE + j);
  Frag<NT> f0, f1;
  f0.template load<0>(gaddr, aaddr);
#pragma unroll 1
  for (int ks = 0; ks < DW_CHUNK / 4 - 2; ks += 2) {
    f1.template load<KSTEP>(gaddr, aaddr);
    f0.template landed<NT + 2>();     // f1's reads stay in flight
    f0.mma(acc);
    f0.template load<2 * KSTEP>(gaddr, aaddr);
    f1.template landed<NT + 2>();
    f1.mma(acc);
    gaddr += 2 * KSTEP;
    aaddr += 2 * KSTEP;
  }
  f1.template load<KSTEP>(gaddr, aaddr);
  f0.template landed<NT + 2>();
  f0.mma(acc);
  f1.template landed<0>();
  f1.mma(acc);
}

// column sums of the staged G rows: thread (col = t & 127, grp = t >> 7) adds
// its 32 rows
__device__ __forceinline__ float colsum32(const float* __restrict__ g) {
  const int col = threadIdx.x & 127, grp = threadIdx.x >> 7;
  float s = 0.f;
#pragma unroll 8
  for (int r = 0; r < 32; ++r) s += g[(grp * 32 + r) * DW_STRIDE + col];
  return s;
}

__global__ __launch_bounds__(DW_THREADS, 1) void vox_dw_kernel(
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
  f32x4 a0[2][1], a1[2][8], ao[2][8], ac[2][9];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    a0[t][0] = z4;
#pragma unroll
    for (int i = 0; i < 8; ++i) a1[t][i] = ao[t][i] = z4;
#pragma unroll
    for (int i = 0; i < 9; ++i) ac[t][i] = z4;
  }
  float b0 = 0.f, b1 = 0.f, bo = 0.f, bc = 0.f;     // bias column sums
  float r_gs = 0.f, r4[3] = {0.f, 0.f, 0.f};       // one-row products
  f32x4 bsv = {0.f, 0.f, 0.f, 0.f};   // gc3 column sums: row t of every chunk
  // (loads are unconditional on a clamped row — no branch per load — and rows
  // beyond the live count become zeros when they are stored to LDS)
  f32x4 vg[8], va[8], vx, vs;     // G rows, A rows (128 wide), x / gc3 rows
  const int64_t last = n - 1;     // n > 0 here
  auto fetch128 = [&](f32x4* v, const float* __restrict__ src, int64_t p0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = threadIdx.x + u * DW_THREADS;
      const int c = (i & 31) << 2;
      int64_t row = p0 + (i >> 5);
      row = row < last ? row : last;
      v[u] = *reinterpret_cast<const f32x4*>(src + row * 128 + c);
    }
  };
  auto store128 = [&](float* __restrict__ dst, const f32x4* v, int64_t p0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = threadIdx.x + u * DW_THREADS;
      *reinterpret_cast<f32x4*>(dst + (i >> 5) * DW_STRIDE + ((i & 31) << 2)) =
          p0 + (i >> 5) < n ? v[u] : z4;
    }
  };
  // x rows (16 wide): one quad per thread
  auto fetch_x = [&](int64_t p0) {
    const int c = (threadIdx.x & 3) << 2;
    int64_t row = p0 + (threadIdx.x >> 2);
    row = row < last ? row : last;
    vx = *reinterpret_cast<const f32x4*>(sx + row * 16 + c);
  };
  auto store_x = [&](int col0, int64_t p0) {
    const int r = threadIdx.x >> 2, c = (threadIdx.x & 3) << 2;
    *reinterpret_cast<f32x4*>(A + r * DW_STRIDE + col0 + c) =
        p0 + r < n ? vx : z4;
  };
  auto fetch_l0 = [&](int64_t p0) {
    fetch128(vg, gh1, p0);
    fetch_x(p0);
    int64_t row = p0 + (threadIdx.x & (DW_CHUNK - 1));
    row = row < last ? row : last;
    vs = *reinterpret_cast<const f32x4*>(gc3 + row * 4);
  };
  fetch_l0((int64_t)blockIdx.x * DW_CHUNK);
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t p0 = ch * DW_CHUNK;
    // ---- layer 0: gh1^T x ------------------------------------------------
    __syncthreads();
    store128(G, vg, p0);
    store_x(0, p0);
    if (threadIdx.x < DW_CHUNK) {
      const f32x4 row = p0 + threadIdx.x < n ? vs : z4;
      *reinterpret_cast<f32x4*>(S + threadIdx.x * 4) = row;
      bsv += row;
    }
    __syncthreads();
    fetch128(vg, gh2, p0);
    fetch128(va, sh1, p0);
    contract<1>(G, A, wave, lane, a0);
    b0 += colsum32(G);
    // ---- layer 1: gh2^T h1 -------------------------------------------------
    __syncthreads();
    store128(G, vg, p0);
    store128(A, va, p0);
    __syncthreads();
    fetch128(vg, gf, p0);
    fetch128(va, sh2, p0);
    contract<8>(G, A, wave, lane, a1);
    b1 += colsum32(G);
    // ---- sdf_out: [gs | gf]^T h2 ---------------------------------------------
    __syncthreads();
    store128(G, vg, p0);
    store128(A, va, p0);
    __syncthreads();
    fetch128(vg, ghc, p0);
    fetch128(va, sf, p0);
    fetch_x(p0);
    contract<8>(G, A, wave, lane, ao);
    bo += colsum32(G);
#pragma unroll 8
    for (int r = 0; r < 32; ++r)
      r_gs = fmaf(S[(grp * 32 + r) * 4 + 3],
                  A[(grp * 32 + r) * DW_STRIDE + col], r_gs);
    // ---- colour layer 0: ghc^T [f | x] ------------------------------------------
    __syncthreads();
    store128(G, vg, p0);
    store128(A, va, p0);
    store_x(128, p0);
    __syncthreads();
    fetch128(va, shc, p0);
    contract<9>(G, A, wave, lane, ac);
    bc += colsum32(G);
    // ---- colour layer 1: g3^T hc (three rows, VALU) ---------------------------
    __syncthreads();
    store128(A, va, p0);
    __syncthreads();
    if (ch + gridDim.x < nchunks) fetch_l0((ch + gridDim.x) * DW_CHUNK);
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const float h = A[(grp * 32 + r) * DW_STRIDE + col];
#pragma unroll
      for (int o = 0; o < 3; ++o)
        r4[o] = fmaf(S[(grp * 32 + r) * 4 + o], h, r4[o]);
    }
  }
  // ---- this block's partial ------------------------------------------------------
  float* out = partial + (int64_t)blockIdx.x * DW_LEN;
  const int q = lane >> 4, j = lane & 15;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 32 * wave + 16 * t + 4 * q + r;
      out[F_W0 + o * 16 + j] = a0[t][0][r];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        out[F_W1 + o * 128 + 16 * it + j] = a1[t][it][r];
        out[F_WO + (1 + o) * 128 + 16 * it + j] = ao[t][it][r];
      }
#pragma unroll
      for (int it = 0; it < 9; ++it)
        out[F_WC + o * 144 + 16 * it + j] = ac[t][it][r];
    }
  }
  // VALU sums: 2 row groups -> one value per column, through LDS
  __syncthreads();
  float* R = G;   // [8 quantities][2 groups][128]
  const float vals[8] = {b0, b1, bo, bc, r_gs, r4[0], r4[1], r4[2]};
#pragma unroll
  for (int k = 0; k < 8; ++k) R[(k * 2 + grp) * 128 + col] = vals[k];
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float t = wave_sum(bsv[c]);
      if (lane == 0) S[c] = t;
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      t[k] = R[(k * 2 + 0) * 128 + col] + R[(k * 2 + 1) * 128 + col];
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
    hipLaunchKernelGGL(vox_dw_kernel, dim3(nb), dim3(DW_THREADS), lds, st,
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
