// Vox-Fusion: fused "voxel features + decoder" for gfx950 (MI355X).
//
//   sample point (xyz, leaf voxel id)
//     -> 8 vertex ids of the voxel -> 8 embedding rows [16] -> trilinear
//        feature x                       (voxel_helpers_voxfusion.py:97-123)
//     -> decoder 16 -> 128 -> 128 -> (sdf, f[128]); [f, x] -> 128 -> rgb
//                                        (decoder_voxfusion.py:123-149)
// as one kernel forward and one backward.  One wave = 16 points; every layer
// is an in-register chain of v_mfma_f32_16x16x4_f32 (exact f32): rows = output
// features (8 tiles of 16), columns = the 16 points, the accumulators of a
// layer are the B operand of the next (nice_layout.h).
//
// A block = 16 waves = one CU.  The 16-point tiles are dealt EVENLY over the
// blocks (block b: tiles [b T / B, (b+1) T / B), a round = 16 tiles, the
// waves beyond a block's last tile skip the arithmetic), so a SIMD gets
// ceil(T / 1024) tiles or one more — with whole 128-point groups per block
// (rounds 1-5) a mapping iteration's 4 570 tiles put 6 or 8 on some SIMDs and
// 4 on most.  A layer's fragments (<= 76 KB) reach LDS by LDS-DMA loads
// (global_load_lds_dwordx4: no register round trip) into one of TWO buffers:
// the next layer streams in while this one computes — 4 barriers a round
// forward, 5 backward, none of them behind an exposed L2 round trip (rounds
// 1-5: 8 / 9 barriers, each stage loaded between two of them).
//
// The backward returns d loss / d xyz (pose gradient), scatters the embedding
// gradient with atomics and writes the per-point operands of the decoder's
// weight gradients ([P,128] matrices) for csrc/vox_dw.hip — the 54 276
// weight gradients would need 848 accumulator registers a lane to stay in the
// kernel.
//
// Reference behaviour restated, never copied; parity: tests/test_vox_hip.py
// (torch fp32 modules) and the reference-made golden of the whole model
// (tests/test_voxfusion_hip.py).
#include <hip/hip_runtime.h>

#include "common.h"
#include "vox_layout.h"

namespace xrd {
namespace {

constexpr int VW = 16;           // waves (16-point tiles) per block
constexpr int kVoxBlocks = 256;  // one block a CU (142 KB of LDS)
// per-wave LDS scratch of the embedding-gradient scatter (backward):
// gt [16 points][17] | row [16][8] | w [16][8]
constexpr int kVoxScatter = 16 * 17 + 16 * 8 + 16 * 8;

constexpr int cmax(int a, int b) { return a > b ? a : b; }
// the two staging buffers (floats).  Forward: A = layers 0+1 (adjacent in the
// packed buffer: one copy), then the colour head; B = sdf_out.  Backward:
// A = colour head, then layer 1, then the scatter scratch; B = sdf_out, then
// layer 0.
constexpr int kBufA =
    cmax(cmax(VoxPack::F0_LEN + VoxPack::F1_LEN, VoxPack::FC_LEN),
         cmax(cmax(VoxPack::RC_LEN, VoxPack::R1_LEN), VW * kVoxScatter));
constexpr int kBufB =
    cmax(VoxPack::FS_LEN, cmax(VoxPack::RS_LEN, VoxPack::R0_LEN));
static_assert(VoxPack::F1 == VoxPack::F0 + VoxPack::F0_LEN, "one copy");
static_assert(kBufA % 4 == 0 && (kBufA + kBufB) * 4 <= 160 * 1024, "LDS");

// acc[jt] += W[16jt.., kin(s)] * in[kin(s)] for K-steps s0 .. s0+KS-1 of a
// layer whose fragments are laid out (jt * KTOT + s); in: D-layout registers
template <int JT, int KTOT, int KS>
__device__ __forceinline__ void dense(const float* __restrict__ w, int lane,
                                      int s0, const f32x4* in, f32x4* acc) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float b = in[s >> 2][s & 3];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
      acc[jt] = XRD_MFMA4(w[(jt * KTOT + s0 + s) * 64 + lane], b, acc[jt]);
  }
}

struct Corner {
  int row[8];    // embedding row of corner c (-1: no voxel)
  float w[8];    // trilinear weights
  float p[3];    // local coordinate
};

// voxel_helpers_voxfusion.py:97-107: p = (xyz - centre)/voxel_size + 0.5,
// corner c = 4 ix + 2 iy + iz selects q_a in {0,1}, weight = prod over axes of
// (p q + (1-p)(1-q)), evaluated in the reference's order ((x * y) * z)
__device__ __forceinline__ void corners(const float* __restrict__ xyz,
                                        const int* __restrict__ vox,
                                        const float* __restrict__ centres,
                                        const int* __restrict__ vertex_idx,
                                        float voxel_size, int64_t pt,
                                        bool valid, Corner& C) {
  int v = valid ? vox[pt] : -1;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    C.row[c] = -1;
    C.w[c] = 0.f;
  }
  C.p[0] = C.p[1] = C.p[2] = 0.f;
  if (v < 0) return;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    C.p[a] = (xyz[pt * 3 + a] - centres[(int64_t)v * 3 + a]) / voxel_size +
             0.5f;
  const int4 lo = *reinterpret_cast<const int4*>(vertex_idx + (int64_t)v * 8);
  const int4 hi =
      *reinterpret_cast<const int4*>(vertex_idx + (int64_t)v * 8 + 4);
  const int ids[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float wx = (c & 4) ? C.p[0] : 1.f - C.p[0];
    const float wy = (c & 2) ? C.p[1] : 1.f - C.p[1];
    const float wz = (c & 1) ? C.p[2] : 1.f - C.p[2];
    C.row[c] = ids[c];
    C.w[c] = (wx * wy) * wz;
  }
}

__device__ __forceinline__ f32x4 relu4(const f32x4 a, uint32_t& bits,
                                       int shift) {
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool on = a[k] > 0.f;
    r[k] = on ? a[k] : 0.f;
    if (on) bits |= 1u << (shift + k);
  }
  return r;
}

// store a D-layout activation (features 16jt+4q+r of point li) to [P][128]
__device__ __forceinline__ void save128(float* __restrict__ dst, int64_t pt,
                                        int q, const f32x4* v) {
#pragma unroll
  for (int jt = 0; jt < 8; ++jt)
    *reinterpret_cast<f32x4*>(dst + pt * 128 + 16 * jt + 4 * q) = v[jt];
}

__global__ __launch_bounds__(VW * 64) void vox_points_fwd_kernel(
    int64_t P, const float* __restrict__ xyz, const int* __restrict__ vox,
    const float* __restrict__ centres, const int* __restrict__ vertex_idx,
    const float* __restrict__ emb, float voxel_size,
    const float* __restrict__ pk, float* __restrict__ sdf,
    float* __restrict__ rgb, float* __restrict__ sx, float* __restrict__ sh1,
    float* __restrict__ sh2, float* __restrict__ sf, float* __restrict__ shc,
    uint32_t* __restrict__ masks, const int* __restrict__ n_dev) {
  using K = VoxPack;
  // static-capacity launches: the live point count comes from the device
  if (n_dev != nullptr) P = *n_dev < P ? (*n_dev > 0 ? *n_dev : 0) : P;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wa = reinterpret_cast<float*>(smem_raw);
  float* wb = wa + kBufA;
  const uint32_t la = lds_addr(wa), lb = lds_addr(wb);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  const int64_t ntiles = (P + 15) / 16;
  const int64_t t0 = ntiles * blockIdx.x / gridDim.x;
  const int64_t t1 = ntiles * (blockIdx.x + 1) / gridDim.x;
  if (t0 >= t1) return;   // uniform per block
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  lds_dma_issue<VW, K::F0_LEN + K::F1_LEN>(pk + K::F0, la, wave, lane);
  lds_dma_issue<VW, K::FS_LEN>(pk + K::FS, lb, wave, lane);
  for (int64_t tb = t0; tb < t1; tb += VW) {
    const bool active = tb + wave < t1;    // wave-uniform
    const bool more = tb + VW < t1;
    const int64_t pt = (tb + wave) * 16 + li;
    const bool valid = active && pt < P;
    f32x4 x[1] = {z4};
    f32x4 h[8], a[8];
    uint32_t m1 = 0, m2 = 0, mc = 0;
    if (active) {
      // -- trilinear voxel feature: lane (q, li) gathers features 4q..4q+3 --
      Corner C;
      corners(xyz, vox, centres, vertex_idx, voxel_size, pt, valid, C);
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (C.row[c] >= 0)
          x[0] += *reinterpret_cast<const f32x4*>(
                      emb + (int64_t)C.row[c] * 16 + 4 * q) * C.w[c];
      if (valid && sx)
        *reinterpret_cast<f32x4*>(sx + pt * 16 + 4 * q) = x[0];
    }
    lds_dma_landed();
    __syncthreads();        // A = layers 0 + 1, B = sdf_out
    if (active) {
      // ---- layer 0 -----------------------------------------------------------
      const float* w0 = wa;
#pragma unroll
      for (int jt = 0; jt < 8; ++jt)
        a[jt] = *reinterpret_cast<const f32x4*>(w0 + (K::B0 - K::F0) +
                                                16 * jt + 4 * q);
      dense<8, 4, 4>(w0 + (K::W0 - K::F0), lane, 0, x, a);
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) h[jt] = relu4(a[jt], m1, 4 * jt);
      if (valid && sh1) save128(sh1, pt, q, h);
      // ---- layer 1 -----------------------------------------------------------
      const float* w1 = wa + K::F0_LEN;
#pragma unroll
      for (int jt = 0; jt < 8; ++jt)
        a[jt] = *reinterpret_cast<const f32x4*>(w1 + (K::B1 - K::F1) +
                                                16 * jt + 4 * q);
      dense<8, 32, 32>(w1 + (K::W1 - K::F1), lane, 0, h, a);
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) h[jt] = relu4(a[jt], m2, 4 * jt);
      if (valid && sh2) save128(sh2, pt, q, h);
    }
    __syncthreads();        // A free
    lds_dma_issue<VW, K::FC_LEN>(pk + K::FC, la, wave, lane);
    if (active) {
      // ---- sdf_out: sdf (row 0, on the VALU) and the sdf feature f ---------
      float s = 0.f;
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(
            wb + (K::WS0 - K::FS) + 16 * jt + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) s = fmaf(w0[r], h[jt][r], s);
        a[jt] = *reinterpret_cast<const f32x4*>(wb + (K::BS - K::FS) +
                                                16 * jt + 4 * q);
      }
      s = group4_sum(s) + wb[K::BS0 - K::FS];
      if (valid && q == 0) sdf[pt] = s;
      dense<8, 32, 32>(wb + (K::WS - K::FS), lane, 0, h, a);
      if (valid && sf) save128(sf, pt, q, a);  // f = a (no activation)
    }
    lds_dma_landed();
    __syncthreads();        // A = colour head, B free
    if (more) lds_dma_issue<VW, K::FS_LEN>(pk + K::FS, lb, wave, lane);
    if (active) {
      // ---- colour head -------------------------------------------------------
#pragma unroll
      for (int jt = 0; jt < 8; ++jt)
        h[jt] = *reinterpret_cast<const f32x4*>(wa + (K::BC - K::FC) +
                                                16 * jt + 4 * q);
      dense<8, 36, 32>(wa + (K::WC - K::FC), lane, 0, a, h);
      dense<8, 36, 4>(wa + (K::WC - K::FC), lane, 32, x, h);
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) h[jt] = relu4(h[jt], mc, 4 * jt);
      if (valid && shc) save128(shc, pt, q, h);
      float col[3];
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        float s = 0.f;
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(
              wa + (K::WO - K::FC) + o * 128 + 16 * jt + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) s = fmaf(w[r], h[jt][r], s);
        }
        s = group4_sum(s) + wa[(K::BO - K::FC) + o];
        col[o] = 1.f / (1.f + expf(-s));
      }
      if (valid) {
        if (q == 0) {
          rgb[pt * 3 + 0] = col[0];
          rgb[pt * 3 + 1] = col[1];
          rgb[pt * 3 + 2] = col[2];
        }
        if (masks) {  // [P][3][4]: ReLU bits of the lane's 32 features
          masks[(pt * 3 + 0) * 4 + q] = m1;
          masks[(pt * 3 + 1) * 4 + q] = m2;
          masks[(pt * 3 + 2) * 4 + q] = mc;
        }
      }
    }
    if (more) {
      __syncthreads();      // A free
      lds_dma_issue<VW, K::F0_LEN + K::F1_LEN>(pk + K::F0, la, wave, lane);
    }
  }
}

__device__ __forceinline__ f32x4 mask4(const f32x4 g, uint32_t bits,
                                       int shift) {
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = ((bits >> (shift + k)) & 1u) ? g[k] : 0.f;
  return r;
}

// DW: the launch also writes the weight-gradient operands (mapping); the
// tracking variant drops that code at compile time (and the two show up under
// their own names in profiles and counter passes)
template <bool DW>
__global__ __launch_bounds__(VW * 64) void vox_points_bwd_kernel(
    int64_t P, const float* __restrict__ xyz, const int* __restrict__ vox,
    const float* __restrict__ centres, const int* __restrict__ vertex_idx,
    const float* __restrict__ emb, float voxel_size,
    const float* __restrict__ pk, const float* __restrict__ rgb,
    const uint32_t* __restrict__ masks, const float* __restrict__ g_sdf,
    const float* __restrict__ g_rgb, float* __restrict__ g_xyz,
    float* __restrict__ g_emb, float* __restrict__ gc3,
    float* __restrict__ ghc, float* __restrict__ gf, float* __restrict__ gh2,
    float* __restrict__ gh1, const int* __restrict__ n_dev) {
  using K = VoxPack;
  if (!DW) gc3 = ghc = gf = gh2 = gh1 = nullptr;
  if (n_dev != nullptr) P = *n_dev < P ? (*n_dev > 0 ? *n_dev : 0) : P;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wa = reinterpret_cast<float*>(smem_raw);
  float* wb = wa + kBufA;
  const uint32_t la = lds_addr(wa), lb = lds_addr(wb);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  const int64_t ntiles = (P + 15) / 16;
  const int64_t t0 = ntiles * blockIdx.x / gridDim.x;
  const int64_t t1 = ntiles * (blockIdx.x + 1) / gridDim.x;
  if (t0 >= t1) return;   // uniform per block
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  lds_dma_issue<VW, K::RC_LEN>(pk + K::RC, la, wave, lane);
  lds_dma_issue<VW, K::RS_LEN>(pk + K::RS, lb, wave, lane);
  for (int64_t tb = t0; tb < t1; tb += VW) {
    const bool active = tb + wave < t1;    // wave-uniform
    const bool more = tb + VW < t1;
    const int64_t pt = (tb + wave) * 16 + li;
    const bool valid = active && pt < P;
    uint32_t m1 = 0, m2 = 0, mc = 0;
    float gs = 0.f, g3[3] = {0.f, 0.f, 0.f};
    if (valid) {
      m1 = masks[(pt * 3 + 0) * 4 + q];
      m2 = masks[(pt * 3 + 1) * 4 + q];
      mc = masks[(pt * 3 + 2) * 4 + q];
      gs = g_sdf ? g_sdf[pt] : 0.f;
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        const float c = rgb[pt * 3 + o];
        g3[o] = g_rgb ? g_rgb[pt * 3 + o] * (c * (1.f - c)) : 0.f;
      }
      if (gc3 && q == 0) {
        gc3[pt * 4 + 0] = g3[0];
        gc3[pt * 4 + 1] = g3[1];
        gc3[pt * 4 + 2] = g3[2];
        gc3[pt * 4 + 3] = gs;
      }
    }
    f32x4 g[8], a[9];
    f32x4 gx_c = z4;
    lds_dma_landed();
    __syncthreads();        // A = colour head, B = sdf_out
    if (active) {
      // ---- colour head: ghc = mask(WO^T g3); [gf, gx] = WC^T ghc ------------
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) {
        f32x4 t = z4;
#pragma unroll
        for (int o = 0; o < 3; ++o)
          t += *reinterpret_cast<const f32x4*>(wa + (K::WOB - K::RC) +
                                               o * 128 + 16 * jt + 4 * q) *
               g3[o];
        g[jt] = mask4(t, mc, 4 * jt);
      }
      if (valid && ghc) save128(ghc, pt, q, g);
#pragma unroll
      for (int kt = 0; kt < 9; ++kt) a[kt] = z4;
      dense<9, 32, 32>(wa + (K::WCT - K::RC), lane, 0, g, a);
      gx_c = a[8];  // colour head's share of d loss / d x
      if (valid && gf) save128(gf, pt, q, a);
    }
    __syncthreads();        // A free
    lds_dma_issue<VW, K::R1_LEN>(pk + K::R1, la, wave, lane);
    if (active) {
      // ---- sdf_out: gh2 = mask(WS[1:]^T gf + WS[0] g_sdf) --------------------
#pragma unroll
      for (int jt = 0; jt < 8; ++jt)
        g[jt] = *reinterpret_cast<const f32x4*>(wb + (K::WS0B - K::RS) +
                                                16 * jt + 4 * q) * gs;
      dense<8, 32, 32>(wb + (K::WST - K::RS), lane, 0, a, g);
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) g[jt] = mask4(g[jt], m2, 4 * jt);
      if (valid && gh2) save128(gh2, pt, q, g);
    }
    lds_dma_landed();
    __syncthreads();        // A = layer 1, B free
    lds_dma_issue<VW, K::R0_LEN>(pk + K::R0, lb, wave, lane);
    if (active) {
      // ---- layer 1: gh1 = mask(W1^T gh2) ------------------------------------
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) a[kt] = z4;
      dense<8, 32, 32>(wa + (K::W1T - K::R1), lane, 0, g, a);
#pragma unroll
      for (int jt = 0; jt < 8; ++jt) a[jt] = mask4(a[jt], m1, 4 * jt);
      if (valid && gh1) save128(gh1, pt, q, a);
    }
    lds_dma_landed();
    __syncthreads();        // A free (the scatter's scratch), B = layer 0
    if (active) {
      // ---- layer 0: gx = W0^T gh1 (+ the colour head's share) ----------------
      f32x4 gx[1] = {gx_c};
      dense<1, 32, 32>(wb + (K::W0T - K::R0), lane, 0, a, gx);
      // ---- trilinear backward ------------------------------------------------
      Corner C;
      corners(xyz, vox, centres, vertex_idx, voxel_size, pt, valid, C);
      float gp[3] = {0.f, 0.f, 0.f};
      if (g_xyz) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (C.row[c] < 0) continue;
          const f32x4 e = *reinterpret_cast<const f32x4*>(
              emb + (int64_t)C.row[c] * 16 + 4 * q);
          const float dot = e[0] * gx[0][0] + e[1] * gx[0][1] +
                            e[2] * gx[0][2] + e[3] * gx[0][3];
          const float wx = (c & 4) ? C.p[0] : 1.f - C.p[0];
          const float wy = (c & 2) ? C.p[1] : 1.f - C.p[1];
          const float wz = (c & 1) ? C.p[2] : 1.f - C.p[2];
          gp[0] += ((c & 4) ? dot : -dot) * wy * wz;
          gp[1] += ((c & 2) ? dot : -dot) * wx * wz;
          gp[2] += ((c & 1) ? dot : -dot) * wx * wy;
        }
      }
      if (g_emb) {
        // Embedding gradient.  Consecutive points are consecutive samples of
        // a ray and share their voxel, and a scene has only a few thousand
        // vertices: one atomic per (point, corner, feature) — 128 a point —
        // serialises on the same addresses (measured 0.8 ms for 48 000
        // points).  The tile is transposed through LDS (this wave's scratch
        // in buffer A, free since the last barrier); lane group k walks the
        // 16 points for corners 2k, 2k+1, merges runs that hit the same
        // embedding row in a register and issues one coalesced 64-byte atomic
        // per run.
        float* gt = wa + wave * kVoxScatter;
        int* rw = reinterpret_cast<int*>(gt + 16 * 17);
        float* ww = gt + 16 * 17 + 16 * 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) gt[li * 17 + 4 * q + r] = gx[0][r];
        if (q == 0) {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            rw[li * 8 + c] = C.row[c];
            ww[li * 8 + c] = C.w[c];
          }
        }
        wave_lds_sync();
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int c = 2 * q + cc;
          int cur = -1;
          float acc = 0.f;
#pragma unroll 4
          for (int j = 0; j < 16; ++j) {
            const int row = rw[j * 8 + c];
            if (row != cur) {
              if (cur >= 0 && acc != 0.f)
                atomicAdd(g_emb + (int64_t)cur * 16 + li, acc);
              cur = row;
              acc = 0.f;
            }
            acc = fmaf(ww[j * 8 + c], gt[j * 17 + li], acc);
          }
          if (cur >= 0 && acc != 0.f)
            atomicAdd(g_emb + (int64_t)cur * 16 + li, acc);
        }
      }
      if (g_xyz) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const float s = group4_sum(gp[ax]) / voxel_size;
          if (valid && q == 0) g_xyz[pt * 3 + ax] = s;
        }
      }
    }
    if (more) {
      __syncthreads();      // A (scratch) and B free
      lds_dma_issue<VW, K::RC_LEN>(pk + K::RC, la, wave, lane);
      lds_dma_issue<VW, K::RS_LEN>(pk + K::RS, lb, wave, lane);
    }
  }
}

// host: packed <- flat index table
void build_vox_index(int32_t* idx) {
  using F = VoxFlat;
  using K = VoxPack;
  for (int i = 0; i < K::LEN; ++i) idx[i] = -1;
  for (int jt = 0; jt < 8; ++jt)
    for (int l = 0; l < 64; ++l) {
      const int m = l & 15, q = l >> 4, row = 16 * jt + m;
      for (int s = 0; s < 4; ++s)
        idx[K::W0 + (jt * 4 + s) * 64 + l] = F::W0 + row * 16 + kmap(s, q);
      for (int s = 0; s < 32; ++s) {
        idx[K::W1 + (jt * 32 + s) * 64 + l] = F::W1 + row * 128 + kmap(s, q);
        idx[K::WS + (jt * 32 + s) * 64 + l] =
            F::WS + (1 + row) * 128 + kmap(s, q);
        // transposed: rows = input feature 16kt + m, K-slot = output kmap(s,q)
        idx[K::WST + (jt * 32 + s) * 64 + l] =
            F::WS + (1 + kmap(s, q)) * 128 + row;
        idx[K::W1T + (jt * 32 + s) * 64 + l] = F::W1 + kmap(s, q) * 128 + row;
      }
      for (int s = 0; s < 36; ++s)
        idx[K::WC + (jt * 36 + s) * 64 + l] =
            F::WC + row * 144 + vox_color_in(s, q);
    }
  for (int kt = 0; kt < 9; ++kt)
    for (int s = 0; s < 32; ++s)
      for (int l = 0; l < 64; ++l) {
        const int m = l & 15, q = l >> 4;
        // colour layer input 16kt + m (kt = 8: the 16 voxel features)
        idx[K::WCT + (kt * 32 + s) * 64 + l] =
            F::WC + kmap(s, q) * 144 + 16 * kt + m;
      }
  for (int s = 0; s < 32; ++s)
    for (int l = 0; l < 64; ++l)
      idx[K::W0T + s * 64 + l] = F::W0 + kmap(s, l >> 4) * 16 + (l & 15);
  for (int j = 0; j < 128; ++j) {
    idx[K::B0 + j] = F::B0 + j;
    idx[K::B1 + j] = F::B1 + j;
    idx[K::BS + j] = F::BS + 1 + j;
    idx[K::WS0 + j] = F::WS + j;
    idx[K::WS0B + j] = F::WS + j;
    idx[K::BC + j] = F::BC + j;
  }
  idx[K::BS0] = F::BS;
  for (int o = 0; o < 3; ++o) {
    for (int j = 0; j < 128; ++j) {
      idx[K::WO + o * 128 + j] = F::WO + o * 128 + j;
      idx[K::WOB + o * 128 + j] = F::WO + o * 128 + j;
    }
    idx[K::BO + o] = F::BO + o;
  }
}

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int xrd_vox_flat_len(void) { return VoxFlat::LEN; }
int xrd_vox_pack_len(void) { return VoxPack::LEN; }

int xrd_vox_pack_index(int32_t* idx) {
  if (idx == nullptr) return XRD_ERR_ARG;
  build_vox_index(idx);
  return XRD_OK;
}

static size_t vox_lds_bytes() {
  return (size_t)(kBufA + kBufB) * sizeof(float);
}

static int vox_setup(const void* kern) {
  const int lds = (int)vox_lds_bytes();
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                          lds) != hipSuccess)
    return check_launch("hipFuncSetAttribute");
  return XRD_OK;
}

int xrd_vox_points_fwd(int64_t n_points, const float* xyz,
                       const int32_t* voxel_idx, const float* centres,
                       const int32_t* vertex_idx, const float* embeddings,
                       float voxel_size, const float* packed, float* sdf,
                       float* rgb, float* save_x, float* save_h1,
                       float* save_h2, float* save_f, float* save_hc,
                       uint32_t* masks, const int32_t* n_points_dev,
                       xrd_stream_t stream) {
  if (n_points < 0 || voxel_size <= 0.f) return XRD_ERR_ARG;
  if (n_points == 0) return XRD_OK;
  if (!xyz || !voxel_idx || !centres || !vertex_idx || !embeddings ||
      !packed || !sdf || !rgb)
    return XRD_ERR_ARG;
  static bool ready = false;
  if (!ready) {
    int rc = vox_setup(reinterpret_cast<const void*>(vox_points_fwd_kernel));
    if (rc != XRD_OK) return rc;
    ready = true;
  }
  const int64_t tiles = (n_points + 15) / 16;
  const int nb = (int)(tiles < kVoxBlocks ? tiles : kVoxBlocks);
  hipLaunchKernelGGL(vox_points_fwd_kernel, dim3(nb), dim3(VW * 64),
                     vox_lds_bytes(), (hipStream_t)stream,
                     n_points, xyz, voxel_idx, centres, vertex_idx, embeddings,
                     voxel_size, packed, sdf, rgb, save_x, save_h1, save_h2,
                     save_f, save_hc, masks, n_points_dev);
  return check_launch("xrd_vox_points_fwd");
}

int xrd_vox_points_bwd(int64_t n_points, const float* xyz,
                       const int32_t* voxel_idx, const float* centres,
                       const int32_t* vertex_idx, const float* embeddings,
                       float voxel_size, const float* packed, const float* rgb,
                       const uint32_t* masks, const float* g_sdf,
                       const float* g_rgb, float* g_xyz, float* g_embeddings,
                       float* g_c3, float* g_hc, float* g_f, float* g_h2,
                       float* g_h1, const int32_t* n_points_dev,
                       xrd_stream_t stream) {
  if (n_points < 0 || voxel_size <= 0.f) return XRD_ERR_ARG;
  if (n_points == 0) return XRD_OK;
  if (!xyz || !voxel_idx || !centres || !vertex_idx || !embeddings ||
      !packed || !rgb || !masks)
    return XRD_ERR_ARG;
  static bool ready = false;
  if (!ready) {
    int rc = vox_setup(
        reinterpret_cast<const void*>(vox_points_bwd_kernel<false>));
    if (rc == XRD_OK)
      rc = vox_setup(
          reinterpret_cast<const void*>(vox_points_bwd_kernel<true>));
    if (rc != XRD_OK) return rc;
    ready = true;
  }
  const int64_t tiles = (n_points + 15) / 16;
  const int nb = (int)(tiles < kVoxBlocks ? tiles : kVoxBlocks);
  const bool dw = g_c3 || g_hc || g_f || g_h2 || g_h1;
  auto kern = dw ? vox_points_bwd_kernel<true> : vox_points_bwd_kernel<false>;
  hipLaunchKernelGGL(kern, dim3(nb), dim3(VW * 64), vox_lds_bytes(),
                     (hipStream_t)stream, n_points, xyz, voxel_idx, centres,
                     vertex_idx, embeddings, voxel_size, packed, rgb, masks,
                     g_sdf, g_rgb, g_xyz, g_embeddings, g_c3, g_hc, g_f, g_h2,
                     g_h1, n_points_dev);
  return check_launch("xrd_vox_points_bwd");
}

}  // extern "C"
