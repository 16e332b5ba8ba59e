// NICE-SLAM fused render for gfx950 (MI355X): one wave = one 16-sample tile of a ray.
//
//   sample z (f64, depth-guided 32 uniform + 16 near-surface, rank-sorted)
//   -> trilinear lookup in the channel-last feature grids
//   -> three tiny MLP decoders evaluated as an in-register chain of
//      v_mfma_f32_16x16x4_f32 (exact f32): rows = output features, columns =
//      16 points of a tile; the accumulator of layer i is directly the B
//      operand of layer i+1 (nice_layout.h)
//   -> occupancy compositing with wave shuffles (exclusive product scan).
//
// Reference behaviour restated (never copied): slam/models/conv_onet.py:339-524,
// slam/model_components/decoder_nice.py:195-234,297-320,386-414,
// slam/model_components/utils.py:189-244, torch grid_sample (bilinear, border,
// align_corners=True) semantics.  Parity oracle: oracle/nice_oracle.py.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <vector>

#include "common.h"
#include "nice_layout.h"
#include "nice_device.h"
#include "point_common.h"

namespace xrd {
namespace {

constexpr int PTS = 17;

// ---------------------------------------------------------------------------
// host: packed <- flat index tables
// ---------------------------------------------------------------------------
template <int CD, int OD>
void build_mlp_index(int32_t* idx) {
  using F = MlpFlat<CD, OD>;
  using P = MlpPack<CD, OD>;
  for (int i = 0; i < P::LEN; ++i) idx[i] = -1;
  for (int k = 0; k < kEmbK; ++k)
    for (int a = 0; a < 3; ++a) idx[P::EMB + k * 4 + a] = F::EB + a * kEmbK + k;
  for (int jt = 0; jt < 2; ++jt)
    for (int s = 0; s < kEmbS; ++s)
      for (int l = 0; l < 64; ++l) {
        int j = 16 * jt + (l & 15), k = emap(s, l >> 4);
        if (k < kEmbK) {
          idx[P::W0 + (jt * kEmbS + s) * 64 + l] = F::P0W + j * kEmbK + k;
          idx[P::W3E + (jt * kEmbS + s) * 64 + l] =
              F::P3W + j * (kEmbK + 32) + k;
        }
      }
  for (int i = 1; i <= 4; ++i)
    for (int jt = 0; jt < 2; ++jt)
      for (int s = 0; s < 8; ++s)
        for (int l = 0; l < 64; ++l) {
          int j = 16 * jt + (l & 15), k = kmap(s, l >> 4);
          idx[P::wh(i) + (jt * 8 + s) * 64 + l] =
              F::pw(i) + j * F::pstride(i) + F::pcol(i) + k;
          // transposed: rows = input feature k, k-slots = output feature j
          int kk = 16 * jt + (l & 15), jj = kmap(s, l >> 4);
          idx[P::wht(i) + (jt * 8 + s) * 64 + l] =
              F::pw(i) + jj * F::pstride(i) + F::pcol(i) + kk;
        }
  for (int i = 0; i < 5; ++i) {
    for (int jt = 0; jt < 2; ++jt)
      for (int s = 0; s < P::KC; ++s)
        for (int l = 0; l < 64; ++l) {
          int j = 16 * jt + (l & 15), k = kmap(s, l >> 4);
          idx[P::wc(i) + (jt * P::KC + s) * 64 + l] = F::fcw(i) + j * CD + k;
        }
    for (int kt = 0; kt < P::KTC; ++kt)
      for (int s = 0; s < 8; ++s)
        for (int l = 0; l < 64; ++l) {
          int k = 16 * kt + (l & 15), j = kmap(s, l >> 4);
          idx[P::wct(i) + (kt * 8 + s) * 64 + l] = F::fcw(i) + j * CD + k;
        }
    for (int j = 0; j < 32; ++j) {
      idx[P::B + i * 32 + j] = F::pb(i) + j;
      idx[P::BC + i * 32 + j] = F::fcb(i) + j;
    }
  }
  for (int o = 0; o < OD; ++o) {
    for (int j = 0; j < 32; ++j) idx[P::WOUT + o * 32 + j] = F::OW + o * 32 + j;
    idx[P::BOUT + o] = F::OB + o;
  }
  for (int kt = 0; kt < 6; ++kt)
    for (int s = 0; s < 8; ++s)
      for (int l = 0; l < 64; ++l) {
        int k = emapT(kt, l & 15), j = kmap(s, l >> 4);
        if (k < kEmbK) {
          idx[P::W0T + (kt * 8 + s) * 64 + l] = F::P0W + j * kEmbK + k;
          idx[P::W3ET + (kt * 8 + s) * 64 + l] = F::P3W + j * (kEmbK + 32) + k;
        }
      }
}

void build_noxyz_index(int32_t* idx) {
  using F = NoXyzFlat;
  using P = NoXyzPack;
  for (int i = 0; i < P::LEN; ++i) idx[i] = -1;
  for (int i = 0; i < 5; ++i) {
    const int ks = P::ks(i);
    for (int jt = 0; jt < 2; ++jt)
      for (int s = 0; s < ks; ++s)
        for (int l = 0; l < 64; ++l) {
          int j = 16 * jt + (l & 15);
          int k = (s >= 8 ? 32 : 0) + kmap(s & 7, l >> 4);
          idx[P::w(i) + (jt * ks + s) * 64 + l] = F::pw(i) + j * F::pstride(i) + k;
        }
    const int kts = (i == 3 ? 4 : 2);
    for (int kt = 0; kt < kts; ++kt)
      for (int s = 0; s < 8; ++s)
        for (int l = 0; l < 64; ++l) {
          int k = 16 * kt + (l & 15), j = kmap(s, l >> 4);
          idx[P::wt(i) + (kt * 8 + s) * 64 + l] = F::pw(i) + j * F::pstride(i) + k;
        }
    for (int j = 0; j < 32; ++j) idx[P::B + i * 32 + j] = F::pb(i) + j;
  }
  for (int j = 0; j < 32; ++j) idx[P::WOUT + j] = F::OW + j;
  idx[P::BOUT] = F::OB;
}


template <int STAGE, int NT, bool NEED_DP, bool NEED_DW, int FWV>
__global__ __launch_bounds__(FWV * 64, (FWV + 3) / 4) void
nice_bwd_fused_kernel(
    xrd_nice_scene sc, int n, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ gt_depth,
    const float* __restrict__ dmax_p, const float* __restrict__ raw,
    const double* __restrict__ g_depth, const double* __restrict__ g_var,
    const float* __restrict__ g_rgb, float* gg_middle, float* gg_fine,
    float* gg_color, double* __restrict__ part, float* __restrict__ dw_rep) {
  static_assert(!NEED_DW || STAGE == XRD_STAGE_COLOR, "dW: colour stage");
  constexpr int S = NT * 16;
  static_assert(!NEED_DW || FWV == FWD, "exchange roles assume 8 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);  // staged fragments
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  // behind the largest staged pass in every variant (round 2 put the DW
  // variant's scratch at its exchange region, which the fine decoder's staged
  // fragments overlap: a wave's scatter tile could overwrite fragments other
  // waves were still reading)
  float* scratch = wl + kWMax + wave * kScratch;
  double* zbuf = reinterpret_cast<double*>(scratch);
  ScatterLds SL;
  SL.gt = scratch + 256;
  SL.off = reinterpret_cast<int*>(SL.gt + 16 * 33);
  SL.w = SL.gt + 16 * 33 + 16 * 8;
  DwAcc A;
  if (NEED_DW) dw_acc_zero(A);
  const bool use_depth = gt_depth != nullptr;
  const int ntiles = n * NT;
  const int ngroups = (ntiles + FWV - 1) / FWV;
  using PM = MlpPack<32, 1>;
  using PF = MlpPack<64, 1>;
  using PC = MlpPack<32, 4>;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int tile_id = __builtin_amdgcn_readfirstlane(grp * FWV + wave);
    const bool active = tile_id < ntiles;
    const int ray = active ? tile_id / NT : 0;
    const int tile = active ? tile_id % NT : 0;
    TileGeom tg = {};
    float gocc = 0.f, gcol[3] = {0.f, 0.f, 0.f};
    double gp64[3] = {0.0, 0.0, 0.0};
    float gp32[1][3] = {{0.f, 0.f, 0.f}};
    float p32[1][3] = {{0.f, 0.f, 0.f}};
    f32x4 c_m[1][2];
    uint64_t mask[1] = {0};
    Tri tr;
    if (active) {
      RayCtx rc;
      load_ray(rays_o, rays_d, gt_depth, ray, use_depth, rc);
      const float dmax = use_depth ? dmax_p[0] : 0.f;
      const double zl = sample_z<S>(sc, rc, dmax, lane, zbuf, zbuf + 64);
      float gocc_s, w, grgb[3];
      composite_bwd<S>(raw, ray, lane, zl, g_depth, g_var, g_rgb, gocc_s, w,
                       grgb);
      const int src = 16 * tile + li;
      tile_geom(rc, zbuf[64 + src], sc.bound, tg);
      gocc = __shfl(gocc_s, src);
      if (!tg.inb) gocc = 0.f;  // occupancy was overridden to 100
      const float wsrc = __shfl(w, src);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        gcol[a] = grgb[a] * wsrc;
        p32[0][a] = tg.p32[a];
      }
      tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 3, tr);
      tri_gather(sc.grid[1], tr, q, c_m[0]);
    }
    // ---- middle decoder (every stage) ---------------------------------------
    {
      const float go[1][1] = {{gocc}};
      f32x4 gc[1][2];
      stage_weights(wl, sc.dec[1], PM::WHT);
      if (active) {
        float om[1][1];
        mlp_fwd<1, 32, 1, true, false>(wl, lane, p32, c_m, om, mask, nullptr);
      }
      stage_weights(wl, sc.dec[1] + PM::EMB,
                    (NEED_DP ? PM::LEN : PM::W0T) - PM::EMB);
      if (active) {
        mlp_bwd<1, 32, 1, NEED_DP, NEED_DP>(wl - PM::EMB, lane, p32, c_m, go,
                                            mask, gc, gp32);
        tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 3, tr);
        if (NEED_DP) tri_backward_dp(sc.grid[1], tr, q, gc[0], gp64);
        grid_scatter(gg_middle, sc.gmask[1], tr, lane, gc[0], SL);
      }
    }
    // ---- fine decoder ---------------------------------------------------------
    if (STAGE >= XRD_STAGE_FINE) {
      f32x4 c_f[1][4], gc[1][4];
      const float go[1][1] = {{gocc}};
      if (active) {
        f32x4 cf[2];
        tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 6, tr);
        tri_gather(sc.grid[2], tr, q, cf);
        c_f[0][0] = cf[0];
        c_f[0][1] = cf[1];
        c_f[0][2] = c_m[0][0];
        c_f[0][3] = c_m[0][1];
      }
      stage_weights(wl, sc.dec[2], PF::WHT);
      if (active) {
        float of[1][1];
        mlp_fwd<1, 64, 1, true, false>(wl, lane, p32, c_f, of, mask, nullptr);
      }
      stage_weights(wl, sc.dec[2] + PF::EMB,
                    (NEED_DP ? PF::LEN : PF::W0T) - PF::EMB);
      if (active) {
        mlp_bwd<1, 64, 1, NEED_DP, NEED_DP>(wl - PF::EMB, lane, p32, c_f, go,
                                            mask, gc, gp32);
        const f32x4 g2[2] = {gc[0][0], gc[0][1]};  // c_middle is no_grad
        tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 6, tr);
        if (NEED_DP) tri_backward_dp(sc.grid[2], tr, q, g2, gp64);
        grid_scatter(gg_fine, sc.gmask[2], tr, lane, g2, SL);
      }
    }
    // ---- colour decoder -------------------------------------------------------
    if (STAGE == XRD_STAGE_COLOR) {
      f32x4 c_c[1][2], gc[1][2];
      // channel 3 is overwritten by fine+middle occupancy -> no gradient
      const float go[1][4] = {{gcol[0], gcol[1], gcol[2], 0.f}};
      if (active) {
        tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 9, tr);
        tri_gather(sc.grid[3], tr, q, c_c[0]);
      }
      f32x4 hs[5][2];
      stage_weights(wl, sc.dec[3], PC::WHT);
      if (active) {
        float oc[1][4];
        mlp_fwd<1, 32, 4, true, NEED_DW>(wl, lane, p32, c_c, oc, mask, hs);
      }
      stage_weights(wl, sc.dec[3] + PC::EMB,
                    ((NEED_DP || NEED_DW) ? PC::LEN : PC::W0T) - PC::EMB);
      if (NEED_DW) {
        const int nact = ntiles - grp * FWV < FWV ? ntiles - grp * FWV : FWV;
        color_bwd_dw<NEED_DP>(wl - PC::EMB, wl + kDwBase, wave, lane, active,
                              nact, p32, c_c, go, mask[0], hs, gc, gp32, A);
      } else if (active) {
        mlp_bwd<1, 32, 4, NEED_DP, NEED_DP>(wl - PC::EMB, lane, p32, c_c, go,
                                            mask, gc, gp32);
      }
      if (active) {
        tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 9, tr);
        if (NEED_DP) tri_backward_dp(sc.grid[3], tr, q, gc[0], gp64);
        grid_scatter(gg_color, sc.gmask[3], tr, lane, gc[0], SL);
      }
    }
    if (NEED_DP && active) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double g = gp64[a] + (double)gp32[0][a];
        const double so = wave_sum(g), sd = wave_sum(g * tg.z);
        if (lane == 0) {
          part[(size_t)tile_id * 6 + a] = so;
          part[(size_t)tile_id * 6 + 3 + a] = sd;
        }
      }
    }
  }
  if (NEED_DW)
    dw_flush(dw_rep + (size_t)(blockIdx.x % kDwRep) * kColorFlat, wave, lane,
             A);
}

// ---- tracking batches: the three decoders of a tile in three blocks ---------
// A tracking batch (200 rays) is 600 tiles; the fused backward above gives a
// tile wave a chain of six decoder passes (three recomputed forwards, three
// backwards), each behind a staging barrier, on 150 of the 256 CUs.  Here
// grid.y = the decoder (0 middle, 1 fine, 2 colour): a block stages ONE
// decoder's fragments and runs ONE pass over W tiles (tiles numbered
// ray * NT + tile across the batch, whatever ray they belong to); the three
// roles' d loss / d point meet behind the kernel boundary in the finishing
// launch (9 partial rows a ray).  Roles inside ONE block would need the three
// decoders' fragments at once (197 KB forward); with the fragments read from
// L2 instead the forward measured 66 us, the backward 149 us.
// The decoder pass of a wave is MFMA-issue bound (240 / 336 / 240 16x16x4
// steps forward), so what a launch costs is the number of waves that share a
// SIMD: 12-wave blocks (rounds 3-4: four rays a block) put three on each SIMD
// of 150 CUs and leave 106 CUs idle — in that shape a role-split forward
// bought nothing over the three-pass kernel (40.8 vs 39.6 us, round 3);
// 8-wave blocks put two on each SIMD of 225 CUs.  Measured at 200 rays (HIP
// events over 20 captured calls behind a 10 ms matmul,
// tools/nice_track_timing.py): forward 35.7 -> 29.3 us incl. its finishing launch (the
// three-pass kernel: 39.3), backward from the masks 43.4 -> 33.1 us; 4- and
// 6-wave blocks 36.7 / 39.6 and 43.3 / 47.2 us (more blocks than CUs, every
// block stages its decoder); with the first staging's loads issued before
// the ray set-up and the gathers (stage_issue / stage_commit) 28.7 / 30.5 us.
// Batches whose 8-wave blocks would not fit one round of the 256 CUs take 12.
constexpr int role_waves(int n_rays) {
  return (n_rays * 3 + 7) / 8 * 3 <= 256 ? 8 : 12;
}
// the 9 part rows of a ray must fit the workspace xrd_nice_bwd_ws_floats(n)
// promises (n*36 + replicas + 64)
constexpr int kRoleMaxRays = 340;
static_assert((size_t)kRoleMaxRays * 9 * 6 * 2 <=
                  (size_t)kRoleMaxRays * 36 + (size_t)kDwRep * kColorFlat,
              "part rows fit the workspace");
template <int ROLE> struct RolePack;
template <> struct RolePack<0> { using P = MlpPack<32, 1>; };
template <> struct RolePack<1> { using P = MlpPack<64, 1>; };
template <> struct RolePack<2> { using P = MlpPack<32, 4>; };
constexpr int kRoleFwdMax = MlpPack<64, 1>::WHT;
constexpr int kRoleBwdMax = MlpPack<64, 1>::LEN - MlpPack<64, 1>::EMB;
constexpr int kRoleWl = kRoleFwdMax > kRoleBwdMax ? kRoleFwdMax : kRoleBwdMax;
template <int W>
constexpr size_t role_lds_floats() {
  return (size_t)kRoleWl + W * 256;
}

// backward to the rays (no grid / decoder gradients): part row
// (ray * NT + tile) * 3 + role
template <int NT, int W>
__global__ __launch_bounds__(W * 64, 1) void nice_bwd_roles_kernel(
    xrd_nice_scene sc, int n, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ gt_depth,
    const float* __restrict__ dmax_p, const float* __restrict__ raw,
    const double* __restrict__ g_depth, const double* __restrict__ g_var,
    const float* __restrict__ g_rgb, double* __restrict__ part,
    const uint64_t* __restrict__ masks) {
  // masks != nullptr: the forward kept the ReLU masks (nice_fwd_roles_kernel)
  // — no forward fragments are staged, no forward pass is run
  constexpr int S = NT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  const int role = blockIdx.y;
  const bool have = masks != nullptr;
  double* zbuf = reinterpret_cast<double*>(wl + kRoleWl) + wave * 128;
  using PM = MlpPack<32, 1>;
  using PF = MlpPack<64, 1>;
  using PC = MlpPack<32, 4>;
  // the fragments of the block's first pass are on their way to LDS under the
  // ray set-up, the compositing backward and the gathers of the first group
  constexpr int T = W * 64;
  constexpr int NV = (kRoleWl / 4 + T - 1) / T;
  const float* wfirst = sc.dec[role + 1] +
                        (have ? (role == 1 ? PF::EMB : PM::EMB) : 0);
  const int wlen = have ? (role == 1 ? PF::LEN - PF::EMB : PM::LEN - PM::EMB)
                        : (role == 1 ? PF::WHT : PM::WHT);
  static_assert(PM::EMB == PC::EMB && PM::LEN == PC::LEN && PM::WHT == PC::WHT,
                "middle and colour packs share their layout");
  f32x4 wreg[NV];
  stage_issue<T, NV>(wfirst, wlen, wreg);
  bool first = true;
  auto stage_first = [&]() {
    if (first)
      stage_commit<T, NV>(wl, wlen, wreg);
    else
      stage_weights(wl, wfirst, wlen);
    first = false;
  };
  const int ntiles = n * NT;
  const int ngroups = (ntiles + W - 1) / W;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int t = __builtin_amdgcn_readfirstlane(grp * W + wave);
    const bool active = t < ntiles;
    const int ray = t / NT, tile = t - ray * NT;
    TileGeom tg = {};
    float gocc = 0.f, gcol[3] = {0.f, 0.f, 0.f};
    double gp64[3] = {0.0, 0.0, 0.0};
    float gp32[1][3] = {{0.f, 0.f, 0.f}};
    float p32[1][3] = {{0.f, 0.f, 0.f}};
    uint64_t mask[1] = {0};
    Tri tr;
    f32x4 c_a[1][4];
    if (active) {
      RayCtx rc;
      load_ray(rays_o, rays_d, gt_depth, ray, true, rc);
      const double zl = sample_z<S>(sc, rc, dmax_p[0], lane, zbuf, zbuf + 64);
      float gocc_s, w, grgb[3];
      composite_bwd<S>(raw, ray, lane, zl, g_depth, g_var, g_rgb, gocc_s, w,
                       grgb);
      const int src = 16 * tile + li;
      tile_geom(rc, zbuf[64 + src], sc.bound, tg);
      gocc = __shfl(gocc_s, src);
      if (!tg.inb) gocc = 0.f;  // occupancy was overridden to 100
      const float wsrc = __shfl(w, src);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        gcol[a] = grgb[a] * wsrc;
        p32[0][a] = tg.p32[a];
      }
      f32x4 c2[2];
      // (the two lookups of the fine decoder one after the other: issued
      // together — as the forward does — the backward measured 32.8 instead
      // of 30.5 us, its 240 registers leave no room for a second corner set)
      if (role == 1) {
        tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 3, tr);
        tri_gather(sc.grid[1], tr, q, c2);
        c_a[0][2] = c2[0];
        c_a[0][3] = c2[1];
      }
      const int g = role + 1;
      tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 3 * g, tr);
      tri_gather(sc.grid[g], tr, q, c2);
      c_a[0][0] = c2[0];
      c_a[0][1] = c2[1];
      if (have)
        mask[0] = masks[((size_t)(ray * NT + tile) * 3 + role) * 64 + lane];
    }
    if (role == 0) {
      const f32x4 c_m[1][2] = {{c_a[0][0], c_a[0][1]}};
      const float go[1][1] = {{gocc}};
      f32x4 gc[1][2];
      stage_first();
      if (!have) {
        if (active) {
          float om[1][1];
          mlp_fwd<1, 32, 1, true, false>(wl, lane, p32, c_m, om, mask,
                                         nullptr);
        }
        stage_weights(wl, sc.dec[1] + PM::EMB, PM::LEN - PM::EMB);
      }
      if (active) {
        mlp_bwd<1, 32, 1, true, true>(wl - PM::EMB, lane, p32, c_m, go, mask,
                                      gc, gp32);
        tri_backward_dp(sc.grid[1], tr, q, gc[0], gp64);
      }
    } else if (role == 1) {
      const float go[1][1] = {{gocc}};
      f32x4 gc[1][4];
      stage_first();
      if (!have) {
        if (active) {
          float of[1][1];
          mlp_fwd<1, 64, 1, true, false>(wl, lane, p32, c_a, of, mask,
                                         nullptr);
        }
        stage_weights(wl, sc.dec[2] + PF::EMB, PF::LEN - PF::EMB);
      }
      if (active) {
        mlp_bwd<1, 64, 1, true, true>(wl - PF::EMB, lane, p32, c_a, go, mask,
                                      gc, gp32);
        const f32x4 g2[2] = {gc[0][0], gc[0][1]};  // c_middle is no_grad
        tri_backward_dp(sc.grid[2], tr, q, g2, gp64);
      }
    } else {
      const f32x4 c_c[1][2] = {{c_a[0][0], c_a[0][1]}};
      // channel 3 is overwritten by fine+middle occupancy -> no gradient
      const float go[1][4] = {{gcol[0], gcol[1], gcol[2], 0.f}};
      f32x4 gc[1][2];
      stage_first();
      if (!have) {
        if (active) {
          float oc[1][4];
          mlp_fwd<1, 32, 4, true, false>(wl, lane, p32, c_c, oc, mask,
                                         nullptr);
        }
        stage_weights(wl, sc.dec[3] + PC::EMB, PC::LEN - PC::EMB);
      }
      if (active) {
        mlp_bwd<1, 32, 4, true, true>(wl - PC::EMB, lane, p32, c_c, go, mask,
                                      gc, gp32);
        tri_backward_dp(sc.grid[3], tr, q, gc[0], gp64);
      }
    }
    if (active) {
      const size_t row = ((size_t)ray * NT + tile) * 3 + role;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double g = gp64[a] + (double)gp32[0][a];
        const double so = wave_sum(g), sd = wave_sum(g * tg.z);
        if (lane == 0) {
          part[row * 6 + a] = so;
          part[row * 6 + 3 + a] = sd;
        }
      }
    }
  }
}

// Forward render.  A block = RPB rays (15 / 16 waves, one 16-sample tile
// each); like the backward it stages one decoder's forward fragments at a time
// in LDS and loops over groups of rays (persistent blocks, one per CU).
// (RPB = 5 / 8 rays = 15 / 16 waves for batches that still cover the chip,
// one ray per block below that.)
template <int NT>
__host__ __device__ constexpr int fwd_rays_wide() { return NT == 3 ? 5 : 8; }
template <int NT, int RPB>
constexpr size_t fwd_lds_floats() {
  return (size_t)kWMax + RPB * 256 + RPB * NT * 256;
}

template <int STAGE, int NT, int RPB>
__global__ __launch_bounds__(RPB * NT * 64, (RPB * NT + 3) / 4) void
nice_fwd_kernel(xrd_nice_scene sc, int n, const float* __restrict__ rays_o,
                const float* __restrict__ rays_d,
                const float* __restrict__ gt_depth,
                const float* __restrict__ dmax_p, double* __restrict__ depth,
                double* __restrict__ var, float* __restrict__ rgb,
                float* __restrict__ raw_out) {
  constexpr int S = NT * 16;
  constexpr int NW = RPB * NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);
  float* rawbuf = wl + kWMax;                      // [RPB][64][4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* zbuf = reinterpret_cast<double*>(rawbuf + RPB * 256) + wave * 128;
  const int q = lane >> 4, li = lane & 15;
  const int slot = wave / NT, tile = wave % NT;
  const bool use_depth = (gt_depth != nullptr) && STAGE != XRD_STAGE_COARSE;
  const int ngroups = (n + RPB - 1) / RPB;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int ray = __builtin_amdgcn_readfirstlane(grp * RPB + slot);
    const bool active = ray < n;
    double zl = 0.0;
    TileGeom tg = {};
    float p32[1][3] = {{0.f, 0.f, 0.f}};
    float occ = 0.f, col[3] = {0.f, 0.f, 0.f};
    uint64_t mdummy[1];
    Tri tr;
    f32x4 c_m[1][2];
    if (active) {
      RayCtx rc;
      load_ray(rays_o, rays_d, gt_depth, ray, use_depth, rc);
      const float dmax = use_depth ? dmax_p[0] : 0.f;
      zl = sample_z<S>(sc, rc, dmax, lane, zbuf, zbuf + 64);
      tile_geom(rc, zbuf[64 + 16 * tile + li], sc.bound, tg);
#pragma unroll
      for (int a = 0; a < 3; ++a) p32[0][a] = tg.p32[a];
      if (STAGE == XRD_STAGE_COARSE)
        tri_prepare(tg.p64, sc.bound, sc.coarse_enlarge, sc.gdim + 0, tr);
      else
        tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 3, tr);
      tri_gather(sc.grid[STAGE == XRD_STAGE_COARSE ? 0 : 1], tr, q, c_m[0]);
    }
    if (STAGE == XRD_STAGE_COARSE) {
      stage_weights(wl, sc.dec[0], NoXyzPack::WT);
      if (active) {
        float o1[1];
        noxyz_fwd<1, false>(wl, lane, c_m, o1, mdummy);
        occ = o1[0];
      }
    } else {
      stage_weights(wl, sc.dec[1], MlpPack<32, 1>::WHT);
      if (active) {
        float om[1][1];
        mlp_fwd<1, 32, 1, false, false>(wl, lane, p32, c_m, om, mdummy,
                                            nullptr);
        occ = om[0][0];
      }
      if (STAGE >= XRD_STAGE_FINE) {
        f32x4 c_f[1][4];
        if (active) {
          f32x4 cf[2];
          tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 6, tr);
          tri_gather(sc.grid[2], tr, q, cf);
          c_f[0][0] = cf[0];
          c_f[0][1] = cf[1];
          c_f[0][2] = c_m[0][0];
          c_f[0][3] = c_m[0][1];
        }
        stage_weights(wl, sc.dec[2], MlpPack<64, 1>::WHT);
        if (active) {
          float of[1][1];
          mlp_fwd<1, 64, 1, false, false>(wl, lane, p32, c_f, of, mdummy,
                                              nullptr);
          occ = of[0][0] + occ;  // NICE.forward: fine_occ + middle_occ
        }
      }
      if (STAGE == XRD_STAGE_COLOR) {
        f32x4 c_c[1][2];
        if (active) {
          tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 9, tr);
          tri_gather(sc.grid[3], tr, q, c_c[0]);
        }
        stage_weights(wl, sc.dec[3], MlpPack<32, 4>::WHT);
        if (active) {
          float oc[1][4];
          mlp_fwd<1, 32, 4, false, false>(wl, lane, p32, c_c, oc, mdummy,
                                              nullptr);
          col[0] = oc[0][0];
          col[1] = oc[0][1];
          col[2] = oc[0][2];
        }
      }
    }
    if (active) {
      if (!tg.inb) occ = 100.f;  // conv_onet.py:370
      if (q == 0)
        *reinterpret_cast<f32x4*>(rawbuf + (slot * 64 + 16 * tile + li) * 4) =
            f32x4{col[0], col[1], col[2], occ};
    }
    __syncthreads();
    // (rawbuf is rewritten only behind the next group's staging barriers)
    if (!active || tile != 0) continue;
    // compositing: lane l (< S) is sample l of the ray
    const bool valid = lane < S;
    f32x4 rw = {0.f, 0.f, 0.f, 0.f};
    if (valid)
      rw = *reinterpret_cast<const f32x4*>(rawbuf + (slot * 64 + lane) * 4);
    if (raw_out != nullptr && valid)
      *reinterpret_cast<f32x4*>(raw_out + ((size_t)ray * S + lane) * 4) = rw;
    const float alpha = valid ? 1.f / (1.f + expf(-10.f * rw[3])) : 0.f;
    const float f = valid ? (1.f - alpha + 1e-10f) : 1.f;
    float incl = f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float u = __shfl_up(incl, o);
      if (lane >= o) incl *= u;
    }
    float T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.f;
    const float w = alpha * T;
    const float R = wave_sum(w * rw[0]), G = wave_sum(w * rw[1]),
                B = wave_sum(w * rw[2]);
    const double dep = wave_sum(valid ? (double)w * zl : 0.0);
    const double tmp = zl - dep;
    const double vr = wave_sum(valid ? (double)w * tmp * tmp : 0.0);
    if (lane == 0) {
      depth[ray] = dep;
      var[ray] = vr;
      rgb[ray * 3 + 0] = R;
      rgb[ray * 3 + 1] = G;
      rgb[ray * 3 + 2] = B;
    }
  }
}

// Tracking-sized forward (colour stage, <= kRoleMaxRays rays, masks kept): the
// three decoders of a tile run on three BLOCKS at once (blockIdx.y = decoder,
// like nice_bwd_roles_kernel) instead of one after the other on one wave —
// at 200 rays x 3 tiles the plain forward is 600 waves each walking three
// staging barriers and three dependent decoder chains; here 1800 waves walk
// one.  The decoders of a sample meet in nice_fwd_roles_finish_kernel: the
// colour block writes raw[..][0:3], the fine block raw[..][3], the middle block
// ``occ_m`` [ray][64]; the finishing launch adds the two occupancies in the
// forward's order (fine + middle, conv_onet.py:370 override after it), writes
// the final raw row and composites the ray (same code, same values, bit for
// bit: tests/test_nice_hip.py::test_tracking_backward_from_the_forwards_masks).
template <int W>  // waves of a block: role_waves()
constexpr size_t role_fwd_lds_floats() {
  return (size_t)kRoleFwdMax + W * 256;
}

template <int NT, int W>
__global__ __launch_bounds__(W * 64, 1) void
nice_fwd_roles_kernel(xrd_nice_scene sc, int n,
                      const float* __restrict__ rays_o,
                      const float* __restrict__ rays_d,
                      const float* __restrict__ gt_depth,
                      const float* __restrict__ dmax_p,
                      float* __restrict__ raw_out,
                      uint64_t* __restrict__ masks,
                      float* __restrict__ occ_m) {
  constexpr int S = NT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  const int role = blockIdx.y;
  double* zbuf = reinterpret_cast<double*>(wl + kRoleFwdMax) + wave * 128;
  using PM = MlpPack<32, 1>;
  using PF = MlpPack<64, 1>;
  using PC = MlpPack<32, 4>;
  // the block's decoder: fragments on their way to LDS under the ray set-up
  // and the gathers of the first group
  constexpr int T = W * 64;
  constexpr int NV = (kRoleFwdMax / 4 + T - 1) / T;
  const float* wsrc = sc.dec[role + 1];
  const int wlen = role == 0 ? PM::WHT : role == 1 ? PF::WHT : PC::WHT;
  f32x4 wreg[NV];
  stage_issue<T, NV>(wsrc, wlen, wreg);
  bool first = true;
  const int ntiles = n * NT;
  const int ngroups = (ntiles + W - 1) / W;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int t = __builtin_amdgcn_readfirstlane(grp * W + wave);
    const bool active = t < ntiles;
    const int ray = t / NT, tile = t - ray * NT;
    float p32[1][3] = {{0.f, 0.f, 0.f}};
    uint64_t mask[1] = {0};
    f32x4 c_a[1][4];
    if (active) {
      RayCtx rc;
      TileGeom tg;
      Tri tr;
      load_ray(rays_o, rays_d, gt_depth, ray, true, rc);
      sample_z<S>(sc, rc, dmax_p[0], lane, zbuf, zbuf + 64);
      tile_geom(rc, zbuf[64 + 16 * tile + li], sc.bound, tg);
#pragma unroll
      for (int a = 0; a < 3; ++a) p32[0][a] = tg.p32[a];
      f32x4 c2[2];
      if (role == 1) {
        // both lookups of the fine decoder in one basic block: the 32 loads
        // go out together (28.7 -> 28.0 us)
        Tri trm;
        f32x4 cm[2];
        tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 3, trm);
        tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 6, tr);
        tri_gather(sc.grid[1], trm, q, cm);
        tri_gather(sc.grid[2], tr, q, c2);
        c_a[0][2] = cm[0];
        c_a[0][3] = cm[1];
      } else {
        const int g = role + 1;
        tri_prepare(tg.p64, sc.bound, 1.0, sc.gdim + 3 * g, tr);
        tri_gather(sc.grid[g], tr, q, c2);
      }
      c_a[0][0] = c2[0];
      c_a[0][1] = c2[1];
    }
    const size_t mrow = ((size_t)(ray * NT + tile) * 3 + role) * 64 + lane;
    const size_t srow = (size_t)ray * S + 16 * tile + li;
    if (first)
      stage_commit<T, NV>(wl, wlen, wreg);
    else
      stage_weights(wl, wsrc, wlen);
    first = false;
    if (role == 0) {
      if (active) {
        const f32x4 c_m[1][2] = {{c_a[0][0], c_a[0][1]}};
        float om[1][1];
        mlp_fwd<1, 32, 1, true, false>(wl, lane, p32, c_m, om, mask, nullptr);
        masks[mrow] = mask[0];
        if (q == 0) occ_m[(size_t)ray * 64 + 16 * tile + li] = om[0][0];
      }
    } else if (role == 1) {
      if (active) {
        float of[1][1];
        mlp_fwd<1, 64, 1, true, false>(wl, lane, p32, c_a, of, mask, nullptr);
        masks[mrow] = mask[0];
        if (q == 0) raw_out[srow * 4 + 3] = of[0][0];
      }
    } else {
      if (active) {
        const f32x4 c_c[1][2] = {{c_a[0][0], c_a[0][1]}};
        float oc[1][4];
        mlp_fwd<1, 32, 4, true, false>(wl, lane, p32, c_c, oc, mask, nullptr);
        masks[mrow] = mask[0];
        if (q == 0) {
          raw_out[srow * 4 + 0] = oc[0][0];
          raw_out[srow * 4 + 1] = oc[0][1];
          raw_out[srow * 4 + 2] = oc[0][2];
        }
      }
    }
  }
}

// one wave = one ray: occupancy = fine + middle (100 outside the bound),
// final raw row, compositing (the tail of nice_fwd_kernel)
template <int NT>
__global__ __launch_bounds__(256) void nice_fwd_roles_finish_kernel(
    xrd_nice_scene sc, int n, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ gt_depth,
    const float* __restrict__ dmax_p, const float* __restrict__ occ_m,
    float* __restrict__ raw_out, double* __restrict__ depth,
    double* __restrict__ var, float* __restrict__ rgb) {
  constexpr int S = NT * 16;
  __shared__ double zsh[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (ray >= n) return;
  RayCtx rc;
  TileGeom tg;
  load_ray(rays_o, rays_d, gt_depth, ray, true, rc);
  const double zl = sample_z<S>(sc, rc, dmax_p[0], lane, zsh[wave],
                                zsh[wave] + 64);
  tile_geom(rc, zl, sc.bound, tg);
  const bool valid = lane < S;
  f32x4 rw = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
    rw = *reinterpret_cast<const f32x4*>(raw_out + ((size_t)ray * S + lane) * 4);
    float occ = rw[3] + occ_m[(size_t)ray * 64 + lane];  // fine + middle
    if (!tg.inb) occ = 100.f;  // conv_onet.py:370
    rw[3] = occ;
    *reinterpret_cast<f32x4*>(raw_out + ((size_t)ray * S + lane) * 4) = rw;
  }
  const float alpha = valid ? 1.f / (1.f + expf(-10.f * rw[3])) : 0.f;
  const float f = valid ? (1.f - alpha + 1e-10f) : 1.f;
  float incl = f;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float u = __shfl_up(incl, o);
    if (lane >= o) incl *= u;
  }
  float T = __shfl_up(incl, 1);
  if (lane == 0) T = 1.f;
  const float w = alpha * T;
  const float R = wave_sum(w * rw[0]), G = wave_sum(w * rw[1]),
              B = wave_sum(w * rw[2]);
  const double dep = wave_sum(valid ? (double)w * zl : 0.0);
  const double tmp = zl - dep;
  const double vr = wave_sum(valid ? (double)w * tmp * tmp : 0.0);
  if (lane == 0) {
    depth[ray] = dep;
    var[ray] = vr;
    rgb[ray * 3 + 0] = R;
    rgb[ray * 3 + 1] = G;
    rgb[ray * 3 + 2] = B;
  }
}

// Point queries (the mesher's query_fn / color_func, conv_onet.py:213-240 ->
// NICE.forward with stage 'fine' / 'color', and ConvOnet.eval_points'
// out-of-bound override conv_onet.py:358-370): raw [n,4] = (rgb raw or 0,
// occupancy logit) of n free points.  One wave = 16 points, 8 waves a block,
// the decoders' forward fragments staged once per group of 128 points.
template <int STAGE>
__global__ __launch_bounds__(8 * 64, 2) void nice_points_kernel(
    xrd_nice_scene sc, int64_t n, const float* __restrict__ pts,
    float* __restrict__ raw_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  const int64_t ngroups = (n + 127) / 128;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t pt = (grp * 8 + wave) * 16 + li;
    const bool active = pt < n;
    double p64[3] = {0.0, 0.0, 0.0};
    float p32[1][3] = {{0.f, 0.f, 0.f}};
    if (active) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        p32[0][a] = pts[pt * 3 + a];
        p64[a] = (double)p32[0][a];
      }
    }
    const bool inb = in_bound(p64, sc.bound);
    float occ = 0.f, col[3] = {0.f, 0.f, 0.f};
    uint64_t mdummy[1];
    Tri tr;
    f32x4 c_m[1][2];
    tri_prepare(p64, sc.bound, 1.0, sc.gdim + 3, tr);
    tri_gather(sc.grid[1], tr, q, c_m[0]);
    stage_weights(wl, sc.dec[1], MlpPack<32, 1>::WHT);
    {
      float om[1][1];
      mlp_fwd<1, 32, 1, false, false>(wl, lane, p32, c_m, om, mdummy, nullptr);
      occ = om[0][0];
    }
    {
      f32x4 c_f[1][4], cf[2];
      tri_prepare(p64, sc.bound, 1.0, sc.gdim + 6, tr);
      tri_gather(sc.grid[2], tr, q, cf);
      c_f[0][0] = cf[0];
      c_f[0][1] = cf[1];
      c_f[0][2] = c_m[0][0];
      c_f[0][3] = c_m[0][1];
      stage_weights(wl, sc.dec[2], MlpPack<64, 1>::WHT);
      float of[1][1];
      mlp_fwd<1, 64, 1, false, false>(wl, lane, p32, c_f, of, mdummy, nullptr);
      occ = of[0][0] + occ;  // NICE.forward: fine_occ + middle_occ
    }
    if (STAGE == XRD_STAGE_COLOR) {
      f32x4 c_c[1][2];
      tri_prepare(p64, sc.bound, 1.0, sc.gdim + 9, tr);
      tri_gather(sc.grid[3], tr, q, c_c[0]);
      stage_weights(wl, sc.dec[3], MlpPack<32, 4>::WHT);
      float oc[1][4];
      mlp_fwd<1, 32, 4, false, false>(wl, lane, p32, c_c, oc, mdummy, nullptr);
      col[0] = oc[0][0];
      col[1] = oc[0][1];
      col[2] = oc[0][2];
    }
    if (!inb) occ = 100.f;  // conv_onet.py:370
    if (active && q == 0)
      *reinterpret_cast<f32x4*>(raw_out + pt * 4) =
          f32x4{col[0], col[1], col[2], occ};
  }
}

// ---------------------------------------------------------------------------
// Point-SLAM geometry path: neighbour interpolation + the geometry decoder.
//   MLP_geometry.get_feature_at_pos / forward
//   (slam/model_components/decoder_pointslam.py:162-273): the <= 8 nearest
//   neural points of a sample (ids from the kNN), inverse squared-distance
//   weights zeroed beyond the query radius, L1-normalised; c = sum w f_geo;
//   samples with fewer than min_nn neighbours inside the radius get the
//   call's random feature.  The decoder is the NICE `MLP` (5 x 32 ReLU with
//   the feature added after every layer, Fourier features sin(2 pi p B):
//   2 pi is folded into the packed B) — mlp_fwd / mlp_bwd<.,32,1> as they
//   are.  One wave = 16 samples; the distances are recomputed from the
//   positions so that they carry the pose gradient (is_tracker, :181-186).
__device__ __forceinline__ void point_feature(
    const PointNb& nb, const float* __restrict__ feats,
    const uint8_t* __restrict__ fmask, const float* __restrict__ empty, int q,
    f32x4 (&c)[1][2]) {
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  c[0][0] = z4;
  c[0][1] = z4;
  if (!nb.has) {
    c[0][0] = *reinterpret_cast<const f32x4*>(empty + 4 * q);
    c[0][1] = *reinterpret_cast<const f32x4*>(empty + 16 + 4 * q);
    return;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (nb.u[k] == 0.f) continue;
    if (fmask != nullptr && fmask[nb.id[k]] == 0) continue;
    const float w = nb.u[k] / nb.den;
    const float* f = feats + (int64_t)nb.id[k] * 32 + 4 * q;
    c[0][0] += *reinterpret_cast<const f32x4*>(f) * w;
    c[0][1] += *reinterpret_cast<const f32x4*>(f + 16) * w;
  }
}

__global__ __launch_bounds__(8 * 64, 2) void point_geo_fwd_kernel(
    int64_t n, const float* __restrict__ pts, const int64_t* __restrict__ nbr,
    const int* __restrict__ n_nb, const float* __restrict__ cloud,
    const float* __restrict__ feats, const uint8_t* __restrict__ fmask,
    const float* __restrict__ radius, float radius_all, int min_nn,
    const float* __restrict__ empty, const float* __restrict__ dec,
    float* __restrict__ occ, uint8_t* __restrict__ has_out,
    uint64_t* __restrict__ masks) {
  using PM = MlpPack<32, 1>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  stage_weights(wl, dec, PM::WHT);
  const int64_t ngroups = (n + 127) / 128;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t pt = (grp * 8 + wave) * 16 + li;
    const bool valid = pt < n;
    float p32[1][3] = {{0.f, 0.f, 0.f}};
    if (valid) {
#pragma unroll
      for (int a = 0; a < 3; ++a) p32[0][a] = pts[pt * 3 + a];
    }
    PointNb nb;
    point_neighbors(nbr, cloud, n_nb, radius, radius_all, min_nn, pt, valid,
                    p32[0], nb);
    f32x4 c[1][2];
    point_feature(nb, feats, fmask, empty, q, c);
    float out[1][1];
    uint64_t mask[1];
    mlp_fwd<1, 32, 1, true, false>(wl, lane, p32, c, out, mask, nullptr);
    if (valid) {
      if (q == 0) {
        occ[pt] = out[0][0];
        has_out[pt] = nb.has ? 1 : 0;
      }
      if (masks != nullptr) masks[pt * 4 + q] = mask[0];
    }
  }
}

// per-wave transposition tile of the feature-gradient scatter: the tail of
// the staged-fragment region (the 32-wide decoder fills well under half of it)
constexpr int kGeoTileLen = 16 * 33;
constexpr int kGeoTile = kWMax - 8 * kGeoTileLen;
static_assert(MlpPack<32, 1>::LEN - MlpPack<32, 1>::EMB <= kGeoTile,
              "geometry decoder + scatter tiles fit");

template <bool NEED_DP, bool NEED_DF>
__global__ __launch_bounds__(8 * 64, 2) void point_geo_bwd_kernel(
    int64_t n, const float* __restrict__ pts, const int64_t* __restrict__ nbr,
    const int* __restrict__ n_nb, const float* __restrict__ cloud,
    const float* __restrict__ feats, const uint8_t* __restrict__ fmask,
    const float* __restrict__ radius, float radius_all, int min_nn,
    const float* __restrict__ empty, const float* __restrict__ dec,
    const uint64_t* __restrict__ masks, const float* __restrict__ g_occ,
    float* __restrict__ g_pts, float* __restrict__ g_feats) {
  using PM = MlpPack<32, 1>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = lane >> 4, li = lane & 15;
  stage_weights(wl, dec + PM::EMB, (NEED_DP ? PM::LEN : PM::W0T) - PM::EMB);
  const int64_t ngroups = (n + 127) / 128;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t pt = (grp * 8 + wave) * 16 + li;
    const bool valid = pt < n;
    float p32[1][3] = {{0.f, 0.f, 0.f}};
    if (valid) {
#pragma unroll
      for (int a = 0; a < 3; ++a) p32[0][a] = pts[pt * 3 + a];
    }
    PointNb nb;
    point_neighbors(nbr, cloud, n_nb, radius, radius_all, min_nn, pt, valid,
                    p32[0], nb);
    f32x4 c[1][2], gc[1][2];
    point_feature(nb, feats, fmask, empty, q, c);
    const float go[1][1] = {{valid ? g_occ[pt] : 0.f}};
    const uint64_t mask[1] = {valid ? masks[pt * 4 + q] : 0};
    float gp[1][3] = {{0.f, 0.f, 0.f}};
    mlp_bwd<1, 32, 1, NEED_DP, NEED_DP>(wl - PM::EMB, lane, p32, c, go, mask,
                                        gc, gp);
    float gpos[3] = {0.f, 0.f, 0.f};
    if (NEED_DP) {
#pragma unroll
      for (int a = 0; a < 3; ++a) gpos[a] = group4_sum(gp[0][a]);
    }
    // interpolation backward (nothing flows through the random feature)
    if (NEED_DF) {
      // feature gradient: the tile's d/dc goes through LDS so that 32
      // consecutive lanes add the 32 features of ONE neighbour (two 128-byte
      // rows per atomic instruction instead of 16 rows x 16 bytes)
      float* T = wl + kGeoTile + wave * kGeoTileLen;   // [16][33]
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T[li * 33 + 4 * q + r] = gc[0][0][r];
        T[li * 33 + 16 + 4 * q + r] = gc[0][1][r];
      }
      wave_lds_sync();
      const int f = lane & 31, half = lane >> 5;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool live = nb.has && nb.u[k] != 0.f &&
                          (fmask == nullptr || fmask[nb.id[k]] != 0);
        const float wk = live ? nb.u[k] / nb.den : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int pt2 = 2 * i + half;   // lane pt2 = (q 0, li pt2)
          const float ww = __shfl(wk, pt2);
          const int id2 = __shfl(nb.id[k], pt2);
          if (ww != 0.f)
            atomicAdd(g_feats + (int64_t)id2 * 32 + f, ww * T[pt2 * 33 + f]);
        }
      }
      wave_lds_sync();
    }
    float gw[8];
    float aw = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gw[k] = 0.f;
      const bool live = nb.has && nb.u[k] != 0.f &&
                        (fmask == nullptr || fmask[nb.id[k]] != 0);
      const float w = nb.u[k] / nb.den;
      if (live) {
        const float* f = feats + (int64_t)nb.id[k] * 32 + 4 * q;
        if (NEED_DP) {
          const f32x4 f0 = *reinterpret_cast<const f32x4*>(f);
          const f32x4 f1 = *reinterpret_cast<const f32x4*>(f + 16);
          float d = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) d += gc[0][0][r] * f0[r] + gc[0][1][r] * f1[r];
          gw[k] = d;
        }
      }
      if (NEED_DP) {
        gw[k] = group4_sum(gw[k]);   // every lane group holds 8 of 32 features
        aw += gw[k] * w;
      }
    }
    if (NEED_DP) {
      if (nb.has) {
        const bool norm = nb.den > 1e-12f;   // else the clamp: constant
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (nb.u[k] == 0.f) continue;
          const float gu = (gw[k] - (norm ? aw : 0.f)) / nb.den;
          const float gD = -nb.u[k] * nb.u[k] * gu;
          const int64_t id = nb.id[k];
#pragma unroll
          for (int a = 0; a < 3; ++a)
            gpos[a] += 2.f * (p32[0][a] - cloud[id * 3 + a]) * gD;
        }
      }
      if (valid && q == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) g_pts[pt * 3 + a] = gpos[a];
      }
    }
  }
}

// g_dec = sum of the dW replicas; g_rays_{o,d}[ray] = sum of the ray's tile
// partials (fixed order: deterministic)
__global__ __launch_bounds__(256) void nice_bwd_finish_kernel(
    const float* __restrict__ rep, int len, float* __restrict__ g_dec,
    const double* __restrict__ part, int n, int nt,
    float* __restrict__ g_rays_o, float* __restrict__ g_rays_d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < len) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < kDwRep; ++r) s += rep[(size_t)r * len + i];
    g_dec[i] = s;
    return;
  }
  const int j = i - len;
  if (j >= n * 6) return;
  const int ray = j / 6, a = j % 6;
  double s = 0.0;
  for (int t = 0; t < nt; ++t) s += part[((size_t)ray * nt + t) * 6 + a];
  float* dst = a < 3 ? g_rays_o + ray * 3 + a : g_rays_d + ray * 3 + a - 3;
  *dst = (float)s;
}

__global__ void mfma_selftest_kernel(const float* a, const float* b,
                                     float* out) {
  const int l = threadIdx.x;
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = XRD_MFMA4(a[(l & 15) * 4 + (l >> 4)], b[(l >> 4) * 16 + (l & 15)], d);
#pragma unroll
  for (int r = 0; r < 4; ++r) out[((l >> 4) * 4 + r) * 16 + (l & 15)] = d[r];
}

// persistent blocks of the fused backward: one per CU (the staged fragments
// take most of its LDS)
constexpr int kFusedBlocks = 256;

}  // namespace
}  // namespace xrd

using namespace xrd;

extern "C" {

int xrd_nice_flat_len(int kind) {
  switch (kind) {
    case XRD_DEC_COARSE: return NoXyzFlat::LEN;
    case XRD_DEC_MIDDLE: return MlpFlat<32, 1>::LEN;
    case XRD_DEC_FINE: return MlpFlat<64, 1>::LEN;
    case XRD_DEC_COLOR: return MlpFlat<32, 4>::LEN;
  }
  return -1;
}

int xrd_nice_pack_len(int kind) {
  switch (kind) {
    case XRD_DEC_COARSE: return NoXyzPack::LEN;
    case XRD_DEC_MIDDLE: return MlpPack<32, 1>::LEN;
    case XRD_DEC_FINE: return MlpPack<64, 1>::LEN;
    case XRD_DEC_COLOR: return MlpPack<32, 4>::LEN;
  }
  return -1;
}

int xrd_nice_pack_index(int kind, int32_t* idx) {
  if (idx == nullptr) return XRD_ERR_ARG;
  switch (kind) {
    case XRD_DEC_COARSE: build_noxyz_index(idx); return XRD_OK;
    case XRD_DEC_MIDDLE: build_mlp_index<32, 1>(idx); return XRD_OK;
    case XRD_DEC_FINE: build_mlp_index<64, 1>(idx); return XRD_OK;
    case XRD_DEC_COLOR: build_mlp_index<32, 4>(idx); return XRD_OK;
  }
  return XRD_ERR_ARG;
}

static int nice_check(const xrd_nice_scene* sc, int stage, int n,
                      const float* gt_depth, int* nt) {
  if (sc == nullptr || n < 0 || stage < 0 || stage > 3) return XRD_ERR_ARG;
  const bool has_d = gt_depth != nullptr && stage != XRD_STAGE_COARSE;
  const int S = sc->n_samples + (has_d ? sc->n_surface : 0);
  if (S != 32 && S != 48) return XRD_ERR_UNSUPPORTED;
  if (sc->t_uniform == nullptr || (has_d && sc->n_surface > 0 &&
                                   sc->t_surface == nullptr))
    return XRD_ERR_ARG;
  const int need[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 1, 1, 0}, {0, 1, 1, 1}};
  for (int g = 0; g < 4; ++g)
    if (need[stage][g] && (sc->grid[g] == nullptr || sc->dec[g] == nullptr))
      return XRD_ERR_ARG;
  *nt = S / 16;
  return XRD_OK;
}

}  // extern "C"

template <int ST, int NTV, int RPBV>
static int launch_fwd(const xrd_nice_scene* scene, int n, const float* rays_o,
                      const float* rays_d, const float* gt_depth,
                      const float* dmax, double* depth, double* var,
                      float* rgb, float* raw_out, hipStream_t st) {
  auto kern = nice_fwd_kernel<ST, NTV, RPBV>;
  const size_t lds = fwd_lds_floats<NTV, RPBV>() * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute");
    attr_set = true;
  }
  if (n == 0) return XRD_OK;  // warm-up call: attributes only
  constexpr int rpb = RPBV;
  const int ngroups = (n + rpb - 1) / rpb;
  const int nb = ngroups < kFusedBlocks ? ngroups : kFusedBlocks;
  hipLaunchKernelGGL(kern, dim3(nb), dim3(rpb * NTV * 64), lds, st, *scene, n,
                     rays_o, rays_d, gt_depth, dmax, depth, var, rgb, raw_out);
  return check_launch("xrd_nice_render_fwd");
}

#define FWD_CASE(ST, NTV)                                                    \
  {                                                                          \
    if (wide)                                                                \
      return launch_fwd<ST, NTV, fwd_rays_wide<NTV>()>(                      \
          scene, n_rays, rays_o, rays_d, gt_depth, dmax, depth, var, rgb,    \
          raw_out, st);                                                      \
    return launch_fwd<ST, NTV, 1>(scene, n_rays, rays_o, rays_d, gt_depth,   \
                                  dmax, depth, var, rgb, raw_out, st);       \
  }

static int fwd_dispatch(const xrd_nice_scene* scene, int stage, int nt,
                        int n_rays, const float* rays_o, const float* rays_d,
                        const float* gt_depth, const float* dmax,
                        double* depth, double* var, float* rgb,
                        float* raw_out, hipStream_t st, int wide = -1) {
  if (wide < 0) wide = n_rays >= 512;  // else one ray per block
  switch (stage * 4 + nt) {
    case XRD_STAGE_COARSE * 4 + 2: FWD_CASE(XRD_STAGE_COARSE, 2);
    case XRD_STAGE_MIDDLE * 4 + 2: FWD_CASE(XRD_STAGE_MIDDLE, 2);
    case XRD_STAGE_MIDDLE * 4 + 3: FWD_CASE(XRD_STAGE_MIDDLE, 3);
    case XRD_STAGE_FINE * 4 + 2: FWD_CASE(XRD_STAGE_FINE, 2);
    case XRD_STAGE_FINE * 4 + 3: FWD_CASE(XRD_STAGE_FINE, 3);
    case XRD_STAGE_COLOR * 4 + 2: FWD_CASE(XRD_STAGE_COLOR, 2);
    case XRD_STAGE_COLOR * 4 + 3: FWD_CASE(XRD_STAGE_COLOR, 3);
    default: return XRD_ERR_UNSUPPORTED;
  }
}

extern "C" {

int xrd_nice_render_fwd(const xrd_nice_scene* scene, int stage, int n_rays,
                        const float* rays_o, const float* rays_d,
                        const float* gt_depth, const float* dmax,
                        double* depth, double* var, float* rgb, float* raw_out,
                        xrd_stream_t stream) {
  int nt = 0;
  int rc = nice_check(scene, stage, n_rays, gt_depth, &nt);
  if (rc != XRD_OK) return rc;
  if (!rays_o || !rays_d || !depth || !var || !rgb) return XRD_ERR_ARG;
  if (gt_depth && stage != XRD_STAGE_COARSE && !dmax) return XRD_ERR_ARG;
  if (n_rays == 0) return XRD_OK;
  hipStream_t st = (hipStream_t)stream;
  if (stage == XRD_STAGE_COARSE) gt_depth = nullptr;
  return fwd_dispatch(scene, stage, nt, n_rays, rays_o, rays_d, gt_depth,
                      dmax, depth, var, rgb, raw_out, st);
}

int64_t xrd_nice_coarse_ws_floats(const xrd_nice_scene* scene) {
  if (scene == nullptr) return -1;
  return (int64_t)kCoarseRep * scene->gdim[0] * scene->gdim[1] *
         scene->gdim[2] * 32;
}

int xrd_nice_eval_points(const xrd_nice_scene* scene, int stage,
                         int64_t n_points, const float* points, float* raw,
                         xrd_stream_t stream) {
  if (scene == nullptr || n_points < 0) return XRD_ERR_ARG;
  if (stage != XRD_STAGE_FINE && stage != XRD_STAGE_COLOR)
    return XRD_ERR_UNSUPPORTED;
  for (int g = 1; g <= stage; ++g)
    if (scene->grid[g] == nullptr || scene->dec[g] == nullptr)
      return XRD_ERR_ARG;
  if (n_points == 0) return XRD_OK;
  if (!points || !raw) return XRD_ERR_ARG;
  const size_t lds = (size_t)kWMax * sizeof(float);
  auto kf = nice_points_kernel<XRD_STAGE_FINE>;
  auto kc = nice_points_kernel<XRD_STAGE_COLOR>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kf),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kc),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute");
    attr_set = true;
  }
  const int64_t ngroups = (n_points + 127) / 128;
  const int nb = (int)(ngroups < 512 ? ngroups : 512);
  if (stage == XRD_STAGE_FINE)
    hipLaunchKernelGGL(kf, dim3(nb), dim3(512), lds, (hipStream_t)stream,
                       *scene, n_points, points, raw);
  else
    hipLaunchKernelGGL(kc, dim3(nb), dim3(512), lds, (hipStream_t)stream,
                       *scene, n_points, points, raw);
  return check_launch("xrd_nice_eval_points");
}

static int point_geo_attr() {
  static bool done = false;
  if (done) return XRD_OK;
  const int lds = (int)(kWMax * sizeof(float));
  const void* ks[] = {
      reinterpret_cast<const void*>(point_geo_fwd_kernel),
      reinterpret_cast<const void*>(point_geo_bwd_kernel<false, true>),
      reinterpret_cast<const void*>(point_geo_bwd_kernel<true, false>),
      reinterpret_cast<const void*>(point_geo_bwd_kernel<true, true>)};
  for (const void* k : ks)
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute");
  done = true;
  return XRD_OK;
}

int xrd_point_geo_fwd(int64_t n_points, const float* points,
                      const int64_t* neighbors, const int32_t* n_neighbors,
                      const float* cloud, const float* geo_feats,
                      const uint8_t* feat_mask, const float* radius,
                      float radius_all, int min_nn, const float* empty_feat,
                      const float* packed_decoder, float* occ, uint8_t* has,
                      uint64_t* relu_masks, xrd_stream_t stream) {
  if (n_points < 0 || min_nn < 0) return XRD_ERR_ARG;
  if (n_points == 0) return XRD_OK;
  if (!points || !neighbors || !n_neighbors || !cloud || !geo_feats ||
      !empty_feat || !packed_decoder || !occ || !has)
    return XRD_ERR_ARG;
  int rc = point_geo_attr();
  if (rc != XRD_OK) return rc;
  const int64_t ngroups = (n_points + 127) / 128;
  const int nb = (int)(ngroups < 512 ? ngroups : 512);
  hipLaunchKernelGGL(point_geo_fwd_kernel, dim3(nb), dim3(512),
                     kWMax * sizeof(float), (hipStream_t)stream, n_points,
                     points, neighbors, n_neighbors, cloud, geo_feats,
                     feat_mask, radius, radius_all, min_nn, empty_feat,
                     packed_decoder, occ, has, relu_masks);
  return check_launch("xrd_point_geo_fwd");
}

int xrd_point_geo_bwd(int64_t n_points, const float* points,
                      const int64_t* neighbors, const int32_t* n_neighbors,
                      const float* cloud, const float* geo_feats,
                      const uint8_t* feat_mask, const float* radius,
                      float radius_all, int min_nn, const float* empty_feat,
                      const float* packed_decoder, const uint64_t* relu_masks,
                      const float* g_occ, float* g_points, float* g_geo_feats,
                      xrd_stream_t stream) {
  if (n_points < 0 || min_nn < 0) return XRD_ERR_ARG;
  if (n_points == 0) return XRD_OK;
  if (!points || !neighbors || !n_neighbors || !cloud || !geo_feats ||
      !empty_feat || !packed_decoder || !relu_masks || !g_occ)
    return XRD_ERR_ARG;
  if (!g_points && !g_geo_feats) return XRD_OK;
  int rc = point_geo_attr();
  if (rc != XRD_OK) return rc;
  const int64_t ngroups = (n_points + 127) / 128;
  const dim3 grid((unsigned)(ngroups < 512 ? ngroups : 512)), block(512);
  const size_t lds = kWMax * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define XRD_PG_ARGS                                                          \
  n_points, points, neighbors, n_neighbors, cloud, geo_feats, feat_mask,     \
      radius, radius_all, min_nn, empty_feat, packed_decoder, relu_masks,    \
      g_occ, g_points, g_geo_feats
  if (g_points && g_geo_feats)
    hipLaunchKernelGGL((point_geo_bwd_kernel<true, true>), grid, block, lds,
                       st, XRD_PG_ARGS);
  else if (g_points)
    hipLaunchKernelGGL((point_geo_bwd_kernel<true, false>), grid, block, lds,
                       st, XRD_PG_ARGS);
  else
    hipLaunchKernelGGL((point_geo_bwd_kernel<false, true>), grid, block, lds,
                       st, XRD_PG_ARGS);
#undef XRD_PG_ARGS
  return check_launch("xrd_point_geo_bwd");
}

int64_t xrd_nice_bwd_ws_floats(int n_rays) {
  // per-tile ray-gradient partials (6 doubles each, at most 3 tiles a ray) +
  // the replicas of the colour-decoder gradient
  return (int64_t)n_rays * 3 * 6 * 2 + (int64_t)kDwRep * kColorFlat + 64;
}

}  // extern "C"

static size_t fused_lds_bytes(bool dw) {
  return fused_lds_floats(dw) * sizeof(float);
}

template <int ST, int NTV, bool DP, bool DW, int W>
static int launch_fused(const xrd_nice_scene* scene, int n,
                        const float* rays_o, const float* rays_d,
                        const float* gt_depth, const float* dmax,
                        const float* raw, const double* g_depth,
                        const double* g_var, const float* g_rgb,
                        float* const gg[4], double* part, float* dw_rep,
                        hipStream_t st) {
  auto kern = nice_bwd_fused_kernel<ST, NTV, DP, DW, W>;
  const size_t lds =
      (DW ? fused_lds_floats(true) : (size_t)kWMax + W * kScratch) *
      sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute");
    attr_set = true;
  }
  if (n == 0) return XRD_OK;  // warm-up call: attributes only
  int64_t ngroups = ((int64_t)n * NTV + W - 1) / W;
  const int nb = (int)(ngroups < kFusedBlocks ? ngroups : kFusedBlocks);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(W * 64), lds, st, *scene, n, rays_o,
                     rays_d, gt_depth, dmax, raw, g_depth, g_var, g_rgb, gg[1],
                     gg[2], gg[3], part, dw_rep);
  return check_launch("xrd_nice_render_bwd");
}

// waves per block: as wide as the batch still spreads over >= ~half the CUs;
// 16 only without pose gradients (128 registers a lane); weight gradients: 8
static int fused_width(int64_t tiles, bool dp, bool dw) {
  if (dw) return FWD;
  if (!dp && tiles >= 16 * 128) return 16;
  if (tiles >= 8 * 128) return 8;
  return 4;
}

#define FUSED_ARGS                                                          \
  scene, n_rays, rays_o, rays_d, gt_depth, dmax, raw, g_depth, g_var, g_rgb, \
      gg, part, dw_rep, st

template <int ST, int NTV>
static int fused_dispatch_st(int width, bool dp, bool dw,
                             const xrd_nice_scene* scene, int n_rays,
                             const float* rays_o, const float* rays_d,
                             const float* gt_depth, const float* dmax,
                             const float* raw, const double* g_depth,
                             const double* g_var, const float* g_rgb,
                             float* const gg[4], double* part, float* dw_rep,
                             hipStream_t st) {
  if constexpr (ST == XRD_STAGE_COLOR) {
    if (dw) {
      if (dp) return launch_fused<ST, NTV, true, true, FWD>(FUSED_ARGS);
      return launch_fused<ST, NTV, false, true, FWD>(FUSED_ARGS);
    }
  }
  if (dp) {
    if (width >= 8) return launch_fused<ST, NTV, true, false, 8>(FUSED_ARGS);
    return launch_fused<ST, NTV, true, false, 4>(FUSED_ARGS);
  }
  if (width >= 16) return launch_fused<ST, NTV, false, false, 16>(FUSED_ARGS);
  if (width >= 8) return launch_fused<ST, NTV, false, false, 8>(FUSED_ARGS);
  return launch_fused<ST, NTV, false, false, 4>(FUSED_ARGS);
}

static int fused_dispatch(const xrd_nice_scene* scene, int stage, int nt,
                          bool dp, bool dw, int n_rays, const float* rays_o,
                          const float* rays_d, const float* gt_depth,
                          const float* dmax, const float* raw,
                          const double* g_depth, const double* g_var,
                          const float* g_rgb, float* const gg[4], double* part,
                          float* dw_rep, hipStream_t st, int width = 0) {
  if (width == 0) width = fused_width((int64_t)n_rays * nt, dp, dw);
#define FUSED_ST(ST, NTV) \
  return fused_dispatch_st<ST, NTV>(width, dp, dw, FUSED_ARGS)
  if (stage == XRD_STAGE_MIDDLE) {
    if (nt == 3) FUSED_ST(XRD_STAGE_MIDDLE, 3);
    FUSED_ST(XRD_STAGE_MIDDLE, 2);
  }
  if (stage == XRD_STAGE_FINE) {
    if (nt == 3) FUSED_ST(XRD_STAGE_FINE, 3);
    FUSED_ST(XRD_STAGE_FINE, 2);
  }
  if (nt == 3) FUSED_ST(XRD_STAGE_COLOR, 3);
  FUSED_ST(XRD_STAGE_COLOR, 2);
#undef FUSED_ST
}

template <int W>
static int roles_attr_w() {
  if (hipFuncSetAttribute(
          reinterpret_cast<const void*>(nice_bwd_roles_kernel<3, W>),
          hipFuncAttributeMaxDynamicSharedMemorySize,
          (int)(role_lds_floats<W>() * sizeof(float))) != hipSuccess ||
      hipFuncSetAttribute(
          reinterpret_cast<const void*>(nice_fwd_roles_kernel<3, W>),
          hipFuncAttributeMaxDynamicSharedMemorySize,
          (int)(role_fwd_lds_floats<W>() * sizeof(float))) != hipSuccess)
    return check_launch("hipFuncSetAttribute");
  return XRD_OK;
}

static int roles_attr() {
  static bool attr = false;
  if (!attr) {
    if (int rc = roles_attr_w<8>(); rc != XRD_OK) return rc;
    if (int rc = roles_attr_w<12>(); rc != XRD_OK) return rc;
    attr = true;
  }
  return XRD_OK;
}

template <int W>
static void launch_fwd_roles(const xrd_nice_scene* scene, int n_rays,
                             const float* rays_o, const float* rays_d,
                             const float* gt_depth, const float* dmax,
                             float* raw_out, uint64_t* masks, float* occ_m,
                             hipStream_t st) {
  hipLaunchKernelGGL((nice_fwd_roles_kernel<3, W>),
                     dim3((n_rays * 3 + W - 1) / W, 3), dim3(W * 64),
                     role_fwd_lds_floats<W>() * sizeof(float), st, *scene,
                     n_rays, rays_o, rays_d, gt_depth, dmax, raw_out, masks,
                     occ_m);
}

template <int W>
static void launch_bwd_roles(const xrd_nice_scene* scene, int n_rays,
                             const float* rays_o, const float* rays_d,
                             const float* gt_depth, const float* dmax,
                             const float* raw, const double* g_depth,
                             const double* g_var, const float* g_rgb,
                             double* part, const uint64_t* masks,
                             hipStream_t st) {
  hipLaunchKernelGGL((nice_bwd_roles_kernel<3, W>),
                     dim3((n_rays * 3 + W - 1) / W, 3), dim3(W * 64),
                     role_lds_floats<W>() * sizeof(float), st, *scene, n_rays,
                     rays_o, rays_d, gt_depth, dmax, raw, g_depth, g_var,
                     g_rgb, part, masks);
}

extern "C" {

static bool masks_supported(int stage, int nt, int n_rays,
                            const float* gt_depth) {
  return stage == XRD_STAGE_COLOR && nt == 3 && n_rays <= kRoleMaxRays &&
         gt_depth != nullptr;
}

int64_t xrd_nice_fwd_masks_words(int n_rays) {
  // ReLU masks [(ray * 3 + tile) * 3 + decoder][64], then the middle
  // decoder's occupancy [ray][64] f32 (the hand-over between the decoder
  // blocks of the forward and its finishing launch)
  return n_rays < 0 ? -1 : (int64_t)n_rays * (3 * 3 * 64 + 32);
}

int xrd_nice_render_fwd_masks(const xrd_nice_scene* scene, int stage,
                              int n_rays, const float* rays_o,
                              const float* rays_d, const float* gt_depth,
                              const float* dmax, double* depth, double* var,
                              float* rgb, float* raw_out, uint64_t* masks,
                              xrd_stream_t stream) {
  int nt = 0;
  int rc = nice_check(scene, stage, n_rays, gt_depth, &nt);
  if (rc != XRD_OK) return rc;
  if (!rays_o || !rays_d || !depth || !var || !rgb || !masks)
    return XRD_ERR_ARG;
  if (!masks_supported(stage, nt, n_rays, gt_depth))
    return XRD_ERR_UNSUPPORTED;
  if (!dmax) return XRD_ERR_ARG;
  if (!raw_out) return XRD_ERR_ARG;  // the decoder blocks meet in it
  rc = roles_attr();
  if (rc != XRD_OK) return rc;
  if (n_rays == 0) return XRD_OK;
  hipStream_t st = (hipStream_t)stream;
  float* occ_m = reinterpret_cast<float*>(masks + (size_t)n_rays * 3 * 3 * 64);
  if (role_waves(n_rays) == 8)
    launch_fwd_roles<8>(scene, n_rays, rays_o, rays_d, gt_depth, dmax, raw_out,
                        masks, occ_m, st);
  else
    launch_fwd_roles<12>(scene, n_rays, rays_o, rays_d, gt_depth, dmax,
                         raw_out, masks, occ_m, st);
  rc = check_launch("xrd_nice_render_fwd_masks");
  if (rc != XRD_OK) return rc;
  hipLaunchKernelGGL(nice_fwd_roles_finish_kernel<3>, dim3((n_rays + 3) / 4),
                     dim3(256), 0, st, *scene, n_rays, rays_o, rays_d,
                     gt_depth, dmax, occ_m, raw_out, depth, var, rgb);
  return check_launch("xrd_nice_render_fwd_masks");
}

static int render_bwd_impl(const xrd_nice_scene* scene, int stage, int n_rays,
                           const float* rays_o, const float* rays_d,
                           const float* gt_depth, const float* dmax,
                           const float* raw, const double* g_depth,
                           const double* g_var, const float* g_rgb,
                           const uint64_t* masks, float* g_rays_o,
                           float* g_rays_d, float* const g_grid[4],
                           float* const g_dec[4], float* ws,
                           xrd_stream_t stream) {
  int nt = 0;
  int rc = nice_check(scene, stage, n_rays, gt_depth, &nt);
  if (rc != XRD_OK) return rc;
  if (!rays_o || !rays_d || !raw) return XRD_ERR_ARG;
  if ((g_rays_o == nullptr) != (g_rays_d == nullptr)) return XRD_ERR_ARG;
  if (gt_depth && stage != XRD_STAGE_COARSE && !dmax) return XRD_ERR_ARG;
  float* gg[4] = {nullptr, nullptr, nullptr, nullptr};
  if (g_grid)
    for (int g = 0; g < 4; ++g) gg[g] = g_grid[g];
  bool dw = false;
  if (g_dec) {
    if (g_dec[XRD_DEC_COARSE] || g_dec[XRD_DEC_MIDDLE] || g_dec[XRD_DEC_FINE])
      return XRD_ERR_UNSUPPORTED;
    dw = g_dec[XRD_DEC_COLOR] != nullptr;
  }
  if (dw && stage != XRD_STAGE_COLOR) return XRD_ERR_ARG;
  const bool dp = g_rays_o != nullptr;
  if (stage == XRD_STAGE_COARSE && dp) return XRD_ERR_UNSUPPORTED;
  if ((dw || dp) && ws == nullptr) return XRD_ERR_ARG;
  if (n_rays == 0) return XRD_OK;
  hipStream_t st = (hipStream_t)stream;
  if (stage == XRD_STAGE_COARSE) {
    if (nt != 2) return XRD_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(
              reinterpret_cast<const void*>(nice_bwd_coarse_kernel),
              hipFuncAttributeMaxDynamicSharedMemorySize,
              (int)kBwdCoarseLds) != hipSuccess)
        return check_launch("hipFuncSetAttribute");
      attr_set = true;
    }
    const int ngroups = (n_rays + kBwdCoarseRPB - 1) / kBwdCoarseRPB;
    const int nb = ngroups < kMaxBwdBlocks ? ngroups : kMaxBwdBlocks;
    hipLaunchKernelGGL(nice_bwd_coarse_kernel, dim3(nb),
                       dim3(kBwdCoarseWaves * 64), kBwdCoarseLds, st, *scene,
                       n_rays, rays_o, rays_d, raw, g_depth, g_var, g_rgb,
                       gg[0], ws);
    rc = check_launch("xrd_nice_render_bwd/coarse");
    if (rc != XRD_OK) return rc;
    if (ws != nullptr && gg[0] != nullptr) {
      const int64_t ne = (int64_t)scene->gdim[0] * scene->gdim[1] *
                         scene->gdim[2] * 32;
      hipLaunchKernelGGL(coarse_rep_reduce_kernel,
                         dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st,
                         ws, ne, gg[0]);
      return check_launch("coarse_rep_reduce_kernel");
    }
    return XRD_OK;
  }
  // workspace: [n_rays * nt][6] f64 tile partials | kDwRep gradient replicas
  double* part = reinterpret_cast<double*>(ws);
  float* dw_rep = ws ? ws + (size_t)n_rays * 3 * 6 * 2 : nullptr;
  if (dw) {
    rc = zero_floats(dw_rep, (size_t)kDwRep * kColorFlat, stream);
    if (rc != XRD_OK) return rc;
  }
  int nt_rows = nt;
  if (stage == XRD_STAGE_COLOR && nt == 3 && dp && !dw &&
      n_rays <= kRoleMaxRays && gt_depth != nullptr && !gg[1] && !gg[2] &&
      !gg[3]) {
    // tracking: one decoder per block (the part rows of the three roles
    // extend into the — unused — replica region of the workspace)
    rc = roles_attr();
    if (rc != XRD_OK) return rc;
    if (role_waves(n_rays) == 8)
      launch_bwd_roles<8>(scene, n_rays, rays_o, rays_d, gt_depth, dmax, raw,
                          g_depth, g_var, g_rgb, part, masks, st);
    else
      launch_bwd_roles<12>(scene, n_rays, rays_o, rays_d, gt_depth, dmax, raw,
                           g_depth, g_var, g_rgb, part, masks, st);
    rc = check_launch("xrd_nice_render_bwd/roles");
    nt_rows = 9;
  } else if (masks != nullptr) {
    return XRD_ERR_UNSUPPORTED;  // masks: the ray-gradient-only tracking path
  } else {
    rc = fused_dispatch(scene, stage, nt, dp, dw, n_rays, rays_o, rays_d,
                        gt_depth, dmax, raw, g_depth, g_var, g_rgb, gg, part,
                        dw_rep, st);
  }
  if (rc != XRD_OK) return rc;
  if (dw || dp) {
    const int len = dw ? kColorFlat : 0;
    const int total = len + (dp ? n_rays * 6 : 0);
    hipLaunchKernelGGL(nice_bwd_finish_kernel, dim3((total + 255) / 256),
                       dim3(256), 0, st, dw_rep, len,
                       dw ? g_dec[XRD_DEC_COLOR] : nullptr, part,
                       dp ? n_rays : 0, nt_rows, g_rays_o, g_rays_d);
    return check_launch("xrd_nice_render_bwd/finish");
  }
  return XRD_OK;
}

int xrd_nice_render_bwd(const xrd_nice_scene* scene, int stage, int n_rays,
                        const float* rays_o, const float* rays_d,
                        const float* gt_depth, const float* dmax,
                        const float* raw, const double* g_depth,
                        const double* g_var, const float* g_rgb,
                        float* g_rays_o, float* g_rays_d,
                        float* const g_grid[4], float* const g_dec[4],
                        float* ws, xrd_stream_t stream) {
  return render_bwd_impl(scene, stage, n_rays, rays_o, rays_d, gt_depth, dmax,
                         raw, g_depth, g_var, g_rgb, nullptr, g_rays_o,
                         g_rays_d, g_grid, g_dec, ws, stream);
}

int xrd_nice_render_bwd_masks(const xrd_nice_scene* scene, int stage,
                              int n_rays, const float* rays_o,
                              const float* rays_d, const float* gt_depth,
                              const float* dmax, const float* raw,
                              const double* g_depth, const double* g_var,
                              const float* g_rgb, const uint64_t* masks,
                              float* g_rays_o, float* g_rays_d, float* ws,
                              xrd_stream_t stream) {
  if (!masks || !g_rays_o || !g_rays_d) return XRD_ERR_ARG;
  return render_bwd_impl(scene, stage, n_rays, rays_o, rays_d, gt_depth, dmax,
                         raw, g_depth, g_var, g_rgb, masks, g_rays_o,
                         g_rays_d, nullptr, nullptr, ws, stream);
}

int xrd_nice_warmup(void) {
  float* gg[4] = {nullptr, nullptr, nullptr, nullptr};
  if (int rc = roles_attr(); rc != XRD_OK) return rc;
  xrd_nice_scene sc = {};
  for (int stage = XRD_STAGE_COARSE; stage <= XRD_STAGE_COLOR; ++stage)
    for (int nt = 2; nt <= 3; ++nt) {
      if (stage == XRD_STAGE_COARSE && nt != 2) continue;
      for (int wide = 0; wide < 2; ++wide) {
        int rc = fwd_dispatch(&sc, stage, nt, 0, nullptr, nullptr, nullptr,
                              nullptr, nullptr, nullptr, nullptr, nullptr,
                              nullptr, wide);
        if (rc != XRD_OK) return rc;
      }
    }
  for (int stage = XRD_STAGE_MIDDLE; stage <= XRD_STAGE_COLOR; ++stage)
    for (int nt = 2; nt <= 3; ++nt)
      for (int dp = 0; dp < 2; ++dp)
        for (int dw = 0; dw < 2; ++dw) {
          if (stage != XRD_STAGE_COLOR && dw) continue;
          for (int width = 4; width <= 16; width *= 2) {
            int rc = fused_dispatch(&sc, stage, nt, dp, dw, 0, nullptr,
                                    nullptr, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, gg, nullptr,
                                    nullptr, nullptr, width);
            if (rc != XRD_OK) return rc;
          }
        }
  return XRD_OK;
}

int xrd_selftest_mfma(const float* a, const float* b, float* out,
                      xrd_stream_t stream) {
  if (!a || !b || !out) return XRD_ERR_ARG;
  hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0,
                     (hipStream_t)stream, a, b, out);
  return check_launch("xrd_selftest_mfma");
}

}  // extern "C"
