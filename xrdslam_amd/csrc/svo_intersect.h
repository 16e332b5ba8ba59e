// Ray / sparse-voxel-octree intersection of ONE ray by one wave (shared by
// svo.hip's xrd_svo_intersect and the one-launch ray pipeline of vox_rays.hip).
// Semantics: svo_intersect_point_kernel, third_party/sparse_voxels/src/
// intersect_gpu.cu:191-270 (hit order, n_max cut-off); see svo.hip.
#pragma once
#include "common.h"

#pragma clang fp contract(off)  // keep the float expressions as written

namespace xrd {

constexpr int kSvoStack = 128;   // >= 1 + 7 * levels; 256^3 trees need 57

__device__ __forceinline__ void ray_aabb(const float (&o)[3],
                                         const float (&d)[3], const float* c,
                                         float half, float& lo, float& hi) {
  float f_low = 0.f, f_high = 100000.f;
  lo = hi = -1.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float inv = 1.0f / d[k];
    float a = (c[k] - half - o[k]) * inv;
    float b = (c[k] + half - o[k]) * inv;
    if (b < a) {
      const float t = a;
      a = b;
      b = t;
    }
    if (b < f_low) return;
    if (a > f_high) return;
    f_low = (a > f_low) ? a : f_low;
    f_high = (b < f_high) ? b : f_high;
    if (f_low > f_high) return;
  }
  lo = f_low;
  hi = f_high;
}

// The reference pops a node, tests its box and pushes all of its children;
// here the 8 children of a popped node are tested at once on lanes 0-7 and
// only the ones the ray enters are pushed, in child order — the nodes whose
// boxes are hit are therefore visited in the same LIFO order, and leaves are
// recorded in the same order with the same (lo, hi), while a ray takes one
// serial step per HIT internal node instead of one per node looked at (8x
// fewer dependent loads).  Round 6: a step was still TWO dependent round
// trips to L2 (the popped node's child ids, then the children's sizes and
// centres) and a ray is one wave's chain of 10-20 such steps — 24 of the ray
// pipeline's 52 us.  The child ids of a node now travel with its stack entry
// (s_kids): when a child is pushed, the eight lanes 8 j .. 8 j + 7 fetch ITS
// child ids in the same round trip that fetches its size and centre, so a
// pop reads them from LDS and a step is ONE round trip.  Same nodes, same
// order, same (lo, hi).  s_*: this wave's DFS stack in LDS [kSvoStack],
// s_kids [kSvoStack][8]; emit(slot, node, lo, hi) is called by EVERY lane for
// each recorded leaf.  Returns the number of leaves recorded; overflow: stack
// overflow.
template <class Emit>
__device__ __forceinline__ int svo_intersect_ray(
    int lane, int* s_node, int* s_side, float* s_lo, float* s_hi, int* s_kids,
    const float (&o)[3], const float (&d)[3], const float* __restrict__ P,
    const int* __restrict__ C, float voxelsize, int n_max, bool& overflow,
    Emit emit) {
  const float half_voxel = voxelsize * 0.5;
  int ptr = -1, cnt = 0;
  overflow = false;
  {  // root is node 0
    const int side = C[8];
    const int kid = lane < 8 ? C[lane] : -1;
    float lo, hi;
    ray_aabb(o, d, P, half_voxel * (float)side, lo, hi);
    if (lo > -1.0f) {
      ptr = 0;
      if (lane == 0) {
        s_node[0] = 0;
        s_side[0] = side;
        s_lo[0] = lo;
        s_hi[0] = hi;
      }
      if (lane < 8) s_kids[lane] = kid;
    }
  }
  wave_lds_sync();
  while (ptr > -1 && cnt < n_max) {
    const int k = s_node[ptr];
    const int side = s_side[ptr];
    if (side == 1) {  // terminal node
      emit(cnt, k, s_lo[ptr], s_hi[ptr]);
      ++cnt;
      --ptr;
      continue;
    }
    int c = -1, cs = 0;
    float lo = -1.f, hi = -1.f;
    if (lane < 8) c = s_kids[ptr * 8 + lane];
    --ptr;
    // lane 8 j + g: child j's child g (fetched next to child j's size and
    // centre; used only if child j is pushed)
    const int cj = __shfl(c, lane >> 3);
    int kid = -1;
    if (cj > -1) kid = C[cj * 9 + (lane & 7)];
    if (c > -1) {
      cs = C[c * 9 + 8];
      ray_aabb(o, d, P + c * 3, half_voxel * (float)cs, lo, hi);
    }
    const uint64_t mask = __ballot(c > -1 && lo > -1.0f);
    const int n_push = __popcll(mask);
    if (ptr + 1 + n_push > kSvoStack) {
      overflow = true;
      break;
    }
    wave_lds_sync();  // every lane has read the popped entry
    if ((mask >> lane) & 1) {
      const int at = ptr + 1 + __popcll(mask & ((1ull << lane) - 1));
      s_node[at] = c;
      s_side[at] = cs;
      s_lo[at] = lo;
      s_hi[at] = hi;
    }
    if ((mask >> (lane >> 3)) & 1) {
      const int j = lane >> 3;
      const int at = ptr + 1 + __popcll(mask & ((1ull << j) - 1));
      s_kids[at * 8 + (lane & 7)] = kid;
    }
    ptr += n_push;
    wave_lds_sync();
  }
  return cnt;
}

// ---- level-synchronous walk (round 6) ------------------------------------------
// The depth-first walk above is one ray's chain of 20-40 DEPENDENT L2 round
// trips (one per internal node the ray enters): 24 us of latency with the GPU
// idle, 45 times a Vox-Fusion frame.  The same set of leaves is found level by
// level: the wave holds the FRONTIER of internal nodes the ray enters on one
// level (their child ids in LDS), tests 16 of them x 8 children per pass — all
// loads of a pass in flight together — and a ray costs one or two round trips a
// LEVEL (8-10 in a 256^3 tree).  The reference's output order (the LIFO order
// of its stack: at every node the entered children are visited in DESCENDING
// child order) and its n_max cut-off are restored from a path key — 3 bits a
// level, (7 - child), left-aligned — by ranking the collected leaves: same
// leaves, same order, same (lo, hi) as the depth-first walk, bit for bit.
// Returns false (nothing emitted) when a capacity below is exceeded, the tree
// is deeper than the key or the root is a leaf: the caller then runs the
// depth-first walk.
constexpr int kBfsFront = 56;    // internal nodes a ray enters on one level
constexpr int kBfsLeaves = 128;  // leaves collected before the n_max cut
constexpr int kBfsLevels = 10;   // 3 bits of key a level
constexpr int kSvoLds = 12 * kSvoStack;   // ints of LDS a wave needs (both walks)
static_assert(2 * kBfsFront * 9 <= 8 * kSvoStack && 3 * kBfsLeaves <= 8 * kSvoStack &&
              kBfsLeaves <= kSvoStack, "the walks share a wave's LDS");

// s: kSvoLds ints of this wave.  take(cnt, node, lo, hi): the first cnt =
// min(leaves, n_max) hits in the reference's order, in LDS arrays (valid until
// the wave's next call).
template <class Take>
__device__ __forceinline__ bool svo_intersect_ray_bfs(
    int lane, int* s, const float (&o)[3], const float (&d)[3],
    const float* __restrict__ P, const int* __restrict__ C, float voxelsize,
    int n_max, Take take) {
  const float half_voxel = voxelsize * 0.5;
  int* l_key = s;
  int* l_node = s + kSvoStack;
  float* l_lo = reinterpret_cast<float*>(s + 2 * kSvoStack);
  float* l_hi = reinterpret_cast<float*>(s + 3 * kSvoStack);
  int* fr = s + 4 * kSvoStack;   // two frontiers: key [F] | kids [F][8]
  constexpr int FB = kBfsFront * 9;
  const int j = lane >> 3, g = lane & 7;
  const uint64_t below = (1ull << lane) - 1;
  int nf = 0, nl = 0;
  wave_lds_sync();   // the previous ray's results have been read
  {  // root is node 0
    const int side = C[8];
    if (side == 1) return false;
    const int kid = lane < 8 ? C[lane] : -1;
    float lo, hi;
    ray_aabb(o, d, P, half_voxel * (float)side, lo, hi);
    if (lo > -1.0f) {
      nf = 1;
      if (lane == 0) fr[0] = 0;
      if (lane < 8) fr[kBfsFront + lane] = kid;
    }
  }
  wave_lds_sync();
  int cur = 0;
  for (int level = 0; nf > 0; ++level) {
    if (level >= kBfsLevels) return false;
    const int shift = 3 * (kBfsLevels - 1 - level);
    const int* f_key = fr + cur * FB;
    const int* f_kids = f_key + kBfsFront;
    int* n_key = fr + (cur ^ 1) * FB;
    int* n_kids = n_key + kBfsFront;
    int nn = 0;
    for (int f0 = 0; f0 < nf; f0 += 16) {
      int c[2], key[2], cs[2], k8[2][8];
      float ctr[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int fi = f0 + 8 * u + j;
        c[u] = fi < nf ? f_kids[fi * 8 + g] : -1;
        key[u] = fi < nf ? (f_key[fi] | ((7 - g) << shift)) : 0;
      }
      // every load of the pass before any use (clamped ids: no branch)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ci = c[u] > -1 ? c[u] : 0;
        cs[u] = C[ci * 9 + 8];
#pragma unroll
        for (int t = 0; t < 3; ++t) ctr[u][t] = P[ci * 3 + t];
#pragma unroll
        for (int t = 0; t < 8; ++t) k8[u][t] = C[ci * 9 + t];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float lo, hi;
        ray_aabb(o, d, ctr[u], half_voxel * (float)cs[u], lo, hi);
        const bool hit = c[u] > -1 && lo > -1.0f;
        const bool leaf = hit && cs[u] == 1, inner = hit && cs[u] != 1;
        const uint64_t mL = __ballot(leaf), mI = __ballot(inner);
        const int cL = __popcll(mL), cI = __popcll(mI);
        if (nl + cL > kBfsLeaves || nn + cI > kBfsFront) return false;
        if (leaf) {
          const int at = nl + __popcll(mL & below);
          l_key[at] = key[u];
          l_node[at] = c[u];
          l_lo[at] = lo;
          l_hi[at] = hi;
        }
        if (inner) {
          const int at = nn + __popcll(mI & below);
          n_key[at] = key[u];
#pragma unroll
          for (int t = 0; t < 8; ++t) n_kids[at * 8 + t] = k8[u][t];
        }
        nl += cL;
        nn += cI;
      }
    }
    wave_lds_sync();
    cur ^= 1;
    nf = nn;
  }
  // the reference's order: ascending path key (keys are distinct)
  int* o_node = fr;
  float* o_lo = reinterpret_cast<float*>(fr + kBfsLeaves);
  float* o_hi = reinterpret_cast<float*>(fr + 2 * kBfsLeaves);
  for (int i = lane; i < nl; i += 64) {
    const int k = l_key[i];
    int r = 0;
    for (int t = 0; t < nl; ++t) r += l_key[t] < k ? 1 : 0;
    o_node[r] = l_node[i];
    o_lo[r] = l_lo[i];
    o_hi[r] = l_hi[i];
  }
  wave_lds_sync();
  take(nl < n_max ? nl : n_max, o_node, o_lo, o_hi);
  return true;
}

}  // namespace xrd
