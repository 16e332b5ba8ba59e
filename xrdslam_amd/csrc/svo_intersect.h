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

}  // namespace xrd
