// Point-SLAM: the neighbour weights shared by the geometry path
// (nice_render.hip) and the colour path (point_color.hip).
//   get_feature_at_pos (slam/model_components/decoder_pointslam.py:162-209,
//   408-470): squared distances to the <= 8 neighbours recomputed from the
//   positions (they carry the pose gradient), u = 1 / (D + 1e-10), zero beyond
//   the query radius or where the search found no neighbour, w = u / max(sum u,
//   1e-12); a sample needs min_nn neighbours inside the radius (as counted by
//   the search), else it takes the call's random feature.
#pragma once
#include "common.h"

namespace xrd {

struct PointNb {
  int id[8];
  float u[8];    // 1/(D + 1e-10), 0 beyond the radius / missing
  float den;     // max(sum u, 1e-12)
  bool has;
};

__device__ __forceinline__ void point_neighbors(
    const int64_t* __restrict__ nbr, const float* __restrict__ cloud,
    const int* __restrict__ n_nb, const float* __restrict__ radius,
    float radius_all, int min_nn, int64_t pt, bool valid,
    const float (&p)[3], PointNb& nb) {
  float S = 0.f;
  const float r = valid ? (radius ? radius[pt] : radius_all) : 0.f;
  const float bound = r * r;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t id = valid ? nbr[pt * 8 + k] : -1;
    nb.id[k] = (int)id;
    nb.u[k] = 0.f;
    if (id >= 0) {
      const float dx = cloud[id * 3] - p[0], dy = cloud[id * 3 + 1] - p[1],
                  dz = cloud[id * 3 + 2] - p[2];
      const float D = dx * dx + dy * dy + dz * dz;
      if (!(D > bound)) nb.u[k] = 1.f / (D + 1e-10f);
    }
    S += nb.u[k];
  }
  nb.den = fmaxf(S, 1e-12f);
  nb.has = valid && n_nb[pt] > min_nn - 1;
}

}  // namespace xrd
