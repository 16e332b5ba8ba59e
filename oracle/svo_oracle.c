/* TEST INFRASTRUCTURE ONLY — CPU restatement (plain C) of the two live CUDA
 * kernels of the reference's `grid` extension:
 *
 *   svo_intersect_point_kernel   third_party/sparse_voxels/src/intersect_gpu.cu:191-270
 *     + RayAABBIntersection      :75-140
 *   inverse_cdf_sampling_kernel  third_party/sparse_voxels/src/sample_gpu.cu:133-239
 *
 * including their quirks (DFS push order child 0..7 / LIFO pop, n_max cut-off,
 * the -1 miss sentinel, `(~done)` always true, `pts_idx[curr_bin]` without the
 * ray offset, the `num_rays > H + curr_bin` guard).  PINNED: the reference's
 * own two kernels are compiled for the host by oracle/build_ref_grid.py
 * (oracle/_ref/sparse_voxels/grid_ref.so); tests/test_svo_oracle.py checks
 * this restatement bit for bit against them and against the vectors generated
 * from them (tests/golden/svo_grid.npz).  `__fdividef(1,x)` is the exact
 * 1.0f/x in both; the one expression nvcc contracts is an explicit fmaf.  Outputs must be pre-filled like
 * the host wrappers do (intersect.cpp:98-106, sample.cpp:77-86): idx -1 is
 * written by the kernel itself; sampled_idx = -1, depths/dists = 0.
 *
 * Build: oracle/build_oracle.py (gcc -O2 -ffp-contract=off -shared).        */
#include <math.h>
#include <stdint.h>

static void ray_aabb(const float* o, const float* d, const float* c,
                     float half, float* lo, float* hi) {
  float f_low = 0.f, f_high = 100000.f;
  for (int k = 0; k < 3; ++k) {
    const float inv = 1.0f / d[k];
    float a = (c[k] - half - o[k]) * inv;
    float b = (c[k] + half - o[k]) * inv;
    if (b < a) { float t = a; a = b; b = t; }
    if (b < f_low) { *lo = -1.f; *hi = -1.f; return; }
    if (a > f_high) { *lo = -1.f; *hi = -1.f; return; }
    f_low = (a > f_low) ? a : f_low;
    f_high = (b < f_high) ? b : f_high;
    if (f_low > f_high) { *lo = -1.f; *hi = -1.f; return; }
  }
  *lo = f_low;
  *hi = f_high;
}

/* returns the deepest stack pointer seen (the kernel asserts ptr < 256) */
int svo_intersect_ref(int b, int n, int m, float voxelsize, int n_max,
                      const float* ray_start, const float* ray_dir,
                      const float* points, const int* children, int* idx,
                      float* min_depth, float* max_depth) {
  int deepest = 0;
  const float half_voxel = voxelsize * 0.5;
  for (int bi = 0; bi < b; ++bi) {
    const float* P = points + (int64_t)bi * n * 3;
    const int* C = children + (int64_t)bi * n * 9;
    const float* RS = ray_start + (int64_t)bi * m * 3;
    const float* RD = ray_dir + (int64_t)bi * m * 3;
    int* I = idx + (int64_t)bi * m * n_max;
    float* MN = min_depth + (int64_t)bi * m * n_max;
    float* MX = max_depth + (int64_t)bi * m * n_max;
    for (int j = 0; j < m; ++j) {
      for (int l = 0; l < n_max; ++l) I[j * n_max + l] = -1;
      int stack[256];
      int ptr = 0, cnt = 0, k;
      stack[0] = 0;
      while (ptr > -1 && cnt < n_max) {
        if (ptr > deepest) deepest = ptr;
        k = stack[ptr];
        float lo, hi;
        ray_aabb(RS + j * 3, RD + j * 3, P + k * 3,
                 half_voxel * (float)C[k * 9 + 8], &lo, &hi);
        ptr--;
        if (lo > -1.0f) {
          if (C[k * 9 + 8] == 1) {
            I[j * n_max + cnt] = k;
            MN[j * n_max + cnt] = lo;
            MX[j * n_max + cnt] = hi;
            ++cnt;
            continue;
          }
          for (int u = 0; u < 8; ++u)
            if (C[k * 9 + u] > -1) stack[++ptr] = C[k * 9 + u];
        }
      }
    }
  }
  return deepest;
}

void inverse_cdf_sampling_ref(int b, int num_rays, int max_hits, int max_steps,
                              float fixed_step_size, const int* pts_idx,
                              const float* min_depth, const float* max_depth,
                              const float* uniform_noise, const float* probs,
                              const float* steps, int* sampled_idx,
                              float* sampled_depth, float* sampled_dists) {
  for (int bi = 0; bi < b; ++bi) {
    const int* PI = pts_idx + (int64_t)bi * num_rays * max_hits;
    const float* MN = min_depth + (int64_t)bi * num_rays * max_hits;
    const float* MX = max_depth + (int64_t)bi * num_rays * max_hits;
    const float* PR = probs + (int64_t)bi * num_rays * max_hits;
    const float* ST = steps + (int64_t)bi * num_rays;
    const float* UN = uniform_noise + (int64_t)bi * num_rays * max_steps;
    int* SI = sampled_idx + (int64_t)bi * num_rays * max_steps;
    float* SD = sampled_depth + (int64_t)bi * num_rays * max_steps;
    float* SS = sampled_dists + (int64_t)bi * num_rays * max_steps;
    for (int j = 0; j < num_rays; ++j) {
      const int H = j * max_hits, K = j * max_steps;
      int curr_bin = 0, s = 0;
      float curr_min_depth = MN[H], curr_max_depth = MX[H];
      float curr_min_cdf = 0, curr_max_cdf = PR[H];
      float step_size = 1.0 / ST[j];
      float z_low = curr_min_depth;
      const int total_steps = (int)ceil(ST[j]);
      int done = 0;
      if (fixed_step_size > 0.0) step_size = fixed_step_size;
      for (int curr_step = 0; curr_step < total_steps; curr_step++) {
        const float curr_cdf = ((float)curr_step + UN[K + curr_step]) * step_size;
        while (curr_cdf > curr_max_cdf) {
          SI[K + s] = PI[H + curr_bin];
          SS[K + s] = (curr_max_depth - z_low);
          SD[K + s] = (curr_max_depth + z_low) * .5;
          curr_bin++;
          s++;
          if ((curr_bin >= max_hits) || (PI[H + curr_bin] == -1)) {
            done = 1;
            break;
          }
          curr_min_depth = MN[H + curr_bin];
          curr_max_depth = MX[H + curr_bin];
          curr_min_cdf = curr_max_cdf;
          curr_max_cdf = curr_max_cdf + PR[H + curr_bin];
          z_low = curr_min_depth;
        }
        if (done) break;
        const float u = (curr_cdf - curr_min_cdf) / (curr_max_cdf - curr_min_cdf);
        /* nvcc's default --fmad=true contracts this expression */
        const float z = fmaf(u, curr_max_depth - curr_min_depth, curr_min_depth);
        SI[K + s] = PI[H + curr_bin];
        SS[K + s] = (z - z_low);
        SD[K + s] = (z + z_low) * .5;
        z_low = z;
        s++;
      }
      /* "(~done)" is always true in the reference (bitwise not of a bool) */
      while ((z_low < curr_max_depth) && (num_rays > (H + curr_bin))) {
        SI[K + s] = PI[H + curr_bin];
        SS[K + s] = (curr_max_depth - z_low);
        SD[K + s] = (curr_max_depth + z_low) * .5;
        curr_bin++;
        s++;
        /* reference quirk: pts_idx[curr_bin] lacks the ray offset H */
        if ((curr_bin >= max_hits) || (PI[curr_bin] == -1)) break;
        curr_min_depth = MN[H + curr_bin];
        curr_max_depth = MX[H + curr_bin];
        z_low = curr_min_depth;
      }
    }
  }
}
