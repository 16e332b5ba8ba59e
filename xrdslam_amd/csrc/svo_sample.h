// Inverse-CDF ray sampling of Vox-Fusion, one WAVE per ray — the body shared by
// xrd_inverse_cdf_sampling (svo.hip: the reference's [G,R,P] tensors) and the
// fused ray pipeline (vox_rays.hip: static-capacity rows, logical sizes read
// from the device).  Reference: third_party/sparse_voxels/src/sample_gpu.cu:
// 133-239 walks a ray's steps serially on one thread, carrying (bin, z_low);
// tests/svo_parallel_model.py states and checks the re-formulation used here:
//   * cum[b] = serial float prefix sum of the ray's probs (same addition
//     order as the reference), kept in LDS;
//   * lane c owns step c: cdf(c), bin(c) = first b with !(cdf > cum[b])
//     (running max over the lanes by a wave scan), its in-bin sample goes to
//     slot c + bin(c); the bin boundaries crossed since step c-1 go to slots
//     c + b; z_low comes from lane c-1 by shuffle when it lies in the same
//     bin, else it is the bin's entry depth;
//   * the first lane whose bin reaches the number of valid bins ends the ray
//     ("done" in the reference);
//   * the reference's trailing loop over the remaining bins, with its quirks
//     (`~done` always true, `pts_idx[curr_bin]` read without the ray offset,
//     the `num_rays > H + curr_bin` guard), runs on lane 0.
//
// The caller supplies the ray's own rows (MN, MX, PR: max_hits entries), the
// prefix-sum scratch `cum` (LDS, >= max_hits floats, private to the wave) and
//   pi(i)    pts_idx at FLAT index i of the ray's batch ([num_rays, max_hits]
//            row-major; the ray's own entries are i = H + b), -1 beyond it;
//   noise(c) the uniform draw of step c;
//   emit(slot, id, z_hi, z_lo)   one sample (the caller drops slots beyond
//            its row length, like the reference's max_steps cut-off).
#pragma once
#include "common.h"

// keep the float expressions as written (bit-exact sample depths); file scope:
// also holds for the including translation unit from here on
#pragma clang fp contract(off)

namespace xrd {

template <class Pi, class Noise, class Emit>
__device__ __forceinline__ void inverse_cdf_ray(
    int lane, float* __restrict__ cum, int max_hits, int num_rays, int H,
    const float* __restrict__ MN, const float* __restrict__ MX,
    const float* __restrict__ PR, float st, float fixed_step_size, Pi pi,
    Noise noise, Emit emit) {
  // valid bins: bin 0 always, then up to the first -1
  int nbv = max_hits;
  for (int b0 = 0; b0 < max_hits; b0 += 64) {
    const int b = b0 + lane;
    const bool stop = b >= 1 && b < max_hits && pi(H + b) == -1;
    const uint64_t m = __ballot(stop);
    if (m) {
      nbv = b0 + __builtin_ctzll(m);
      break;
    }
  }
  {  // serial prefix sum, one writer
    float acc = 0.f;
    for (int b = 0; b < nbv; ++b) {
      acc = acc + PR[b];
      if (lane == 0) cum[b] = acc;
    }
  }
  wave_lds_sync();
  float step_size = (float)(1.0 / (double)st);
  if (fixed_step_size > 0.0) step_size = fixed_step_size;
  const int total = (int)ceil((double)st);
  int carry_bin = 0;         // bin of the last step of the previous round
  float carry_z = MN[0];     // its z
  int run_bin = 0;
  bool done = false;
  int tail_bin = 0, tail_s = 0;
  float tail_zlow = carry_z, tail_max = MX[0];
  for (int base = 0; base < total; base += 64) {
    const int c = base + lane;
    const bool act = c < total;
    int f = 0;
    float cdf = 0.f;
    if (act) {
      cdf = ((float)c + noise(c)) * step_size;
      while (f < nbv && cdf > cum[f]) ++f;
    }
    int bn = f;  // inclusive running max over the lanes, then the carry
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(bn, o);
      if (lane >= o) bn = bn > u ? bn : u;
    }
    bn = bn > run_bin ? bn : run_bin;
    const uint64_t dmask = __ballot(act && bn >= nbv);
    const int first_done = dmask ? __builtin_ctzll(dmask) : 64;
    const bool mine = act && lane <= first_done;
    const bool is_done = lane == first_done;
    float z = 0.f;
    if (mine && !is_done) {
      const float cmin = bn > 0 ? cum[bn - 1] : 0.f;
      const float u = (cdf - cmin) / (cum[bn] - cmin);
      const float lo = MN[bn];
      z = fmaf(u, MX[bn] - lo, lo);  // nvcc contracts this expression
    }
    int pb = __shfl_up(bn, 1);
    float pz = __shfl_up(z, 1);
    if (lane == 0) {
      pb = carry_bin;
      pz = carry_z;
    }
    if (mine) {
      const int hi = bn < nbv ? bn : nbv;
      for (int b = pb; b < hi; ++b)  // boundaries crossed since step c-1
        emit(c + b, pi(H + b), MX[b], b == pb ? pz : MN[b]);
      if (!is_done) emit(c + bn, pi(H + bn), z, bn == pb ? pz : MN[bn]);
    }
    if (first_done < 64) {
      // state after the reference's `done` break, from the done lane
      const float zl = (nbv - 1 == pb) ? pz : MN[nbv - 1];
      tail_zlow = __shfl(zl, first_done);
      tail_bin = nbv;
      tail_s = base + first_done + nbv;
      tail_max = MX[nbv - 1];
      done = true;
      break;
    }
    const int last = (total - base < 64 ? total - base : 64) - 1;
    carry_bin = __shfl(bn, last);
    carry_z = __shfl(z, last);
    run_bin = carry_bin;
  }
  if (!done) {
    tail_bin = carry_bin;
    tail_s = total + carry_bin;
    tail_zlow = carry_z;
    tail_max = MX[carry_bin];
  }
  if (lane != 0) return;
  while (tail_zlow < tail_max && num_rays > H + tail_bin) {
    emit(tail_s, pi(H + tail_bin), tail_max, tail_zlow);
    ++tail_bin;
    ++tail_s;
    if (tail_bin >= max_hits || pi(tail_bin) == -1) break;
    tail_max = MX[tail_bin];
    tail_zlow = MN[tail_bin];
  }
}

}  // namespace xrd
