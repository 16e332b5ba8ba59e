"""TEST INFRASTRUCTURE ONLY.  Writes tests/golden/svo_grid.npz: inputs and
outputs of the REFERENCE's svo_intersect / inverse_cdf_sampling kernels,
executed on the host through oracle/build_ref_grid.py (the reference's own
.cu kernel bodies compiled with g++).  Run in the build container (needs
/root/reference); the vectors let boxes without the reference tree pin the
restatements and the HIP kernels to the reference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE, os.path.join(ROOT, 'tests')]

from svo_util import (inverse_cdf_ref, make_tree, sampler_case,  # noqa: E402
                      svo_intersect_ref)


def main():
    centres, childs = make_tree(11, 4000)
    rng = np.random.default_rng(12)
    B, M, n_max = 2, 300, 50
    o = (np.array([[13.0, 13.0, 9.0]]) + rng.uniform(-1, 1, (B, M, 3))).astype(
        np.float32)
    d = rng.standard_normal((B, M, 3)).astype(np.float32)
    d[..., 2] = np.abs(d[..., 2]) + 0.1
    pts = np.tile(centres[None], (B, 1, 1))
    ch = np.tile(childs[None], (B, 1, 1))
    idx, mn, mx = svo_intersect_ref(o, d, pts, ch, 0.2, n_max)
    hit = idx >= 0
    mn[~hit] = 0
    mx[~hit] = 0
    # the reference wrapper's geometry: G = 200 batches of ceil(N/200) rays
    args = sampler_case(13, 200, n_rays=520, n_vox=4000)
    s_idx, s_depth, s_dists = inverse_cdf_ref(*args, 0.0)
    out = os.path.join(ROOT, 'tests', 'golden', 'svo_grid.npz')
    np.savez_compressed(
        out, ray_start=o, ray_dir=d, points1=centres, children1=childs,
        voxelsize=np.float32(0.2), n_max=np.int32(n_max), idx=idx,
        min_depth=mn, max_depth=mx,
        **{'s_' + k: v for k, v in zip(
            ('pts_idx', 'min_depth', 'max_depth', 'noise', 'probs', 'steps'),
            args)}, s_idx=s_idx, s_depth=s_depth, s_dists=s_dists)
    print(out, os.path.getsize(out), 'bytes;', int(hit.sum()), 'hits,',
          int((s_idx >= 0).sum()), 'samples')


if __name__ == '__main__':
    main()
