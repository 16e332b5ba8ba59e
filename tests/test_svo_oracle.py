"""CPU checks of the oracles of svo_intersect / inverse_cdf_sampling:

* the REFERENCE's own kernels compiled for the host
  (oracle/build_ref_grid.py -> oracle/_ref/sparse_voxels/grid_ref.so) are the
  pin; tests/golden/svo_grid.npz holds vectors generated from them
  (oracle/make_golden_svo.py) for boxes without the reference tree;
* the plain-C restatement (oracle/svo_oracle.c) and the wave-cooperative
  re-formulation the HIP kernel uses (tests/svo_parallel_model.py) must equal
  the compiled reference bit for bit — ids, depths and distances."""
import os

import numpy as np
import pytest

from svo_parallel_model import inverse_cdf_parallel_model
from svo_util import (inverse_cdf_oracle, inverse_cdf_ref, make_tree, ref_lib,
                      sampler_case, svo_intersect_oracle, svo_intersect_ref)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                    'svo_grid.npz')
needs_ref = pytest.mark.skipif(ref_lib() is None,
                               reason='compiled reference kernels not built')


def test_intersect_oracle_properties():
    centres, childs = make_tree(0)
    rng = np.random.default_rng(1)
    M = 200
    o = np.tile(np.array([[13.0, 13.0, 9.0]], np.float32), (M, 1))
    d = rng.standard_normal((M, 3)).astype(np.float32)
    d[:, 2] = np.abs(d[:, 2]) + 0.2
    idx, mn, mx, deepest = svo_intersect_oracle(
        o[None], d[None], centres[None], childs[None], 0.2, 50)
    assert deepest < 64  # 256^3 trees: 1 + 7*8 = 57 entries at most
    hit = idx[0] >= 0
    assert hit.any()
    assert (childs[idx[0][hit], 8] == 1).all()  # terminal nodes only
    assert (mx[0][hit] >= mn[0][hit]).all() and (mn[0][hit] >= 0).all()
    first_pad = np.argmax(~hit, 1)
    for r in range(M):
        if (~hit[r]).any():
            assert not hit[r, first_pad[r]:].any()


def test_inverse_cdf_oracle_single_bin():
    # one ray, one bin [1,2], prob 1, 4 steps, noise 0.5 -> mid-points
    pts = np.array([[[7, -1]]], np.int32)
    mn = np.array([[[1.0, 0.0]]], np.float32)
    mx = np.array([[[2.0, 0.0]]], np.float32)
    probs = np.array([[[1.0, 0.0]]], np.float32)
    steps = np.array([[4.0]], np.float32)
    noise = np.full((1, 1, 6), 0.5, np.float32)
    sidx, sdep, sdis = inverse_cdf_oracle(pts, mn, mx, noise, probs, steps,
                                          0.0)
    assert list(sidx[0, 0, :4]) == [7, 7, 7, 7]
    z = 1.0 + (np.arange(4) + 0.5) / 4
    zl = np.concatenate([[1.0], z[:-1]])
    assert np.allclose(sdep[0, 0, :4], (z + zl) / 2)
    assert np.allclose(sdis[0, 0, :4], z - zl)
    # the trailing "remaining bins" loop emits the last interval up to max
    assert sidx[0, 0, 4] == 7 and np.isclose(sdis[0, 0, 4], 2.0 - z[-1])


def _rays(M, seed):
    rng = np.random.default_rng(seed)
    o = (np.array([[13.0, 13.0, 9.0]]) + rng.uniform(-1, 1, (M, 3))).astype(
        np.float32)
    d = rng.standard_normal((M, 3)).astype(np.float32)
    d[:, 2] = np.abs(d[:, 2]) + 0.1
    return o, d


@needs_ref
@pytest.mark.parametrize('B,M,n_max', [(1, 1, 50), (1, 700, 50), (3, 77, 50),
                                       (2, 130, 3)])
def test_intersect_restatement_equals_compiled_reference(B, M, n_max):
    centres, childs = make_tree(B + M)
    o = np.stack([_rays(M, 10 + b)[0] for b in range(B)])
    d = np.stack([_rays(M, 10 + b)[1] for b in range(B)])
    pts = np.tile(centres[None], (B, 1, 1))
    ch = np.tile(childs[None], (B, 1, 1))
    ridx, rmn, rmx = svo_intersect_ref(o, d, pts, ch, 0.2, n_max)
    oidx, omn, omx, _ = svo_intersect_oracle(o, d, pts, ch, 0.2, n_max)
    assert np.array_equal(oidx, ridx)
    hit = ridx >= 0
    assert np.array_equal(omn[hit], rmn[hit])
    assert np.array_equal(omx[hit], rmx[hit])


@needs_ref
@pytest.mark.parametrize('seed,G,det', [(0, 1, False), (1, 4, False),
                                        (2, 200, False), (3, 3, True)])
def test_sampler_restatements_equal_compiled_reference(seed, G, det):
    """both the serial C restatement and the parallel re-formulation
    reproduce the compiled reference, including the rows where the trailing
    loop's un-offset ``pts_idx[curr_bin]`` / ``num_rays > H + curr_bin``
    quirks fire (small ray indices, full hit rows)"""
    args = sampler_case(seed, G, deterministic=det)
    ref = inverse_cdf_ref(*args, 0.0)
    for fn in (inverse_cdf_oracle, inverse_cdf_parallel_model):
        got = fn(*args, 0.0)
        for a, b, name in zip(got, ref, ('idx', 'depth', 'dists')):
            assert np.array_equal(a, b), (fn.__name__, name)
    assert (ref[0] >= 0).sum() > 5 * args[0].shape[0] * args[0].shape[1]


@needs_ref
def test_sampler_fixed_step_size():
    args = sampler_case(5, 2)
    ref = inverse_cdf_ref(*args, 0.004)
    for fn in (inverse_cdf_oracle, inverse_cdf_parallel_model):
        got = fn(*args, 0.004)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), fn.__name__


def test_oracles_match_committed_reference_vectors():
    """tests/golden/svo_grid.npz was written by oracle/make_golden_svo.py from
    the compiled reference; holds on boxes without /root/reference too"""
    g = np.load(GOLD)
    B = g['ray_start'].shape[0]
    oidx, omn, omx, _ = svo_intersect_oracle(
        g['ray_start'], g['ray_dir'], np.tile(g['points1'][None], (B, 1, 1)),
        np.tile(g['children1'][None], (B, 1, 1)),
        float(g['voxelsize']), int(g['n_max']))
    assert np.array_equal(oidx, g['idx'])
    hit = oidx >= 0
    assert np.array_equal(omn[hit], g['min_depth'][hit])
    assert np.array_equal(omx[hit], g['max_depth'][hit])
    args = [g['s_' + k] for k in ('pts_idx', 'min_depth', 'max_depth',
                                  'noise', 'probs', 'steps')]
    for fn in (inverse_cdf_oracle, inverse_cdf_parallel_model):
        got = fn(*args, 0.0)
        assert np.array_equal(got[0], g['s_idx']), fn.__name__
        assert np.array_equal(got[1], g['s_depth']), fn.__name__
        assert np.array_equal(got[2], g['s_dists']), fn.__name__
