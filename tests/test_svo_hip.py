"""GPU parity of the Vox-Fusion native ops (grid shim -> HIP kernels): voxel
and sample ids BIT-EXACT; depths within 1e-6 relative for the intersection
(the reference's __fdividef is approximate on NVIDIA anyway), bit-exact for
the sampler.  Checked against (a) vectors produced by the REFERENCE's own
kernels compiled for the host (tests/golden/svo_grid.npz,
oracle/make_golden_svo.py), (b) those compiled kernels directly when
oracle/_ref/sparse_voxels/grid_ref.so travelled to this box, (c) the C
restatement, which the CPU suite pins to (a) and (b)."""
import os

import numpy as np
import pytest
import torch

from svo_util import (inverse_cdf_oracle, inverse_cdf_ref, make_tree, ref_lib,
                      sampler_case, svo_intersect_oracle)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                    'svo_grid.npz')

pytestmark = pytest.mark.gpu


def _rays(M, seed):
    rng = np.random.default_rng(seed)
    o = (np.array([[13.0, 13.0, 9.0]]) + rng.uniform(-1, 1, (M, 3))).astype(
        np.float32)
    d = rng.standard_normal((M, 3)).astype(np.float32)
    d[:, 2] = np.abs(d[:, 2]) + 0.1
    return o, d


@pytest.mark.parametrize('B,M', [(1, 1), (1, 1024), (3, 77)])
def test_svo_intersect_bit_exact(B, M):
    from xrdslam_amd.compat import grid
    centres, childs = make_tree(B + M)
    o = np.stack([_rays(M, 10 + b)[0] for b in range(B)])
    d = np.stack([_rays(M, 10 + b)[1] for b in range(B)])
    pts = np.tile(centres[None], (B, 1, 1))
    ch = np.tile(childs[None], (B, 1, 1))
    ridx, rmn, rmx, _ = svo_intersect_oracle(o, d, pts, ch, 0.2, 50)
    c = lambda a: torch.from_numpy(a).cuda()
    idx, mn, mx = grid.svo_intersect(c(o), c(d), c(pts), c(ch), 0.2, 50)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.allclose(mn.cpu().numpy(), rmn, rtol=1e-6, atol=0)
    assert np.allclose(mx.cpu().numpy(), rmx, rtol=1e-6, atol=0)
    # one shared tree for all batches gives the same answer
    idx2, mn2, mx2 = grid.svo_intersect(c(o), c(d), c(centres[None]),
                                        c(childs[None]), 0.2, 50)
    if B > 1:
        assert torch.equal(idx2, idx) and torch.equal(mn2, mn)
    # n_max cut-off
    idx3, _, _ = grid.svo_intersect(c(o), c(d), c(pts), c(ch), 0.2, 3)
    r3, _, _, _ = svo_intersect_oracle(o, d, pts, ch, 0.2, 3)
    assert np.array_equal(idx3.cpu().numpy(), r3)


def _chain_tree(levels):
    """a root of side 2**levels over ONE chain of single children down to a
    2x2x2 block of leaves: deeper than the level-synchronous walk's path key
    (10 levels) when levels > 10"""
    nodes = []          # (corner xyz in voxels, side)
    corner = np.zeros(3, np.int64)
    for lv in range(levels, 0, -1):
        nodes.append((corner.copy(), 2**lv))
        if lv > 1:      # descend into child 5 = (1, 0, 1)
            corner = corner + np.array([1, 0, 1]) * 2**(lv - 1)
    n_int = len(nodes)
    children = -np.ones((n_int + 8, 9), np.int32)
    centres = np.zeros((n_int + 8, 3), np.float32)
    for i, (c, side) in enumerate(nodes):
        centres[i] = (c + side / 2) * 0.2
        children[i, 8] = side
        if i + 1 < n_int:
            children[i, 5] = i + 1
    last, _ = nodes[-1]
    for k in range(8):
        off = np.array([(k >> 2) & 1, (k >> 1) & 1, k & 1])
        centres[n_int + k] = (last + off + 0.5) * 0.2
        children[n_int + k, 8] = 1
        children[n_int - 1, k] = n_int + k
    return centres, children, (last + 1.0) * 0.2


@pytest.mark.parametrize('levels', [4, 10, 11, 13])
def test_deep_chain_tree_falls_back_to_the_depth_first_walk(levels):
    """levels <= 10: the level-synchronous walk; deeper: its path key is full
    and the depth-first walk answers — same hits either way"""
    from xrdslam_amd.compat import grid
    centres, childs, target = _chain_tree(levels)
    rng = np.random.default_rng(levels)
    M = 200
    o = (target + np.array([0.0, 0.0, -3.0]) +
         rng.uniform(-0.5, 0.5, (M, 3))).astype(np.float32)
    d = (target + rng.uniform(-0.3, 0.3, (M, 3)) - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ridx, rmn, rmx, _ = svo_intersect_oracle(o[None], d[None], centres[None],
                                             childs[None], 0.2, 50)
    assert (ridx >= 0).any(-1).mean() > 0.3
    c = lambda a: torch.from_numpy(a).cuda()
    idx, mn, mx = grid.svo_intersect(c(o[None]), c(d[None]), c(centres[None]),
                                     c(childs[None]), 0.2, 50)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    hit = ridx >= 0
    assert np.allclose(mn.cpu().numpy()[hit], rmn[hit], rtol=1e-6, atol=0)
    assert np.allclose(mx.cpu().numpy()[hit], rmx[hit], rtol=1e-6, atol=0)


@pytest.mark.parametrize('n_max', [7, 50, 64, 200])
def test_dense_block_many_leaves(n_max):
    """a solid 52^3 block: rays cross 40-100+ leaves — more than one lane
    each, more than n_max (the cut keeps the reference's FIRST n_max in ITS
    stack order) and, along the diagonal, more than the 128 the
    level-synchronous walk collects (depth-first fallback)"""
    from xrdslam_amd.compat import grid, svo
    g = np.arange(40, 92, dtype=np.int32)
    vox = np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(-1, 3)
    svo.reset_id_counter()
    tree = svo.Octree()
    tree.init(256, 16, 0.2)
    tree.insert(torch.from_numpy(vox))
    voxels, children, _ = tree.get_centres_and_children()
    centres = ((voxels[:, :3] + voxels[:, -1:] / 2) * 0.2).numpy() \
        .astype(np.float32)
    childs = torch.cat([children, voxels[:, -1:]], -1).int().numpy()
    rng = np.random.default_rng(n_max)
    M = 256
    o = (np.array([[7.0, 7.0, 6.0]]) + rng.uniform(-0.5, 0.5, (M, 3))) \
        .astype(np.float32)
    d = rng.uniform(0.2, 1.0, (M, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ridx, rmn, rmx, _ = svo_intersect_oracle(o[None], d[None], centres[None],
                                             childs[None], 0.2, n_max)
    if n_max == 200:
        assert (ridx >= 0).sum(-1).max() > 128
    c = lambda a: torch.from_numpy(a).cuda()
    idx, mn, mx = grid.svo_intersect(c(o[None]), c(d[None]), c(centres[None]),
                                     c(childs[None]), 0.2, n_max)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    hit = ridx >= 0
    assert np.allclose(mn.cpu().numpy()[hit], rmn[hit], rtol=1e-6, atol=0)
    assert np.allclose(mx.cpu().numpy()[hit], rmx[hit], rtol=1e-6, atol=0)


def test_matches_reference_vectors():
    from xrdslam_amd.compat import grid
    g = np.load(GOLD)
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    idx, mn, mx = grid.svo_intersect(
        c(g['ray_start']), c(g['ray_dir']), c(g['points1'][None]),
        c(g['children1'][None]), float(g['voxelsize']), int(g['n_max']))
    assert np.array_equal(idx.cpu().numpy(), g['idx'])
    hit = g['idx'] >= 0
    assert np.allclose(mn.cpu().numpy()[hit], g['min_depth'][hit], rtol=1e-6,
                       atol=0)
    assert np.allclose(mx.cpu().numpy()[hit], g['max_depth'][hit], rtol=1e-6,
                       atol=0)
    args = [g['s_' + k] for k in ('pts_idx', 'min_depth', 'max_depth',
                                  'noise', 'probs', 'steps')]
    sidx, sdep, sdis = grid.inverse_cdf_sampling(*[c(a) for a in args], 0.0)
    assert np.array_equal(sidx.cpu().numpy(), g['s_idx'])
    assert np.array_equal(sdep.cpu().numpy(), g['s_depth'])
    assert np.array_equal(sdis.cpu().numpy(), g['s_dists'])


@pytest.mark.parametrize('seed,G,det,fixed', [(0, 1, False, 0.0),
                                              (1, 4, False, 0.0),
                                              (2, 200, False, 0.0),
                                              (3, 3, True, 0.0),
                                              (5, 2, False, 0.004),
                                              (6, 200, False, 0.0)])
def test_inverse_cdf_sampling_bit_exact(seed, G, det, fixed):
    """the wave-cooperative sampler against the serial reference semantics on
    the reference wrapper's [G, ceil(N/G), P] geometry, including the rows
    where the trailing loop's quirks fire and rays of zero chord length"""
    from xrdslam_amd.compat import grid
    n_rays = 5000 if seed == 6 else None
    args = sampler_case(seed, G, n_rays=n_rays, deterministic=det)
    want = (inverse_cdf_ref if ref_lib() is not None
            else inverse_cdf_oracle)(*args, fixed)
    c = lambda a: torch.from_numpy(a).cuda()
    got = grid.inverse_cdf_sampling(*[c(a) for a in args], fixed)
    for a, b, name in zip(got, want, ('idx', 'depth', 'dists')):
        assert np.array_equal(a.cpu().numpy(), b), name
    assert (want[0] >= 0).sum() > 5 * args[0].shape[0] * args[0].shape[1]


def test_grid_shim_checks_inputs_and_dead_functions():
    from xrdslam_amd.compat import grid
    a = torch.zeros(1, 4, 3)
    with pytest.raises(RuntimeError):
        grid.svo_intersect(a, a, a, torch.zeros(1, 4, 9, dtype=torch.int32),
                           0.2, 5)  # CPU tensors
    with pytest.raises(NotImplementedError):
        grid.ball_intersect()
