"""GPU parity of the Vox-Fusion native ops (grid shim -> HIP kernels) against
the C oracle: voxel ids BIT-EXACT, depths within 1e-6 relative."""
import numpy as np
import pytest
import torch

from svo_util import inverse_cdf_oracle, make_tree, svo_intersect_oracle

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


def test_inverse_cdf_sampling_matches_oracle():
    """inputs built like voxel_helpers_voxfusion.py:647-714: sorted hits,
    probs = len/sum(len), steps = sum(len)/0.01, noise in (0.001,0.999)"""
    from xrdslam_amd.compat import grid
    centres, childs = make_tree(5)
    M = 600
    o, d = _rays(M, 3)
    idx, mn, mx, _ = svo_intersect_oracle(o[None], d[None], centres[None],
                                          childs[None], 0.2, 50)
    idx, mn, mx = idx[0], mn[0], mx[0]
    keep = (idx >= 0).any(1)
    idx, mn, mx = idx[keep], mn[keep], mx[keep]
    mn_s = np.where(idx >= 0, mn, 1e10).astype(np.float32)
    order = np.argsort(mn_s, 1, kind='stable')
    idx = np.take_along_axis(idx, order, 1)
    mn = np.take_along_axis(mn, order, 1)
    mx = np.take_along_axis(mx, order, 1)
    nh = int((idx >= 0).sum(1).max())
    idx, mn, mx = idx[:, :nh].copy(), mn[:, :nh].copy(), mx[:, :nh].copy()
    length = np.where(idx >= 0, mx - mn, 0).astype(np.float32)
    tot = length.sum(1, keepdims=True)
    probs = (length / tot).astype(np.float32)
    steps = (tot[:, 0] / 0.01).astype(np.float32)
    S = int(np.ceil(steps.max())) + nh
    rng = np.random.default_rng(0)
    noise = rng.uniform(0.001, 0.999, (idx.shape[0], S)).astype(np.float32)
    args = [a[None].copy() for a in (idx.astype(np.int32), mn, mx, noise,
                                     probs)] + [steps[None].copy()]
    rs_idx, rs_dep, rs_dis = inverse_cdf_oracle(*args, 0.0)
    c = lambda a: torch.from_numpy(a).cuda()
    g_idx, g_dep, g_dis = grid.inverse_cdf_sampling(*[c(a) for a in args], 0.0)
    assert np.array_equal(g_idx.cpu().numpy(), rs_idx)
    assert np.allclose(g_dep.cpu().numpy(), rs_dep, rtol=1e-6, atol=1e-7)
    assert np.allclose(g_dis.cpu().numpy(), rs_dis, rtol=1e-6, atol=1e-7)
    assert (rs_idx >= 0).sum() > 10 * idx.shape[0]  # the case is not trivial


def test_grid_shim_checks_inputs_and_dead_functions():
    from xrdslam_amd.compat import grid
    a = torch.zeros(1, 4, 3)
    with pytest.raises(RuntimeError):
        grid.svo_intersect(a, a, a, torch.zeros(1, 4, 9, dtype=torch.int32),
                           0.2, 5)  # CPU tensors
    with pytest.raises(NotImplementedError):
        grid.ball_intersect()
