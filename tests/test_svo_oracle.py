"""CPU sanity of the C oracle for svo_intersect / inverse_cdf_sampling
(oracle/svo_oracle.c): hits are leaves, come out in DFS order with valid
intervals, and the sampler reproduces a hand-computed case."""
import numpy as np

from svo_util import inverse_cdf_oracle, make_tree, svo_intersect_oracle


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
    # padding is a suffix
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
