"""CPU checks of the tiny-cuda-nn encoding oracle (parity unpinned by the
reference — self-consistency only) and of the host level-table function."""
import ctypes as C

import numpy as np
import torch

import tcnn_oracle as to
from xrdslam_amd import _lib

# office0 Co-SLAM: max extent 6.5 -> res 325
RES_TABLE = [16, 20, 24, 30, 36, 44, 54, 66, 80, 98, 120, 146, 178, 218, 266,
             325]


def coslam_pls():
    return float(np.exp2(np.log2(325 / 16) / 15))


def test_level_table_matches_survey_and_host_function():
    levels, total = to.hash_levels(16, 16, coslam_pls(), 16)
    assert [l[1] for l in levels] == RES_TABLE
    assert total == 820472  # entries (x2 floats = 6.56 MB)
    L = 16
    sc, rs = np.zeros(L, np.float32), np.zeros(L, np.uint32)
    sz, of = np.zeros(L, np.uint32), np.zeros(L, np.uint32)
    tot = C.c_uint32(0)
    assert _lib.lib().xrd_hashgrid_levels(
        L, 16, coslam_pls(), 16, 0, sc.ctypes.data, rs.ctypes.data,
        sz.ctypes.data, of.ctypes.data, C.byref(tot)) == 0
    # the top level sits exactly on an integer boundary (16*325/16 - 1 = 324):
    # glibc's exp2f/log2f (what tiny-cuda-nn's host code calls) lands on
    # 324.00003 -> res 326, numpy's float32 path on 323.9999 -> 325.  Hashed
    # levels do not use `res` for indexing, only `scale` (agrees to 1e-6).
    assert list(rs)[:15] == RES_TABLE[:15] and int(rs[15]) in (325, 326)
    assert tot.value == total
    assert np.allclose(sc, [l[0] for l in levels], rtol=1e-6)
    assert [int(s) for s in sz] == [l[2] for l in levels]
    assert [int(o) for o in of] == [l[3] for l in levels]
    # levels 0-4 dense, 5-15 hashed (65536 entries)
    assert all(int(s) == 65536 for s in sz[5:]) and int(sz[4]) == 46656


def test_oneblob_partition_of_unity_and_grad():
    x = torch.rand(200, 3, dtype=torch.float64).float()
    y = to.oneblob_forward(x, 16)
    assert y.shape == (200, 48)
    assert torch.allclose(y.reshape(200, 3, 16).sum(-1), torch.ones(200, 3),
                          atol=1e-5)
    assert (y >= -1e-6).all()


def test_hashgrid_oracle_interpolates_and_is_differentiable():
    levels, total = to.hash_levels(4, 4, 1.5, 8)
    params = (torch.rand(total * 2) - 0.5).requires_grad_(True)
    x = torch.rand(50, 3).requires_grad_(True)
    y = to.hashgrid_forward(x, params, levels)
    assert y.shape == (50, 8)
    y.square().sum().backward()
    assert params.grad.abs().sum() > 0 and x.grad.abs().sum() > 0
    # at an exact lattice point of a dense level the value is the table entry
    scale, res, n, off = levels[0]
    cell = torch.tensor([[1, 2, 0]])
    xp = (cell.float() + 0.5 - 0.5) / scale  # pos = x*scale + 0.5 -> cell+0.5?
    xq = (cell.float() - 0.5) / scale + 1e-7  # pos slightly above cell
    yq = to.hashgrid_forward(xq.clamp(min=0), params.detach(), levels)
    idx = cell[0, 0] + cell[0, 1] * res + cell[0, 2] * res * res
    if (xq >= 0).all():
        assert torch.allclose(yq[0, :2], params.detach().reshape(-1, 2)[off + idx],
                              atol=1e-4)
