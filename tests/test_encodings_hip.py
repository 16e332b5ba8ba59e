"""GPU parity of the hash-grid / OneBlob kernels (tinycudann shim) against the
CPU oracle (oracle/tcnn_oracle.py; parity unpinned by the reference, so also
finite-difference and invariant checks).  Tolerance 1e-4 relative."""
import numpy as np
import pytest
import torch

import tcnn_oracle as to
from nice_golden_util import rel_err

pytestmark = pytest.mark.gpu


def _enc(cfg, dims=3):
    from xrdslam_amd.compat import tinycudann as tcnn
    return tcnn.Encoding(dims, cfg).cuda()


@pytest.mark.parametrize('n', [1, 17, 5000])
def test_hashgrid_matches_oracle_fwd_bwd(n):
    pls = float(np.exp2(np.log2(325 / 16) / 15))
    enc = _enc({'otype': 'HashGrid', 'n_levels': 16,
                'n_features_per_level': 2, 'log2_hashmap_size': 16,
                'base_resolution': 16, 'per_level_scale': pls})
    assert enc.n_output_dims == 32
    with torch.no_grad():  # make the table non-trivial
        enc.params.copy_(torch.randn_like(enc.params) * 0.1)
    # the oracle evaluates the SAME level table as the kernel (see
    # tests/test_tcnn_oracle.py on the rounding sensitivity of the top level)
    levels = [(float(a), int(b), int(c), int(d)) for a, b, c, d in
              zip(enc._scales, enc._res, enc._sizes, enc._offsets)]
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, 3, generator=g)
    x[0] = torch.tensor([0.0, 1.0, 0.5])  # domain edges
    w = torch.randn(n, 32, generator=g)
    xr = x.clone().requires_grad_(True)
    pr = enc.params.detach().cpu().clone().requires_grad_(True)
    yr = to.hashgrid_forward(xr, pr, levels)
    (yr * w).sum().backward()
    xg = x.cuda().requires_grad_(True)
    yg = enc(xg)
    (yg * w.cuda()).sum().backward()
    assert rel_err(yg.detach().cpu(), yr.detach()) < 1e-4
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-4
    assert rel_err(enc.params.grad.cpu(), pr.grad) < 1e-4


def test_hashgrid_accepts_float64_inputs_like_the_reference():
    enc = _enc({'otype': 'HashGrid', 'n_levels': 4, 'n_features_per_level': 2,
                'log2_hashmap_size': 10, 'base_resolution': 4,
                'per_level_scale': 1.5})
    x = torch.rand(64, 3, dtype=torch.float64, device='cuda',
                   requires_grad=True)
    y = enc(x)
    y.sum().backward()
    assert y.dtype == torch.float32 and x.grad.dtype == torch.float64


def test_oneblob_matches_oracle_and_sums_to_one():
    enc = _enc({'otype': 'OneBlob', 'n_bins': 16})
    assert enc.n_output_dims == 48
    g = torch.Generator().manual_seed(0)
    x = torch.rand(3000, 3, generator=g)
    x[0] = torch.tensor([0.0, 1.0, 0.03])
    w = torch.randn(3000, 48, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = to.oneblob_forward(xr, 16)
    (yr * w).sum().backward()
    xg = x.cuda().requires_grad_(True)
    yg = enc(xg)
    (yg * w.cuda()).sum().backward()
    assert rel_err(yg.detach().cpu(), yr.detach()) < 1e-4
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-4
    assert torch.allclose(yg.detach().reshape(3000, 3, 16).sum(-1),
                          torch.ones(3000, 3, device='cuda'), atol=1e-5)


def test_unsupported_otype_is_loud():
    from xrdslam_amd.compat import tinycudann as tcnn
    with pytest.raises(NotImplementedError):
        tcnn.Encoding(3, {'otype': 'SphericalHarmonics', 'degree': 4})


def test_table_gradient_run_merging_is_order_invariant():
    """the table-gradient scatter merges consecutive points that share a cell
    in registers (csrc/encodings.hip, round 4): the same point set in ray
    order (long runs per cell, rays that start left of the unit cube — cell
    coordinate -1 wraps like in the forward — and leave it on the right) and
    in a random order (no two neighbours share a cell) must give the same
    table gradient up to the order of the float additions"""
    import ctypes as C

    from xrdslam_amd import _lib
    pls = float(np.exp2(np.log2(325 / 16) / 15))
    enc = _enc({'otype': 'HashGrid', 'n_levels': 16,
                'n_features_per_level': 2, 'log2_hashmap_size': 16,
                'base_resolution': 16, 'per_level_scale': pls})
    g = torch.Generator().manual_seed(5)
    rays, S = 1800, 43                      # >= 65536 points: the long-run path
    o = torch.rand(rays, 1, 3, generator=g) * 0.9
    o[:40, 0, 0] = -0.04                    # cell (-1, 0, 0) at the coarse levels
    o[:40, 0, 1:] = 0.0
    d = torch.nn.functional.normalize(torch.rand(rays, 1, 3, generator=g), dim=-1)
    t = torch.linspace(0, 0.25, S).reshape(1, S, 1)
    x = (o + d * t).reshape(-1, 3).contiguous()
    assert float(x.min()) < -0.03 and float(x.max()) > 1.0
    dy = torch.randn(x.shape[0], 32, generator=g)
    dy[::7] = 0.0                           # skipped points inside runs
    lib, dev = _lib.lib(), torch.device('cuda:0')

    def table_grad(xs, dys):
        xs, dys = xs.to(dev).contiguous(), dys.to(dev).contiguous()
        out = torch.zeros_like(enc.params)
        _lib.check(lib.xrd_hashgrid_bwd(
            enc.n_levels, enc._scales.ctypes.data, enc._res.ctypes.data,
            enc._sizes.ctypes.data, enc._offsets.ctypes.data, xs.shape[0],
            _lib.ptr(xs), _lib.ptr(enc.params.detach()), _lib.ptr(dys),
            _lib.ptr(out), None, _lib.stream_ptr(dev)), 'xrd_hashgrid_bwd')
        return out.cpu()
    a = table_grad(x, dy)
    perm = torch.randperm(x.shape[0], generator=g)
    b = table_grad(x[perm], dy[perm])
    assert float(a.abs().max()) > 0
    assert rel_err(a, b) < 2e-5
    # the short-run path (fewer points) against the same subset, permuted
    sub = slice(0, 300 * S)
    c = table_grad(x[sub], dy[sub])
    p2 = torch.randperm(300 * S, generator=g)
    assert rel_err(c, table_grad(x[sub][p2], dy[sub][p2])) < 2e-5
