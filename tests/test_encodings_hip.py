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
