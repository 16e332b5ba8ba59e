"""CPU: the tile-culled rasteriser stand-in (oracle/gs_tiled.py, what the
reference's SplaTAM loop runs on for tests/golden/c1_splatam.npz) equals the
dense oracle (oracle/gs_oracle.py) — image, depth, radii and the gradient of
every input — on scenes with overlapping, saturated, culled and off-screen
Gaussians, at image sizes that are and are not multiples of the 16-pixel tile."""
import math
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
import gs_oracle  # noqa: E402
import gs_tiled  # noqa: E402


def scene(n, H, W, seed, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    fx = fy = 0.5 * W
    means = torch.stack([(torch.rand(n, generator=g) - 0.5) * 3.0 * spread,
                         (torch.rand(n, generator=g) - 0.5) * 2.4 * spread,
                         0.1 + torch.rand(n, generator=g) * 3.0], 1)
    colors = torch.rand(n, 3, generator=g)
    opac = torch.rand(n, 1, generator=g) * 1.2        # some saturate alpha
    scales = 0.01 + torch.rand(n, 3, generator=g) * 0.25
    rots = torch.randn(n, 4, generator=g)
    rots = rots / rots.norm(dim=1, keepdim=True)
    w2c = torch.eye(4)
    tanx, tany = W / (2 * fx), H / (2 * fy)
    P = torch.zeros(4, 4)
    P[0, 0], P[1, 1] = 1 / tanx, 1 / tany
    P[2, 2], P[2, 3], P[3, 2] = 100 / 99.99, -100 * 0.01 / 99.99, 1.0
    full = P @ w2c
    return (means, colors, opac, scales, rots, w2c.t().contiguous(),
            full.t().contiguous(), H, W, tanx, tany)


@pytest.mark.parametrize('n,H,W,seed', [(60, 32, 48, 0), (200, 40, 56, 1),
                                        (7, 16, 16, 2), (300, 48, 64, 3)])
def test_tiled_equals_dense(n, H, W, seed):
    args = scene(n, H, W, seed)
    bg = torch.tensor([0.1, 0.2, 0.3])
    outs = []
    for fn in (gs_oracle.rasterize, gs_tiled.rasterize):
        leaves = [a.clone().double().requires_grad_() for a in args[:5]]
        rest = [a.double() if torch.is_tensor(a) else a for a in args[5:]]
        kw = {'max_elems': 200_000} if fn is gs_tiled.rasterize else {}
        c, r, d, _ = fn(*leaves, *rest, bg=bg.double(), **kw)
        wc = torch.linspace(0.5, 1.5, c.numel(), dtype=torch.float64) \
            .reshape(c.shape)
        wd = torch.linspace(1.5, 0.5, d.numel(), dtype=torch.float64) \
            .reshape(d.shape)
        ((c * wc).sum() + (d * wd).sum()).backward()
        outs.append((c.detach(), r, d.detach(), [x.grad for x in leaves]))
    (c0, r0, d0, g0), (c1, r1, d1, g1) = outs
    assert torch.equal(r0, r1)
    assert float((c0 - c1).abs().max()) < 1e-10
    assert float((d0 - d1).abs().max()) < 1e-10
    assert float(c0.abs().max()) > 0.05          # something was drawn
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 1e-9 * max(
            1.0, float(a.abs().max()))


def test_empty_and_all_culled():
    args = list(scene(5, 32, 32, 4))
    args[0] = args[0].clone()
    args[0][:, 2] = -1.0                         # behind the camera
    c, r, d, _ = gs_tiled.rasterize(*args, bg=torch.tensor([0.3, 0.2, 0.1]))
    assert int(r.max()) == 0 and float(d.abs().max()) == 0
    assert torch.allclose(c, torch.tensor([0.3, 0.2, 0.1])[:, None, None]
                          .expand(3, 32, 32))
