"""GPU parity of the Gaussian rasteriser (diff_gaussian_rasterization shim ->
HIP kernels) against the torch oracle (oracle/gs_oracle.py; parity unpinned by
the reference: the CUDA dependency is not vendored).  1e-4 relative on the
rendered colour/depth and on every gradient incl. means2D."""
import numpy as np
import pytest
import torch

import gs_oracle as go
from nice_golden_util import rel_err

pytestmark = pytest.mark.gpu


def scene(N, H, W, seed, fx=40.0):
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(N, 3, generator=g) * torch.tensor([0.8, 0.6, 0.5]) + \
        torch.tensor([0.0, 0.0, 2.5])
    means[0] = torch.tensor([0.0, 0.0, 0.1])      # behind the near cull
    means[1] = torch.tensor([9.0, 0.0, 2.0])      # far outside the frustum
    cols = torch.rand(N, 3, generator=g)
    op = torch.rand(N, 1, generator=g) * 0.9 + 0.05
    sc = torch.rand(N, 3, generator=g) * 0.15 + 0.02
    rot = torch.nn.functional.normalize(torch.randn(N, 4, generator=g))
    ang = 0.2
    w2c = torch.tensor([[np.cos(ang), 0, np.sin(ang), 0.05],
                        [0, 1, 0, -0.02], [-np.sin(ang), 0, np.cos(ang), 0.1],
                        [0, 0, 0, 1]], dtype=torch.float32)
    near, far = 0.01, 100.0
    cx, cy = (W - 1) / 2 + 0.3, (H - 1) / 2 - 0.2
    proj = torch.tensor([[2 * fx / W, 0, -(W - 2 * cx) / W, 0],
                         [0, 2 * fx / H, -(H - 2 * cy) / H, 0],
                         [0, 0, far / (far - near), -(far * near) / (far - near)],
                         [0, 0, 1, 0]])
    view = w2c.t().contiguous()
    full = (proj @ w2c).t().contiguous()
    return means, cols, op, sc, rot, view, full, W / (2 * fx), H / (2 * fx)


@pytest.mark.parametrize('N,H,W,iso', [(60, 32, 48, False), (300, 40, 56, True),
                                       (5, 16, 16, False)])
def test_rasterizer_matches_oracle(N, H, W, iso):
    from xrdslam_amd.compat import diff_gaussian_rasterization as dgr
    means, cols, op, sc, rot, view, full, tfx, tfy = scene(N, H, W, N)
    if iso:  # SplaTAM: isotropic scales, identity rotations
        sc = sc[:, :1].repeat(1, 3)
        rot = torch.tensor([[1.0, 0, 0, 0]]).repeat(N, 1)
    wc = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1))
    leaves = [t.clone().requires_grad_(True) for t in (means, cols, op, sc, rot)]
    C_ref, radii_ref, D_ref, ndc = go.rasterize(*leaves, view, full, H, W, tfx,
                                                tfy)
    (C_ref * wc).sum().backward()
    dev = torch.device('cuda:0')
    gl = [t.clone().to(dev).requires_grad_(True)
          for t in (means, cols, op, sc, rot)]
    m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
    rs = dgr.GaussianRasterizationSettings(
        H, W, tfx, tfy, torch.zeros(3, device=dev), 1.0, view.to(dev).unsqueeze(0),
        full.to(dev).unsqueeze(0), 0, torch.zeros(3, device=dev), False)
    color, radii, depth = dgr.GaussianRasterizer(rs)(
        means3D=gl[0], means2D=m2d, opacities=gl[2], colors_precomp=gl[1],
        scales=gl[3], rotations=gl[4])
    (color * wc.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert color.shape == (3, H, W) and depth.shape == (1, H, W)
    assert torch.equal(radii.cpu(), radii_ref)
    assert rel_err(color.detach().cpu(), C_ref.detach()) < 1e-4
    assert rel_err(depth.detach().cpu(), D_ref.detach()) < 1e-4
    names = ['means3D', 'colors', 'opacities', 'scales', 'rotations']
    for name, a, b in zip(names, gl, leaves):
        if b.grad.abs().max() < 1e-12:  # isotropic: rotation grad is exactly 0
            assert a.grad.abs().max() < 1e-5, name
        else:
            assert rel_err(a.grad.cpu(), b.grad) < 1e-4, name
    assert rel_err(m2d.grad[:, :2].cpu(), ndc.grad) < 1e-4
    assert m2d.grad[:, 2].abs().max() == 0


def _big_scene(N, seed, aniso_every=0):
    """640x480 SplaTAM-shaped view (BASELINE configs[3]): N isotropic
    Gaussians of a few pixels radius spread over the frustum in depth 1-4 m;
    ``aniso_every`` = k > 0 turns every k-th Gaussian into an anisotropic one
    with a random orientation (the rotation gradient of an isotropic Gaussian
    is identically zero)"""
    H, W, fx = 480, 640, 320.0
    g = torch.Generator().manual_seed(seed)
    z = 1.0 + 3.0 * torch.rand(N, generator=g)
    u = (torch.rand(N, generator=g) * 1.1 - 0.05) * W
    v = (torch.rand(N, generator=g) * 1.1 - 0.05) * H
    means = torch.stack([(u - 319.5) / fx * z, (v - 239.5) / fx * z, z], -1)
    cols = torch.rand(N, 3, generator=g)
    op = torch.rand(N, 1, generator=g) * 0.9 + 0.05
    sc = (0.004 + 0.012 * torch.rand(N, 1, generator=g)).repeat(1, 3) * \
        z[:, None] / 2.0
    rot = torch.tensor([[1.0, 0, 0, 0]]).repeat(N, 1)
    if aniso_every:
        k = aniso_every
        sc[::k] = sc[::k] * (0.5 + 1.5 * torch.rand(sc[::k].shape,
                                                   generator=g))
        q = torch.randn(rot[::k].shape, generator=g)
        rot[::k] = q / q.norm(dim=-1, keepdim=True)
    w2c = torch.eye(4)
    near, far = 0.01, 100.0
    cx, cy = 319.5, 239.5
    proj = torch.tensor([[2 * fx / W, 0, -(W - 2 * cx) / W, 0],
                         [0, 2 * fx / H, -(H - 2 * cy) / H, 0],
                         [0, 0, far / (far - near), -(far * near) / (far - near)],
                         [0, 0, 1, 0]])
    return (means, cols, op, sc, rot, w2c.t().contiguous(),
            (proj @ w2c).t().contiguous(), W / (2 * fx), H / (2 * fx), H, W)


def test_rasterizer_at_baseline_shape_vs_oracle_crops():
    """640x480, 120 000 Gaussians (BASELINE configs[3]: SplaTAM renders the
    whole image of ~4e5 Gaussians per pass; the dense oracle is O(N H W), so
    it evaluates six 32x32 crops): colour, depth and — with a loss that weights
    only the crops' pixels — every gradient, the rotations' included (a third
    of the Gaussians are anisotropic with random orientations so that it is
    not identically zero).  A fifth of the Gaussians have opacity 0.999, so
    alpha saturates at 0.99 near their centres: the published backward passes
    the gradient through that clamp."""
    from xrdslam_amd.compat import diff_gaussian_rasterization as dgr
    N = 120000
    means, cols, op, sc, rot, view, full, tfx, tfy, H, W = _big_scene(
        N, 3, aniso_every=3)
    op[::5] = 0.999
    crops = [(0, 0), (304, 224), (608, 448), (96, 400), (512, 64), (320, 16)]
    gw = torch.Generator().manual_seed(9)
    wc = torch.zeros(3, H, W)
    for x0, y0 in crops:
        wc[:, y0:y0 + 32, x0:x0 + 32] = torch.rand(3, 32, 32, generator=gw)
    dev = torch.device('cuda:0')
    gl = [t.clone().to(dev).requires_grad_(True)
          for t in (means, cols, op, sc, rot)]
    m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
    rs = dgr.GaussianRasterizationSettings(
        H, W, tfx, tfy, torch.zeros(3, device=dev), 1.0,
        view.to(dev).unsqueeze(0), full.to(dev).unsqueeze(0), 0,
        torch.zeros(3, device=dev), False)
    color, radii, depth = dgr.GaussianRasterizer(rs)(
        means3D=gl[0], means2D=m2d, opacities=gl[2], colors_precomp=gl[1],
        scales=gl[3], rotations=gl[4])
    (color * wc.to(dev)).sum().backward()
    torch.cuda.synchronize()
    leaves = [t.clone().requires_grad_(True) for t in (means, cols, op, sc, rot)]
    ndc_grad = torch.zeros(N, 2)
    saturated = 0
    for x0, y0 in crops:
        C_ref, radii_ref, D_ref, ndc = go.rasterize(
            *leaves, view, full, H, W, tfx, tfy, window=(x0, y0, 32, 32))
        (C_ref * wc[:, y0:y0 + 32, x0:x0 + 32]).sum().backward()
        ndc_grad += ndc.grad
        got_c = color[:, y0:y0 + 32, x0:x0 + 32].detach().cpu()
        got_d = depth[:, y0:y0 + 32, x0:x0 + 32].detach().cpu()
        assert rel_err(got_c, C_ref.detach()) < 1e-4, (x0, y0)
        assert rel_err(got_d, D_ref.detach()) < 1e-4, (x0, y0)
    assert torch.equal(radii.cpu(), radii_ref)
    names = ['means3D', 'colors', 'opacities', 'scales', 'rotations']
    assert len(names) == len(gl) == len(leaves)
    for name, a, b in zip(names, gl, leaves):
        assert float(b.grad.abs().max()) > 0, name
        assert rel_err(a.grad.cpu(), b.grad) < 1e-4, name
    assert rel_err(m2d.grad[:, :2].cpu(), ndc_grad) < 1e-4
    # the saturated branch was exercised: some opacity-0.999 Gaussian centre
    # lies inside a crop and received gradient
    hit = (leaves[2].grad[::5].abs() > 0).sum()
    assert int(hit) > 10, int(hit)


def test_rasterizer_empty_and_unsupported():
    from xrdslam_amd.compat import diff_gaussian_rasterization as dgr
    dev = torch.device('cuda:0')
    means, cols, op, sc, rot, view, full, tfx, tfy = scene(4, 16, 16, 0)
    means[:, 2] = -1.0  # everything behind the camera
    rs = dgr.GaussianRasterizationSettings(
        16, 16, tfx, tfy, torch.tensor([0.2, 0.3, 0.4], device=dev), 1.0,
        view.to(dev), full.to(dev), 0, torch.zeros(3, device=dev), False)
    color, radii, depth = dgr.GaussianRasterizer(rs)(
        means3D=means.to(dev), means2D=torch.zeros(4, 3, device=dev),
        opacities=op.to(dev), colors_precomp=cols.to(dev), scales=sc.to(dev),
        rotations=rot.to(dev))
    assert (radii == 0).all() and (depth == 0).all()
    assert torch.allclose(color[:, 3, 3].cpu(), torch.tensor([0.2, 0.3, 0.4]))
    with pytest.raises(NotImplementedError):
        dgr.GaussianRasterizer(rs)(means3D=means.to(dev), means2D=None,
                                   opacities=op.to(dev), shs=cols.to(dev))


@pytest.mark.parametrize('N,H,W,cap_factor', [(300, 40, 56, 1.5),
                                              (5000, 120, 160, 1.0),
                                              (5000, 120, 160, 0.5),
                                              # > 4096 entries a tile: the
                                              # global rank-sort path
                                              (30000, 32, 48, 1.0)])
def test_device_binning_equals_scan_sort_on_the_host_side(N, H, W, cap_factor):
    """xrd_gs_bin (scan + static-capacity key list + radix sort + ranges, no
    host sync) against the phase-by-phase path it replaced (torch.cumsum,
    xrd_gs_duplicate_keys, stable torch.sort, xrd_gs_tile_ranges): identical
    per-tile lists when the capacity holds the pass, the true count reported
    when it does not"""
    import ctypes as C

    from xrdslam_amd import _lib
    from xrdslam_amd.compat import diff_gaussian_rasterization as dgr
    lib = _lib.lib()
    dev = torch.device('cuda:0')
    means, cols, op, sc, rot, view, full, tfx, tfy = scene(N, H, W, 7)
    rs = dgr.GaussianRasterizationSettings(
        H, W, tfx, tfy, torch.zeros(3, device=dev), 1.0,
        view.to(dev).unsqueeze(0), full.to(dev).unsqueeze(0), 0,
        torch.zeros(3, device=dev), False)
    cam = dgr._camera(rs)
    st = _lib.stream_ptr(dev)
    f = dict(dtype=torch.float32, device=dev)
    i = dict(dtype=torch.int32, device=dev)
    depths, xy = torch.zeros(N, **f), torch.zeros(N, 2, **f)
    conic = torch.zeros(N, 4, **f)
    radii, rect, tiles = torch.zeros(N, **i), torch.zeros(N, 4, **i), \
        torch.zeros(N, **i)
    _lib.check(lib.xrd_gs_preprocess(
        C.byref(cam), N, _lib.ptr(means.to(dev)), _lib.ptr(sc.to(dev)),
        _lib.ptr(rot.to(dev)), _lib.ptr(op.to(dev)), _lib.ptr(depths),
        _lib.ptr(xy), _lib.ptr(conic), _lib.ptr(radii), _lib.ptr(rect),
        _lib.ptr(tiles), st), 'preprocess')
    offsets = torch.cumsum(tiles.long(), 0)
    total = int(offsets[-1])
    assert total > 0
    keys = torch.empty(total, dtype=torch.int64, device=dev)
    vals = torch.empty(total, **i)
    _lib.check(lib.xrd_gs_duplicate_keys(
        N, W, _lib.ptr(rect), _lib.ptr(offsets), _lib.ptr(depths),
        _lib.ptr(keys), _lib.ptr(vals), st), 'dup')
    keys, order = torch.sort(keys, stable=True)
    ref_list = vals[order]
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    ref_ranges = torch.zeros(nt, 2, **i)
    _lib.check(lib.xrd_gs_tile_ranges(total, _lib.ptr(keys),
                                      _lib.ptr(ref_ranges), st), 'ranges')
    cap = max(int(total * cap_factor), 1)
    plist = torch.empty(cap, **i)
    ranges = torch.empty(nt, 2, **i)
    n_keys = torch.empty(1, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.xrd_gs_bin_ws_bytes(N, cap, W, H), dtype=torch.uint8,
                     device=dev)
    _lib.check(lib.xrd_gs_bin(
        N, W, H, _lib.ptr(rect), _lib.ptr(tiles), _lib.ptr(depths), cap,
        _lib.ptr(ws), _lib.ptr(plist), _lib.ptr(ranges), _lib.ptr(n_keys), st),
        'bin')
    torch.cuda.synchronize()
    assert int(n_keys) == total
    if cap >= total:
        assert torch.equal(ranges, ref_ranges)
        assert torch.equal(plist[:total], ref_list)
    else:
        # an under-sized list keeps well-formed ranges inside the capacity
        r = ranges.cpu().numpy()
        assert (r[:, 1] >= r[:, 0]).all() and r.max() <= cap


@pytest.mark.parametrize('N,H,W', [(300, 40, 56), (4000, 120, 160)])
def test_dual_pass_equals_two_single_passes(N, H, W):
    """rasterize_dual (one preprocess / binning / blend each way with two
    colour sets) against two GaussianRasterizer calls over the same Gaussians:
    both images, the depth, and every gradient (the means2D gradient is the
    sum of the two calls')"""
    from xrdslam_amd.compat import diff_gaussian_rasterization as dgr
    means, cols, op, sc, rot, view, full, tfx, tfy = scene(N, H, W, 3)
    g = torch.Generator().manual_seed(9)
    cols_b = torch.rand(N, 3, generator=g) * torch.tensor([3.0, 1.0, 9.0])
    wa = torch.rand(3, H, W, generator=g)
    wb = torch.rand(3, H, W, generator=g)
    dev = torch.device('cuda:0')
    rs = dgr.GaussianRasterizationSettings(
        H, W, tfx, tfy, torch.zeros(3, device=dev), 1.0,
        view.to(dev).unsqueeze(0), full.to(dev).unsqueeze(0), 0,
        torch.zeros(3, device=dev), False)

    def leaves():
        ts = [t.clone().to(dev).requires_grad_(True)
              for t in (means, cols, cols_b, op, sc, rot)]
        return ts, torch.zeros(N, 3, device=dev, requires_grad=True)

    (m, ca, cb, o, s, r), m2 = leaves()
    A, radii, depth = dgr.GaussianRasterizer(rs)(
        means3D=m, means2D=m2, opacities=o, colors_precomp=ca, scales=s,
        rotations=r)
    B, _, _ = dgr.GaussianRasterizer(rs)(
        means3D=m, means2D=m2, opacities=o, colors_precomp=cb, scales=s,
        rotations=r)
    ((A * wa.to(dev)).sum() + (B * wb.to(dev)).sum()).backward()
    ref = [A, B, depth] + [t.grad for t in (m, ca, cb, o, s, r, m2)]
    (m_, ca_, cb_, o_, s_, r_), m2_ = leaves()
    A2, radii2, depth2, B2 = dgr.rasterize_dual(rs, m_, m2_, o_, ca_, cb_, s_,
                                                r_)
    ((A2 * wa.to(dev)).sum() + (B2 * wb.to(dev)).sum()).backward()
    got = [A2, B2, depth2] + [t.grad for t in (m_, ca_, cb_, o_, s_, r_, m2_)]
    assert torch.equal(radii, radii2)
    names = ['img_a', 'img_b', 'depth', 'means3D', 'colors_a', 'colors_b',
             'opacities', 'scales', 'rotations', 'means2D']
    for name, a, b in zip(names, got, ref):
        if b.abs().max() < 1e-12:
            assert a.abs().max() < 1e-6, name
        else:
            assert rel_err(a.detach().cpu(), b.detach().cpu()) < 1e-4, name
