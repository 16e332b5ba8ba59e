"""GPU parity tests of the fused NICE-SLAM render (HIP, through the C-ABI)
against (a) vectors produced by the reference's own modules
(tests/golden/nice_render.npz) and (b) the CPU oracle on fresh seeded inputs.

Tolerance (BASELINE.json north_star): 1e-4 relative (fp32) on rendered
depth/colour and on pose/map gradients (max-norm relative)."""
import os

import numpy as np
import pytest
import torch

import nice_oracle as no
from nice_golden_util import load_nice_golden, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _cuda():
    if not torch.cuda.is_available():
        pytest.fail('gpu test needs a GPU (run with -m "not gpu" elsewhere)')
    return torch.device('cuda:0')


def build_scene(bound, grids, decs, dev, color_requires_grad=False,
                grid_requires_grad=False):
    from xrdslam_amd.engine import nice as en
    scene = en.NiceScene(bound, device=dev)
    gl = {}
    for k, v in grids.items():
        g = en.to_channels_last_grid(v.to(dev))
        g.requires_grad_(grid_requires_grad)
        scene.set_grid(k, g)
        gl[k] = g
    flats = {}
    for kind, sd in decs.items():
        flat = en.flatten_state_dict(sd, kind).to(dev)
        if kind == 'color' and color_requires_grad:
            flat.requires_grad_(True)
        scene.set_decoder(kind, flat)
        flats[kind] = flat
    return scene, gl, flats


def test_mfma_lane_mapping():
    from xrdslam_amd import _lib
    dev = _cuda()
    a = torch.randn(16, 4, device=dev)
    b = torch.arange(64, device=dev, dtype=torch.float32).reshape(4, 16) * 0.1 \
        + torch.randn(4, 16, device=dev)
    out = torch.zeros(16, 16, device=dev)
    _lib.check(_lib.lib().xrd_selftest_mfma(_lib.ptr(a), _lib.ptr(b),
                                            _lib.ptr(out),
                                            _lib.stream_ptr(dev)))
    torch.cuda.synchronize()
    assert torch.allclose(out, a @ b, atol=1e-5)


@pytest.mark.parametrize('tag', ['coarse_map', 'middle_map', 'fine_map',
                                 'color_map', 'color_track'])
def test_render_matches_reference_golden(tag):
    from xrdslam_amd.engine import nice as en
    dev = _cuda()
    g, bound, grids, decs, (fx, fy, cx, cy, W, H) = load_nice_golden()
    stage, mode = tag.split('_')
    is_mapping = mode == 'map'
    scene, gl, flats = build_scene(bound, grids, decs, dev,
                                   color_requires_grad=True,
                                   grid_requires_grad=True)
    c2w = torch.from_numpy(g['c2w']).to(dev).requires_grad_(True)
    i = torch.from_numpy(g['i']).to(dev)
    j = torch.from_numpy(g['j']).to(dev)
    depth = torch.from_numpy(g['gt_depth']).to(dev)
    color = torch.from_numpy(g['gt_color']).to(dev)
    rays_o, rays_d = no.rays_from_uv(i, j, c2w, fx, fy, cx, cy)
    rays_o.retain_grad()
    rays_d.retain_grad()
    d, u, rgb = en.nice_render(scene, stage, rays_o, rays_d, depth)
    out = {'depth': d, 'uncertainty': u, 'rgb': rgb}
    assert d.dtype == torch.float64 and u.dtype == torch.float64
    assert rel_err(d.detach().cpu(), g[f'{tag}/depth']) < TOL
    assert rel_err(u.detach().cpu(), g[f'{tag}/uncertainty']) < TOL
    if stage == 'color':
        assert rel_err(rgb.detach().cpu(), g[f'{tag}/rgb']) < TOL
    ld = no.loss_dict(out, depth, color, is_mapping, stage)
    loss = sum(ld.values())
    assert rel_err(loss.detach().cpu(), g[f'{tag}/loss']) < TOL
    loss.backward()
    torch.cuda.synchronize()
    if stage != 'coarse':
        assert rel_err(rays_o.grad.cpu(), g[f'{tag}/g_rays_o']) < TOL
        assert rel_err(rays_d.grad.cpu(), g[f'{tag}/g_rays_d']) < TOL
        assert rel_err(c2w.grad.cpu(), g[f'{tag}/g_c2w']) < TOL
    for k, grid in gl.items():
        key = f'{tag}/g_{k}'
        if key in g and np.abs(g[key]).max() > 0:
            assert grid.grad is not None, k
            assert rel_err(grid.grad.cpu(), g[key]) < TOL, k
    if stage == 'color':
        gf = flats['color'].grad.cpu()
        off = 0
        for name, shape in en.param_shapes('color'):
            n = int(np.prod(shape))
            key = f'{tag}/g_dec_color/{name}'
            want = g[key] if key in g else np.zeros(shape, np.float32)
            got = gf[off:off + n].reshape(shape)
            scale = max(np.abs(want).max(), 1e-6)
            assert np.abs(got.numpy() - want).max() / scale < TOL, name
            off += n


def test_render_without_depth_matches_golden():
    from xrdslam_amd.engine import nice as en
    dev = _cuda()
    g, bound, grids, decs, (fx, fy, cx, cy, W, H) = load_nice_golden()
    scene, _, _ = build_scene(bound, grids, decs, dev)
    rays_o, rays_d = no.rays_from_uv(torch.from_numpy(g['i']).to(dev),
                                     torch.from_numpy(g['j']).to(dev),
                                     torch.from_numpy(g['c2w']).to(dev),
                                     fx, fy, cx, cy)
    with torch.no_grad():
        d, u, rgb = en.nice_render(scene, 'color', rays_o, rays_d, None)
    assert rel_err(d.cpu(), g['color_nodepth/depth']) < TOL
    assert rel_err(rgb.cpu(), g['color_nodepth/rgb']) < TOL
    assert rel_err(u.cpu(), g['color_nodepth/uncertainty']) < TOL


@pytest.mark.parametrize('n_rays', [1, 3, 1000])
def test_render_vs_oracle_fresh_inputs(n_rays):
    """seeded inputs at sizes the oracle finishes in seconds, incl. ragged ray
    counts (not a multiple of the rays-per-block) and rays leaving the bound"""
    from xrdslam_amd.engine import nice as en
    dev = _cuda()
    g, bound, grids, decs, cam = load_nice_golden()
    gen = torch.Generator().manual_seed(100 + n_rays)
    rays_o = (torch.rand(n_rays, 3, generator=gen) - 0.5) * 0.6
    rays_d = torch.randn(n_rays, 3, generator=gen)
    rays_d = rays_d / rays_d.norm(dim=1, keepdim=True) * (
        0.8 + 0.4 * torch.rand(n_rays, 1, generator=gen))
    depth = 0.3 + 1.5 * torch.rand(n_rays, 1, generator=gen)
    depth[torch.rand(n_rays, 1, generator=gen) < 0.15] = 0.0
    scene, _, _ = build_scene(bound, grids, decs, dev)
    for stage in ('middle', 'fine', 'color', 'coarse'):
        with torch.no_grad():
            ref = no.render_batch_ray(rays_o, rays_d, depth, grids, decs,
                                      bound, stage)
            d, u, rgb = en.nice_render(scene, stage, rays_o.to(dev),
                                       rays_d.to(dev), depth.to(dev))
        assert rel_err(d.cpu(), ref['depth']) < TOL, stage
        assert rel_err(u.cpu(), ref['uncertainty']) < TOL, stage
        if stage == 'color':
            assert rel_err(rgb.cpu(), ref['rgb']) < TOL, stage


OFFICE0_BOUND = [[-5.5, 6.0199995], [-6.7, 5.4599998], [-4.7, 5.5399998]]
OFFICE0_GRIDS = {'grid_coarse': (10, 12, 11), 'grid_middle': (31, 37, 35),
                 'grid_fine': (63, 75, 71), 'grid_color': (63, 75, 71)}


def _office0_case(seed):
    """NICE-SLAM at BASELINE configs[1]: office0 bound (input_config.py:66,
    rounded up to whole cells like conv_onet.py:67-75 does) with the grid
    sizes that follow from it: fine / colour 63x75x71 cells"""
    from xrdslam_amd.engine import nice as en
    g = torch.Generator().manual_seed(seed)
    bound = torch.tensor(OFFICE0_BOUND, dtype=torch.float64)
    grids = {k: torch.randn(1, 32, *s, generator=g) * 0.05
             for k, s in OFFICE0_GRIDS.items()}
    # hidden layers O(1); a small output layer keeps the occupancy logits
    # around +-0.1 so that 10*occ does not saturate the sigmoid and every
    # sample of a ray carries weight and gradient
    def scale(kind, n):
        if n == 'embedder._B':
            return 25.
        if n.startswith('output_linear') and kind != 'color':
            return 0.01
        return 0.2
    decs = {kind: {n: torch.randn(*s, generator=g) * scale(kind, n)
                   for n, s in en.param_shapes(kind)}
            for kind in ('coarse', 'middle', 'fine', 'color')}
    for kind in ('coarse', 'middle', 'fine'):
        # mostly free space along a ray: weights spread over many samples
        decs[kind]['output_linear.bias'] -= 0.12
    return bound, grids, decs


def _office0_rays(n, seed):
    g = torch.Generator().manual_seed(seed)
    rays_o = torch.tensor([0.2, -0.5, 0.3]) + \
        (torch.rand(n, 3, generator=g) - 0.5) * 1.5
    rays_d = torch.randn(n, 3, generator=g)
    rays_d = rays_d / rays_d.norm(dim=1, keepdim=True)
    depth = 0.8 + 3.5 * torch.rand(n, 1, generator=g)
    depth[torch.rand(n, 1, generator=g) < 0.08] = 0.0
    color = torch.rand(n, 3, generator=g)
    return rays_o, rays_d, depth, color


@pytest.mark.parametrize('tag,n_rays', [('color_track', 200),
                                        ('middle_map', 1000),
                                        ('fine_map', 1000),
                                        ('color_map', 1000),
                                        ('coarse_map', 1000)])
def test_render_at_baseline_config_vs_oracle(tag, n_rays):
    """the batch sizes of the reference loop (200 tracking rays, 1000 mapping
    rays) on office0-sized grids, forward and backward, against the CPU oracle
    (pinned to the reference by tests/test_oracle_nice.py): rendered
    depth / uncertainty / colour, loss, ray gradients, all four grid
    gradients and the colour-decoder gradient, 1e-4 in the max norm and
    element-wise (tests/parity.py)."""
    import parity
    from xrdslam_amd.engine import nice as en
    dev = _cuda()
    stage, mode = tag.split('_')
    is_mapping = mode == 'map'
    bound, grids, decs = _office0_case(1)
    rays_o, rays_d, depth, color = _office0_rays(n_rays, 2 + n_rays)
    # oracle
    og = {k: v.clone().requires_grad_(is_mapping) for k, v in grids.items()}
    od = {kind: {n: v.clone().requires_grad_(is_mapping and kind == 'color')
                 for n, v in sd.items()} for kind, sd in decs.items()}
    ro = rays_o.clone().requires_grad_(stage != 'coarse')
    rd = rays_d.clone().requires_grad_(stage != 'coarse')
    ref = no.render_batch_ray(ro, rd, depth, og, od, bound, stage)
    ref_loss = sum(no.loss_dict(ref, depth, color, is_mapping,
                                stage).values())
    ref_loss.backward()
    # HIP
    scene, gl, flats = build_scene(bound, grids, decs, dev,
                                   color_requires_grad=is_mapping,
                                   grid_requires_grad=is_mapping)
    go = rays_o.to(dev).requires_grad_(stage != 'coarse')
    gd = rays_d.to(dev).requires_grad_(stage != 'coarse')
    d, u, rgb = en.nice_render(scene, stage, go, gd, depth.to(dev))
    out = {'depth': d, 'uncertainty': u, 'rgb': rgb}
    loss = sum(no.loss_dict(out, depth.to(dev), color.to(dev), is_mapping,
                            stage).values())
    loss.backward()
    torch.cuda.synchronize()
    pairs = [('depth', d, ref['depth']), ('uncertainty', u,
                                          ref['uncertainty']),
             ('loss', loss, ref_loss)]
    if stage == 'color':
        pairs.append(('rgb', rgb, ref['rgb']))
    if stage != 'coarse' and is_mapping:
        pairs += [('g_rays_o', go.grad, ro.grad), ('g_rays_d', gd.grad,
                                                   rd.grad)]
    if not is_mapping:
        # Per-ray pose gradients.  All but isolated rays agree to ~1e-6; a ray
        # one of whose samples sits on a ReLU kink / a grid-cell border flips
        # that branch under a last-bit difference and its gradient jumps by
        # ~1e-3 of the largest one.  The float32 oracle shows the same rows
        # against a float64 evaluation of the same formulas (recorded below),
        # so: at most 1 % of the rays may exceed 1e-4, none 5e-3.
        with no.high_precision():
            tg = {k: v.double() for k, v in grids.items()}
            td = {kind: {n: v.double() for n, v in sd.items()}
                  for kind, sd in decs.items()}
            to = rays_o.double().requires_grad_(True)
            tdir = rays_d.double().requires_grad_(True)
            tr = no.render_batch_ray(to, tdir, depth, tg, td, bound, stage)
            tout = {'depth': tr['depth'], 'rgb': tr['rgb'],
                    'uncertainty': ref['uncertainty'].detach().double()}
            sum(no.loss_dict(tout, depth, color, is_mapping,
                             stage).values()).backward()
        for name, got, o32, t64 in (('g_rays_o', go.grad, ro.grad, to.grad),
                                    ('g_rays_d', gd.grad, rd.grad,
                                     tdir.grad)):
            parity.report(f'nice_office0/{tag}/{name}[f32 oracle vs f64]',
                          o32, t64)
            parity.report(f'nice_office0/{tag}/{name}[kernel vs f64]', got,
                          t64)
            # REFEREE form (as for Point-SLAM): the f64 evaluation of the
            # oracle judges both f32 evaluations, row by row (row = ray).
            # Measured (profiles/r05_parity_margins.txt, 200 rays): the MEDIAN
            # row of the kernel and of the f32 oracle sit at the same 7.0e-7
            # from f64, the 99th percentile at 9.4e-5 / 7.7e-5; 2 rows (1.0 %)
            # of the kernel and 1 row (0.5 %) of the oracle are further than
            # 1e-4: rays with a sample on a ReLU kink of a 32-wide decoder or
            # on a cell border, which flips under a last-bit difference of the
            # Fourier argument p.B (|p.B| ~ 1e2..1e3 rad: one f32 ulp of the
            # argument is up to 6e-5 rad) and moves that ray's gradient by
            # 1e-4..1e-3 of the largest gradient.  WHICH rays flip differs
            # between any two f32 evaluations, so round 4's "the kernel is 2 x
            # further from f64 than torch" (7.9e-4 vs 3.8e-4 in the max-norm)
            # is ONE ray's jump against another ray's jump, not a systematic
            # factor: every statistic that is not a single row agrees.
            # Bar: median and 90th-percentile row <= 1.5 x the oracle's (the
            # 99th percentile of 200 rows IS the second / third worst row),
            # share of rows beyond 1e-4 <= 1.5 x the oracle's + 2 rows, no
            # row beyond 5e-3 (the size of a kink jump).
            frac, worst = parity.row_outliers(got, t64)
            frac64, worst64 = parity.row_outliers(o32, t64)
            ks, os_ = parity.row_stats(got, t64), parity.row_stats(o32, t64)
            rep = os.environ.get('XRD_PARITY_REPORT')
            if rep:
                with open(rep, 'a') as f:
                    f.write(f'nice_office0/{tag}/{name} rows vs f64 (mean, '
                            f'median, p99, max): kernel {ks[0]:.2e} '
                            f'{ks[1]:.2e} {ks[2]:.2e} {ks[3]:.2e}; f32 oracle '
                            f'{os_[0]:.2e} {os_[1]:.2e} {os_[2]:.2e} '
                            f'{os_[3]:.2e}; rows > 1e-4: kernel {frac:.3%} '
                            f'oracle {frac64:.3%}\n')
            n_rows = got.shape[0]
            msg = (name, 'kernel', ks, frac, 'oracle', os_, frac64)
            assert ks[1] <= 1.5 * os_[1] + 1e-7, msg
            assert ks[4] <= max(1e-5, 1.5 * os_[4]), msg
            assert frac <= 1.5 * frac64 + 2.0 / n_rows, msg
            assert worst < 5e-3, msg
    if is_mapping:
        for k, grid in gl.items():
            want = og[k].grad
            if want is None or float(want.abs().max()) == 0.0:
                assert grid.grad is None or float(grid.grad.abs().max()) == 0
                continue
            pairs.append((f'g_{k}', grid.grad, want))
        if stage == 'color':
            gf = flats['color'].grad.cpu()
            off = 0
            for name, shape in en.param_shapes('color'):
                n = int(np.prod(shape))
                want = od['color'][name].grad
                got = gf[off:off + n].reshape(shape)
                off += n
                if want is None:  # fc_c/.. entries the stage does not reach
                    assert float(got.abs().max()) == 0.0, name
                    continue
                pairs.append((f'g_dec_color/{name}', got, want))
    parity.assert_all([(f'nice_office0/{tag}/{n}', a, b) for n, a, b in pairs])


@pytest.mark.parametrize('stage', ['fine', 'color'])
def test_fine_decoder_weight_gradient_from_the_export(stage):
    """mapping_fix_fine = False (conv_onet.py:62,190-195): the one-launch
    mapping iteration hands out the sample points and d loss / d occupancy
    logit of every sample (xrd_nice_map_iter_export); the fine decoder's weight
    gradient formed from them (engine/nice.decoder_weight_grad: the decoder
    re-evaluated with torch ops on the device) against the CPU oracle's
    autograd, every parameter of the fine decoder, at BASELINE configs[1]
    shapes (office0 grids, 1000 rays, 6 % masked); the exported points are the
    kernel's own sample positions (checked through the gradient) and the other
    outputs are those of the plain call."""
    import parity
    from xrdslam_amd.engine import nice as en
    dev = _cuda()
    n = 1000
    bound, grids, decs = _office0_case(1)
    rays_o, rays_d, depth, color = _office0_rays(n, 11)
    keep = torch.rand(n, generator=torch.Generator().manual_seed(4)) > 0.06
    od = {kind: {m: v.clone().requires_grad_(kind == 'fine')
                 for m, v in sd.items()} for kind, sd in decs.items()}
    ref = no.render_batch_ray(rays_o[keep], rays_d[keep], depth[keep], grids,
                              od, bound, stage)
    sum(no.loss_dict(ref, depth[keep], color[keep], True,
                     stage).values()).backward()
    want = en.flatten_state_dict(
        {m: v.grad for m, v in od['fine'].items()}, 'fine')
    scene, gl, flats = build_scene(bound, grids, decs, dev,
                                   grid_requires_grad=True)
    args = (scene, stage, rays_o.to(dev), rays_d.to(dev), depth.to(dev), None,
            color.to(dev), keep.to(dev).to(torch.uint8), 0.2, True, False)
    plain = en.nice_map_iter(*args)
    export = {}
    got = en.nice_map_iter(*args, export=export)
    torch.cuda.synchronize()
    assert abs(float(got[0]) - float(plain[0])) <= 1e-9 * abs(float(plain[0]))
    assert torch.equal(got[1], plain[1]) and torch.equal(got[2], plain[2])
    assert export['points'].shape == (n * 48, 3)
    # rays that were masked out carry no gradient
    g = export['g_occ'].reshape(n, 48)
    assert float(g[~keep.to(dev)].abs().max()) == 0.0
    g_fine = en.decoder_weight_grad(scene, 'fine', flats['fine'],
                                    export['points'], export['g_occ'])
    assert float(want.abs().max()) > 0
    parity.assert_all([(f'fine_dw/{stage}', g_fine.cpu(), want)])


@pytest.mark.parametrize('n,use_color,handle_dynamic,masked', [
    (200, True, True, False), (200, True, True, True), (333, False, True, False),
    (200, True, False, True), (1024, True, True, False)])
def test_one_launch_tracking_iteration(n, use_color, handle_dynamic, masked):
    """xrd_nice_track_iter (forward + robust tracking loss + backward as ONE
    launch: the batch median is taken at a grid barrier) against the CPU
    oracle (loss_dict's tracking branch, pinned to conv_onet.py:145-176 by
    tests/test_oracle_nice.py) AND against the three-launch chain it replaces
    (xrd_nice_render_fwd / xrd_nice_loss / xrd_nice_render_bwd), at
    BASELINE configs[1] (office0 grids, 200 tracking rays) and at the largest
    batch the barrier admits; twice in a row on the same workspace."""
    import parity
    from xrdslam_amd.engine import nice as en
    dev = _cuda()
    bound, grids, decs = _office0_case(1)
    rays_o, rays_d, depth, color = _office0_rays(n, 12)
    keep = None
    if masked:
        keep = (torch.rand(n, generator=torch.Generator().manual_seed(4))
                > 0.1)
    scene, gl, flats = build_scene(bound, grids, decs, dev)
    args = (scene, rays_o.to(dev), rays_d.to(dev), depth.to(dev), None,
            color.to(dev), None if keep is None else
            keep.to(dev).to(torch.uint8), use_color, handle_dynamic, 0.5)
    got = []
    for one in (True, True, False):
        en.TRACK_ONE_LAUNCH = one
        try:
            loss, g_o, g_d = en.nice_track_iter(*args)
        finally:
            en.TRACK_ONE_LAUNCH = False
        torch.cuda.synchronize()
        got.append((float(loss), g_o.cpu(), g_d.cpu()))
    (l1, o1, d1), (l2, o2, d2), (l3, o3, d3) = got
    # same workspace twice: identical (no atomics on this path)
    assert l1 == l2 and torch.equal(o1, o2) and torch.equal(d1, d2)
    # the chain it replaces: the same arithmetic
    assert abs(l1 - l3) <= 1e-6 * abs(l3)
    assert parity.rel_max(o1, o3) < 1e-5 and parity.rel_max(d1, d3) < 1e-5
    if n > 400 or not (use_color and handle_dynamic):
        # (the oracle leg: seconds per 100 rays on the CPU; its loss_dict is
        # the reference's default tracking loss — colour term and dynamic
        # mask on)
        return
    sel = slice(None) if keep is None else keep
    ro = rays_o.clone().requires_grad_(True)
    rd = rays_d.clone().requires_grad_(True)
    ref = no.render_batch_ray(ro[sel], rd[sel], depth[sel], grids, decs,
                              bound, 'color')
    ld = no.loss_dict(ref, depth[sel], color[sel], False, 'color')
    ref_loss = sum(ld.values())
    ref_loss.backward()
    assert abs(l1 - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    parity.assert_all([('track/g_rays_o', o1, ro.grad),
                       ('track/g_rays_d', d1, rd.grad)])


@pytest.mark.parametrize('stage,need_rays,need_dec', [
    ('middle', True, False), ('fine', True, False), ('color', True, True),
    ('color', False, True), ('middle', False, False), ('fine', False, False),
    ('color', True, False), ('color', False, False), ('coarse', False, False)])
def test_one_launch_mapping_iteration_vs_oracle(stage, need_rays, need_dec):
    """xrd_nice_map_iter (forward + mapping loss + backward as ONE launch) at
    BASELINE configs[1] (office0 grids, 1000 mapping rays, some of them masked
    out like the bbox pre-filter does) against the CPU oracle: loss, ray
    gradients, every grid gradient and the colour-decoder gradient; and twice
    in a row on the same workspace (the call must leave it reusable)."""
    import parity
    from xrdslam_amd.engine import nice as en
    dev = _cuda()
    n = 1000
    bound, grids, decs = _office0_case(1)
    rays_o, rays_d, depth, color = _office0_rays(n, 11)
    keep = torch.rand(n, generator=torch.Generator().manual_seed(4)) > 0.06
    og = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
    od = {kind: {m: v.clone().requires_grad_(kind == 'color')
                 for m, v in sd.items()} for kind, sd in decs.items()}
    ro = rays_o.clone().requires_grad_(stage != 'coarse')
    rd = rays_d.clone().requires_grad_(stage != 'coarse')
    # the reference drops the masked rays from the batch; max(gt_depth) of
    # the kept rays bounds the sampling range (conv_onet.py:418,455)
    ref = no.render_batch_ray(ro[keep], rd[keep], depth[keep], og, od, bound,
                              stage)
    ref_loss = sum(no.loss_dict(ref, depth[keep], color[keep], True,
                                stage).values())
    ref_loss.backward()
    scene, gl, flats = build_scene(bound, grids, decs, dev,
                                   color_requires_grad=True,
                                   grid_requires_grad=True)
    dmax = depth[keep].max().to(dev)
    for rep in range(2):
        for g in gl.values():
            if g.grad is not None:
                g.grad.zero_()
        loss, g_o, g_d, g_flat = en.nice_map_iter(
            scene, stage, rays_o.to(dev), rays_d.to(dev), depth.to(dev), dmax,
            color.to(dev), keep.to(dev).to(torch.uint8), 0.2, need_rays,
            need_dec)
        torch.cuda.synchronize()
        pairs = [('loss', loss, ref_loss)]
        if need_rays and stage != 'coarse':
            assert float(g_o[~keep.to(dev)].abs().max()) == 0.0
            pairs += [('g_rays_o', g_o[keep.to(dev)], ro.grad[keep]),
                      ('g_rays_d', g_d[keep.to(dev)], rd.grad[keep])]
        else:
            assert g_o is None and g_d is None
        for k, grid in gl.items():
            want = og[k].grad
            if want is None or float(want.abs().max()) == 0.0:
                assert grid.grad is None or float(grid.grad.abs().max()) == 0
                continue
            pairs.append((f'g_{k}', grid.grad, want))
        if need_dec:
            gf, off = g_flat.cpu(), 0
            for name, shape in en.param_shapes('color'):
                m = int(np.prod(shape))
                want = od['color'][name].grad
                got = gf[off:off + m].reshape(shape)
                off += m
                if want is None:
                    assert float(got.abs().max()) == 0.0, name
                    continue
                pairs.append((f'g_dec_color/{name}', got, want))
        else:
            assert g_flat is None
        parity.assert_all([(f'nice_map_iter/{stage}/rays{int(need_rays)}/'
                            f'rep{rep}/{m}', a, b) for m, a, b in pairs])


def test_adam_cells_matches_torch():
    from xrdslam_amd import _lib
    dev = _cuda()
    torch.manual_seed(0)
    ncell, cf = 500, 32
    p0 = torch.randn(ncell, cf, device=dev)
    idx = torch.randperm(ncell, device=dev)[:137].int().sort().values
    p_ref = p0[idx.long()].clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
    p = p0.clone()
    m = torch.zeros(idx.numel(), cf, device=dev)  # compact moments
    v = torch.zeros(idx.numel(), cf, device=dev)
    for step in range(1, 6):
        gfull = torch.randn(ncell, cf, device=dev)
        p_ref.grad = gfull[idx.long()].clone()
        opt.step()
        gwork = gfull.clone()
        _lib.check(_lib.lib().xrd_adam_cells(
            _lib.ptr(p), _lib.ptr(gwork), _lib.ptr(m), _lib.ptr(v),
            _lib.ptr(idx), idx.numel(), cf, 0.01, 0.9, 0.999, 1e-8, step, 1,
            _lib.stream_ptr(dev)))
        assert gwork[idx.long()].abs().max() == 0
    torch.cuda.synchronize()
    assert torch.allclose(p[idx.long()], p_ref.detach(), rtol=1e-5, atol=1e-6)
    mask = torch.ones(ncell, dtype=torch.bool, device=dev)
    mask[idx.long()] = False
    assert torch.equal(p[mask], p0[mask])


def test_fused_dense_adam_matches_torch_incl_skipped_parameters():
    """FusedDenseAdam (self-advancing device step counters, one launch per
    parameter or per flat group) against torch.optim.Adam over 7 steps,
    including a parameter whose grad is None in some steps: torch does not
    advance a skipped parameter's step (advisor finding, round 2: the shared
    counter did), and state_dict() reports the same step numbers."""
    from xrdslam_amd.engine.slam_ops import FusedDenseAdam
    dev = _cuda()
    torch.manual_seed(3)
    shapes = [(7, ), (4, 3), (33, ), (2, 5, 3)]
    ref = [torch.randn(*sh, device=dev).requires_grad_(True) for sh in shapes]
    mine = [r.detach().clone().requires_grad_(True) for r in ref]
    o_ref = torch.optim.Adam(ref, lr=0.02, betas=(0.9, 0.999), eps=1e-8)
    o_mine = FusedDenseAdam(mine, lr=0.02, betas=(0.9, 0.999), eps=1e-8)
    for step in range(7):
        for k, (a, b) in enumerate(zip(ref, mine)):
            if k == 2 and step in (2, 3, 5):     # stage-gated parameter
                a.grad = b.grad = None
                continue
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        o_ref.step()
        o_mine.step()
        if step in (1, 4):
            # state_dict() in the middle of a run must leave the live state
            # alone (torch returns the live per-parameter dicts; advisor
            # finding, round 3): step, state_dict, step, state_dict again
            mid = o_mine.state_dict()['state']
            assert all(v['step'].dim() == 0 for v in mid.values())
            for p in mine:
                if o_mine.state[p]:
                    assert o_mine.state[p]['step'].shape == (2, )
                    assert o_mine.state[p]['step'].is_cuda
    torch.cuda.synchronize()
    for a, b in zip(ref, mine):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (a - b).abs().max()
    sd_r, sd_m = o_ref.state_dict()['state'], o_mine.state_dict()['state']
    for k in sd_r:
        assert float(sd_m[k]['step']) == float(sd_r[k]['step']), k
        assert torch.allclose(sd_m[k]['exp_avg'], sd_r[k]['exp_avg'],
                              rtol=2e-6, atol=1e-7)


def test_dense_adam_of_several_optimisers_in_one_launch():
    """FusedDenseAdam.step_together (xrd_adam_dense_multi: one launch for the
    optimisers of several parameter groups — SplaTAM's five Gaussian tensors,
    a model's table / decoder / poses) against one torch.optim.Adam a group:
    different learning rates, sizes from 3 to 70 001 elements, a multi-
    parameter optimiser (launched as step() would) in the mix, a group that
    skips steps; bit-identical to stepping them one by one."""
    from xrdslam_amd.engine.slam_ops import FusedDenseAdam
    dev = _cuda()
    torch.manual_seed(5)
    shapes = [[(70001, )], [(3, )], [(4096, 3)], [(17, 4), (9, )], [(1000, )],
              [(33, )]]
    lrs = [1e-2, 3e-3, 2e-2, 1e-3, 5e-3, 1e-1]
    ref = [[torch.randn(*sh, device=dev).requires_grad_(True) for sh in grp]
           for grp in shapes]
    tog = [[r.detach().clone().requires_grad_(True) for r in grp]
           for grp in ref]
    one = [[r.detach().clone().requires_grad_(True) for r in grp]
           for grp in ref]
    o_ref = [torch.optim.Adam(g, lr=lr) for g, lr in zip(ref, lrs)]
    o_tog = [FusedDenseAdam(g, lr=lr) for g, lr in zip(tog, lrs)]
    o_one = [FusedDenseAdam(g, lr=lr) for g, lr in zip(one, lrs)]
    for step in range(6):
        for k in range(len(shapes)):
            skip = k == 4 and step in (1, 2)
            for a, b, c in zip(ref[k], tog[k], one[k]):
                if skip:
                    a.grad = b.grad = c.grad = None
                    continue
                g = torch.randn_like(a)
                a.grad, b.grad, c.grad = g.clone(), g.clone(), g.clone()
        for o in o_ref + o_one:
            o.step()
        FusedDenseAdam.step_together(o_tog)
    torch.cuda.synchronize()
    for k in range(len(shapes)):
        for a, b, c in zip(ref[k], tog[k], one[k]):
            assert torch.equal(b, c), (k, (b - c).abs().max())
            assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), \
                (k, (a - b).abs().max())
        sd_r = o_ref[k].state_dict()['state']
        sd_t = o_tog[k].state_dict()['state']
        for i in sd_r:
            assert float(sd_t[i]['step']) == float(sd_r[i]['step']), (k, i)
    # two sets on ONE counter in a launch would race: refused by the C-ABI
    from xrdslam_amd import _lib
    sets = (_lib.AdamDenseSet * 2)()
    st = o_tog[0].state[tog[0][0]]
    for k in range(2):
        sets[k].param = _lib.ptr(tog[0][0])
        sets[k].grad = _lib.ptr(tog[0][0].grad)
        sets[k].m, sets[k].v = _lib.ptr(st['exp_avg']), \
            _lib.ptr(st['exp_avg_sq'])
        sets[k].n, sets[k].lr = 4, 1e-3
        sets[k].step_ticket, sets[k].advance = _lib.ptr(st['step']), 1
    rc = _lib.lib().xrd_adam_dense_multi(2, sets, 0.9, 0.999, 1e-8,
                                         _lib.stream_ptr(dev))
    assert rc != 0


def test_fused_cell_adam_self_advancing_counter_matches_torch():
    """FusedCellAdam through xrd_adam_cells_tick (the kernel advances its own
    step counter): 5 steps against torch.optim.Adam over val[mask]"""
    from xrdslam_amd.engine import nice as en
    from xrdslam_amd.slam.engine.optimizers import FusedCellAdam
    dev = _cuda()
    torch.manual_seed(1)
    p = en.to_channels_last_grid(torch.randn(1, 32, 5, 6, 7, device=dev))
    p.requires_grad_(True)
    ncell = 5 * 6 * 7
    idx = torch.randperm(ncell, device=dev)[:61].int().sort().values
    cells = p.detach().permute(0, 2, 3, 4, 1).reshape(ncell, 32)
    p_ref = cells[idx.long()].clone().requires_grad_(True)
    o_ref = torch.optim.Adam([p_ref], lr=0.01)
    p._xrd_cells, p._xrd_cells_count = idx, None
    opt = FusedCellAdam([p], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
    for step in range(5):
        g = torch.randn(ncell, 32, device=dev)
        p_ref.grad = g[idx.long()].clone()
        o_ref.step()
        p.grad = g.reshape(1, 5, 6, 7, 32).permute(0, 4, 1, 2, 3)
        assert en._is_cl(p.grad)
        p._xrd_grad_fresh = True
        opt.step()
    torch.cuda.synchronize()
    assert int(opt._step_dev[0]) == 5 and int(opt._step_dev[1]) == 0
    got = p.detach().permute(0, 2, 3, 4, 1).reshape(ncell, 32)[idx.long()]
    assert torch.allclose(got, p_ref.detach(), rtol=1e-5, atol=1e-6)


def test_grids_stepped_in_one_launch_equal_one_launch_each():
    """FusedCellAdam.step_together (xrd_adam_cells_multi: the feature grids of
    a mapping stage in ONE launch) against one xrd_adam_cells_tick launch a
    grid: same parameters, moments and step counters bit for bit, over 4
    steps, with different learning rates / selections / sizes, a device-side
    count, one empty selection and one grid that skips a step"""
    from xrdslam_amd.engine import nice as en
    from xrdslam_amd.slam.engine.optimizers import FusedCellAdam
    dev = _cuda()

    def build():
        torch.manual_seed(4)
        out = []
        for k, (shape, n_sel, lr) in enumerate(
                (((5, 6, 7), 61, 0.01), ((9, 8, 7), 200, 0.003),
                 ((4, 4, 4), 64, 0.02), ((3, 3, 3), 0, 0.01))):
            p = en.to_channels_last_grid(torch.randn(1, 32, *shape,
                                                     device=dev))
            p.requires_grad_(True)
            ncell = shape[0] * shape[1] * shape[2]
            idx = torch.randperm(ncell, device=dev)[:n_sel].int() \
                .sort().values
            p._xrd_cells, p._xrd_cells_count = idx, None
            if k == 1:      # capacity buffer + device-side count
                cap = torch.zeros(ncell, dtype=torch.int32, device=dev)
                cap[:n_sel] = idx
                p._xrd_cells = cap
                p._xrd_cells_count = torch.tensor([n_sel], dtype=torch.int32,
                                                  device=dev)
            out.append((p, FusedCellAdam([p], lr=lr, betas=(0.9, 0.999),
                                         eps=1e-8)))
        return out

    runs = []
    for together in (False, True):
        grids = build()
        torch.manual_seed(9)
        for step in range(4):
            for k, (p, opt) in enumerate(grids):
                p.grad = torch.randn_like(p)
                p._xrd_grad_fresh = not (k == 2 and step == 1)
            if together:
                done = FusedCellAdam.step_together([o for _, o in grids])
                assert len(done) == len(grids)
            else:
                for _, o in grids:
                    o.step()
        torch.cuda.synchronize()
        runs.append([(p.detach().clone(), o._m, o._v, o._step_dev)
                     for p, o in grids])
    for (pa, ma, va, sa), (pb, mb, vb, sb) in zip(*runs):
        assert torch.equal(pa, pb)
        if ma is not None:
            assert torch.equal(ma, mb) and torch.equal(va, vb)
            assert torch.equal(sa, sb)
    assert [int(r[3][0]) for r in runs[1][:3]] == [4, 4, 3]
    assert runs[1][3][3] is None        # the empty selection never stepped


def test_fused_cell_adam_empty_selection_is_noop():
    """a frustum mask that selects no cell of a grid: the reference's Adam over
    an empty ``val[mask]`` does nothing; so must the fused one (zero-sized
    moments have NULL data pointers)"""
    from xrdslam_amd.engine import nice as en
    from xrdslam_amd.slam.engine.optimizers import FusedCellAdam
    dev = _cuda()
    p = en.to_channels_last_grid(torch.randn(1, 32, 4, 5, 6, device=dev))
    p.requires_grad_(True)
    before = p.detach().clone()
    p.grad = torch.zeros_like(p)
    p._xrd_cells = torch.zeros(0, dtype=torch.int32, device=dev)
    p._xrd_grad_fresh = True
    opt = FusedCellAdam([p], lr=0.1, betas=(0.9, 0.999), eps=1e-8)
    opt.step()
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(p.detach(), before)


def test_point_queries_match_oracle_and_mesher_runs():
    """xrd_nice_eval_points (the mesher's query_fn / color_func) against the
    oracle's eval_points, incl. the out-of-bound override; then the mesher on
    the model hooks"""
    import nice_oracle as no
    from xrdslam_amd.engine import nice as en
    dev = torch.device('cuda:0')
    g, bound, grids, decs, _ = load_nice_golden()
    scene = en.NiceScene(bound, device=dev)
    for k, v in grids.items():
        scene.set_grid(k, en.to_channels_last_grid(v.to(dev)))
    for kind, sd in decs.items():
        scene.set_decoder(kind, en.flatten_state_dict(sd, kind).to(dev))
    gen = torch.Generator().manual_seed(3)
    lo, hi = bound[:, 0].float(), bound[:, 1].float()
    p = lo + (hi - lo) * (torch.rand(5000, 3, generator=gen) * 1.2 - 0.1)
    for stage in ('fine', 'color'):
        ref = no.eval_points(p, grids, decs, bound, stage)
        got = en.nice_eval_points(scene, stage, p.to(dev)).cpu()
        cols = [3] if stage == 'fine' else [0, 1, 2, 3]
        for c in cols:
            assert rel_err(got[:, c], ref[:, c].detach()) < 1e-4, (stage, c)
        outside = ((p < lo) | (p > hi)).any(1)
        assert outside.any() and bool((got[outside, 3] == 100).all())
    # the mesher over the hooks: a lattice strictly inside the bound
    from xrdslam_amd.slam.common.mesher import Mesher, MesherConfig
    inner = torch.stack([lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo)], 1)
    m = Mesher(MesherConfig(resolution=24, points_batch_size=5000), None,
               bound, inner)
    mesh = m.get_mesh([], lambda q: en.nice_eval_points(scene, 'fine', q),
                      lambda q: en.nice_eval_points(scene, 'color', q),
                      device=dev)
    if mesh is not None:    # random-init decoders may have no zero crossing
        assert mesh.faces.max() < mesh.vertices.shape[0]
        assert mesh.vertex_colors.shape[0] == mesh.vertices.shape[0]


@pytest.mark.parametrize('n', [1, 7, 200, 1024, 1025, 3000])
@pytest.mark.parametrize('masked', [False, True])
def test_tracking_loss_median_both_selection_paths(n, masked):
    """xrd_nice_loss, tracking branch with the dynamic-object mask
    (conv_onet.py:152-166: residual < 10 x torch.median of the kept rays):
    the order statistic comes from rank counting up to 1024 rays and from the
    bitonic network above; against the torch formulation in f64, with ties"""
    from xrdslam_amd.engine import slam_ops
    dev = _cuda()
    g = torch.Generator().manual_seed(n)
    depth = (1.0 + torch.rand(n, generator=g, dtype=torch.float64)).to(dev)
    var = (0.01 + torch.rand(n, generator=g, dtype=torch.float64)).to(dev)
    td = (1.0 + torch.rand(n, generator=g)).to(dev)
    td[torch.rand(n, generator=g) < 0.1] = 0.0
    if n > 8:
        # exact ties around the median and a few dynamic outliers
        depth[: n // 4] = 1.5
        var[: n // 4] = 0.04
        td[: n // 4] = 1.75
        td[-max(n // 20, 1):] += 50.0
    rgb = torch.rand(n, 3, generator=g).to(dev).requires_grad_(True)
    tc = torch.rand(n, 3, generator=g).to(dev)
    keep = None
    if masked:
        keep = (torch.rand(n, generator=g) > 0.2).to(dev)
        if not bool(keep.any()):
            keep[0] = True
    d = depth.clone().requires_grad_(True)
    loss = slam_ops.NiceLossFn.apply(
        d, var, rgb, td, tc, None if keep is None else keep.to(torch.uint8),
        False, True, True, 0.5)
    loss.backward()
    d2 = depth.clone().requires_grad_(True)
    rgb2 = rgb.detach().clone().requires_grad_(True)
    kp = torch.ones(n, dtype=torch.bool, device=dev) if keep is None else keep
    res = (td.double() - d2).abs() / torch.sqrt(var + 1e-10)
    med = res[kp].median() if bool(kp.any()) else res.new_tensor(1e300)
    m = kp & (td > 0) & (res < 10 * med)
    want = res[m].sum() + 0.5 * (tc[m] - rgb2[m]).abs().sum().double()
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-6 * max(abs(float(want)), 1)
    assert torch.allclose(d.grad, d2.grad, rtol=1e-12, atol=0)
    assert torch.equal(rgb.grad, rgb2.grad)


@pytest.mark.parametrize('n,masked', [(200, False), (333, True), (340, False),
                                      (5, False), (227, True), (228, False)])
def test_tracking_backward_from_the_forwards_masks(n, masked):
    """xrd_nice_render_fwd_masks / _bwd_masks (the forward keeps the decoders'
    ReLU masks, the decoder-per-block backward skips its forward recompute)
    against the plain pair on the same batch: same loss and — the masks being
    the ones the recompute would produce — the same ray gradients bit for
    bit; above the path's ray limit the call is refused"""
    from xrdslam_amd import _lib
    from xrdslam_amd.engine import nice as en
    dev = _cuda()
    bound, grids, decs = _office0_case(1)
    rays_o, rays_d, depth, color = _office0_rays(n, 21)
    keep = None
    if masked:
        keep = (torch.rand(n, generator=torch.Generator().manual_seed(2))
                > 0.1).to(dev).to(torch.uint8)
    scene, gl, flats = build_scene(bound, grids, decs, dev)
    args = (scene, rays_o.to(dev), rays_d.to(dev), depth.to(dev), None,
            color.to(dev), keep, True, True, 0.5)
    out = {}
    for flag in (True, False):
        en.TRACK_KEEP_MASKS = flag
        try:
            loss, g_o, g_d = en.nice_track_iter(*args)
        finally:
            en.TRACK_KEEP_MASKS = True
        torch.cuda.synchronize()
        out[flag] = (float(loss), g_o.cpu(), g_d.cpu())
    assert out[True][0] == out[False][0]
    assert torch.equal(out[True][1], out[False][1])
    assert torch.equal(out[True][2], out[False][2])
    assert float(out[True][1].abs().max()) > 0
    lib = _lib.lib()
    assert lib.xrd_nice_fwd_masks_words(n) == n * (9 * 64 + 32)
    import ctypes as C
    cs = scene.c_struct()
    p = C.c_void_p(16)
    assert lib.xrd_nice_render_fwd_masks(C.byref(cs), 3, 341, p, p, p, p, p,
                                         p, p, p, p, None) == 3   # unsupported
    assert lib.xrd_nice_render_fwd_masks(C.byref(cs), 3, 8, p, p, p, p, p, p,
                                         p, p, None, None) == 1   # no masks


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1000, 37])
def test_partial_rows_need_no_zeroing(n):
    """the blocks' partial rows of the colour decoder's gradient (the one-launch
    mapping iteration's workspace) are written with plain stores by each
    block's first group and NOT zeroed by the finishing launch any more: every
    entry of a row must be stored in every launch.  The rows are poisoned with
    NaN between two identical calls; the gradient must come out the same."""
    from xrdslam_amd.engine import nice as en
    dev = _cuda()
    bound, grids, decs = _office0_case(1)
    rays_o, rays_d, depth, color = _office0_rays(n, 5)
    scene, gl, flats = build_scene(bound, grids, decs, dev,
                                   color_requires_grad=True,
                                   grid_requires_grad=True)
    args = (scene, 'color', rays_o.to(dev), rays_d.to(dev), depth.to(dev),
            depth.max().to(dev), color.to(dev), None, 0.2, False, True)
    _, _, _, g1 = en.nice_map_iter(*args)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    ws = scene._map_ws[(False, n)]
    rep_off = n * 3 * 6 * 2 + 2 * n + 4
    n_rows = min((n + 3) // 4, 256)
    flat_len = g1.numel()
    ws[rep_off:rep_off + n_rows * flat_len] = float('nan')
    _, _, _, g2 = en.nice_map_iter(*args)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g2).all())          # nothing poisoned is read
    # bit for bit outside embedder._B (its sums meet in LDS float atomics:
    # the order, hence the last bit, is the hardware's)
    off, same = 0, torch.ones_like(g1, dtype=torch.bool)
    for name, shape in en.param_shapes('color'):
        m = int(np.prod(shape))
        if name == 'embedder._B':
            same[off:off + m] = False
        off += m
    assert torch.equal(g1[same], g2[same])
    scale = float(g1[~same].abs().max())
    assert float((g1[~same] - g2[~same]).abs().max()) <= 1e-5 * scale


@pytest.mark.gpu
def test_map_iter_with_more_groups_than_blocks():
    """1500 rays = 375 groups on 256 persistent blocks: some blocks run a
    second group, whose weight-gradient blocks are ADDED to the row the first
    group stored (and whose operand ring is refilled behind a barrier).  The
    mapping losses are plain sums over rays: the call must equal the sum of a
    call on rays [0, 1024) — exactly one group a block — and one on the rest,
    with the same dmax."""
    from xrdslam_amd.engine import nice as en
    dev = _cuda()
    n = 1500
    bound, grids, decs = _office0_case(1)
    rays_o, rays_d, depth, color = _office0_rays(n, 21)
    dmax = depth.max().to(dev)

    def run(lo, hi):
        scene, gl, _ = build_scene(bound, grids, decs, dev,
                                   color_requires_grad=True,
                                   grid_requires_grad=True)
        loss, _, _, g = en.nice_map_iter(
            scene, 'color', rays_o[lo:hi].to(dev), rays_d[lo:hi].to(dev),
            depth[lo:hi].to(dev), dmax, color[lo:hi].to(dev), None, 0.2,
            False, True)
        torch.cuda.synchronize()
        return float(loss), g.double(), {k: v.grad.double().clone()
                                         for k, v in gl.items()
                                         if v.grad is not None}
    la, ga, gga = run(0, n)
    lb, gb, ggb = run(0, 1024)
    lc, gc, ggc = run(1024, n)
    assert abs(la - (lb + lc)) <= 1e-9 * abs(la)
    want = gb + gc
    scale = float(want.abs().max())
    assert float((ga - want).abs().max()) <= 2e-5 * scale
    for k in gga:
        w = ggb[k] + ggc[k]
        assert float((gga[k] - w).abs().max()) <= 2e-5 * float(w.abs().max())
