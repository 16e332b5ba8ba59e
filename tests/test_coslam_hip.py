"""GPU: Co-SLAM on the HIP encodings.  (1) JointEncoding (hash grid + OneBlob
kernels through the C-ABI, 2x32 MLP, SDF rendering, all loss terms incl.
smoothness) against the golden made from the reference's own model, 1e-4 rel;
(2) a short CoSLAM tracking/mapping run on the synthetic room."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import coslam_golden_util as cg  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('tag,is_mapping,first', cg.TAGS)
def test_joint_encoding_vs_reference(tag, is_mapping, first, fused):
    """fused=True: one render kernel forward, one backward (xrd_coslam_*);
    fused=False: modular HIP encodings + torch MLPs"""
    g = np.load(cg.GOLDEN)
    model = cg.build_model(g, 'cuda:0')
    model.use_fused = fused
    assert (model._fused_tables('cuda:0') is not None) == fused
    errs = cg.run_case(model, g, tag, is_mapping, first, 'cuda:0')
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad


@pytest.mark.parametrize('mode', ['fused', 'fused_loss', 'modular'])
@pytest.mark.parametrize('tag,is_mapping,first,n', cg.OFFICE0_CASES)
def test_baseline_config_vs_reference(tag, is_mapping, first, n, mode):
    """BASELINE configuration: the reference's DEFAULT JointEncodingConfig
    (2^16-entry table: levels 5..15 hashed), office0 bound, 1024 tracking /
    2389 mapping rays, against tests/golden/coslam_office0.npz (made from the
    reference's own model).  1e-4 in the max norm AND element-wise
    (tests/parity.py) for outputs, losses, ray / table / decoder gradients."""
    import parity
    gold = np.load(cg.OFFICE0)
    model = cg.build_office0_model('cuda:0')
    model.use_fused = mode != 'modular'
    model.fused_losses = mode == 'fused_loss'
    assert (model._fused_tables('cuda:0') is not None) == model.use_fused
    got = cg.run_office0_case(model, tag, is_mapping, first, n, 'cuda:0')
    pairs = cg.office0_pairs(got, gold, tag)
    if mode == 'fused_loss':
        # the fused loss reports the four data terms as one value
        data = [k for k in gold.files if k.startswith(f'{tag}/loss_')
                and not k.endswith('smooth_loss')]
        total = sum(float(v) for k, v in got.items() if k.startswith('loss_')
                    and not k.endswith('smooth_loss'))
        pairs.append((f'coslam_office0/{tag}/loss_data_total', total,
                      sum(float(gold[k]) for k in data)))
    parity.assert_all([(f'{mode}:{n_}', a, b) for n_, a, b in pairs])


@pytest.mark.parametrize('tag,is_mapping', cg.VARIANT_TAGS)
@pytest.mark.parametrize('name', list(cg.VARIANTS))
def test_non_default_model_options_vs_reference(name, tag, is_mapping):
    """the configurations the fused renderer declines (a separate colour
    grid; the importance-sampling pass) run on the modular HIP encodings:
    both hash grids through xrd_hashgrid_*, OneBlob through xrd_oneblob_*;
    against tests/golden/coslam_variants.npz, element-wise 1e-4"""
    import parity
    g = np.load(cg.VARIANT_GOLDEN)
    model, grids = cg.build_variant(g, name, 'cuda:0')
    assert model._fused_tables('cuda:0') is None
    parity.assert_all(cg.run_variant(model, grids, g, name, tag, is_mapping,
                                     'cuda:0'))


@pytest.mark.parametrize('tag,is_mapping,first', cg.TAGS)
def test_fused_loss_vs_reference(tag, is_mapping, first):
    """xrd_coslam_loss (loss terms + their gradients through the fused
    renderer) against the reference's get_loss_dict"""
    g = np.load(cg.GOLDEN)
    model = cg.build_model(g, 'cuda:0')
    model.fused_losses = True
    errs = cg.run_case(model, g, tag, is_mapping, first, 'cuda:0')
    l5 = model.last_loss_terms.cpu().numpy()
    for k, name in enumerate(('rgb', 'depth', 'sdf', 'fs')):
        errs[f'term_{name}'] = cg.rel_err(l5[k + 1],
                                          g[f'{tag}/loss_{name}_loss'])
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad


def test_map_rows_kernel_equals_the_torch_gathers():
    """xrd_coslam_map_rows against the index / floor-divide / gather / cat
    chain it replaces (coslam.py:139-150,152-210), bit for bit"""
    from xrdslam_amd.engine import slam_ops
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(1)
    HW, per_kf, K = 640 * 480, 15360, 7
    bank = torch.rand(2 * K * per_kf, 7, generator=g).to(dev)
    dirs = torch.randn(HW, 3, generator=g).to(dev)
    rgb = torch.rand(HW, 3, generator=g).to(dev)
    depth = torch.rand(HW, 1, generator=g).to(dev)
    idx = torch.randint(0, K * per_kf, (2048, ), generator=g).to(dev)
    pix = torch.randint(0, HW, (341, ), generator=g).to(dev)
    cur = torch.tensor([K], dtype=torch.int64, device=dev)
    rows, ids = slam_ops.coslam_map_rows(bank, idx, per_kf, pix, dirs, rgb,
                                         depth, cur)
    want = torch.cat([bank[idx], torch.cat([dirs[pix], rgb[pix],
                                            depth[pix]], -1)], 0)
    want_ids = torch.cat([torch.div(idx, per_kf, rounding_mode='floor'),
                          cur.expand(341)], 0)
    assert torch.equal(rows, want) and torch.equal(ids, want_ids)


def test_fused_smoothness_equals_the_torch_formulation():
    """the smoothness term on xrd_hashgrid_tv (lattice points, hash features,
    TV loss and feature gradient as three launches; its table gradient either
    scattered on its own or handed to a pending render backward, whose ONE
    scatter launch then carries both) against the torch formulation
    (JointEncoding.smoothness, pinned to the reference by the goldens): same
    draws, loss and table gradient, alone and next to the render."""
    from xrdslam_amd.engine import coslam as ec
    model = cg.build_office0_model('cuda:0')
    cfg = model.config
    w = cfg.trainging_smooth_weight * 1e4     # well above f32 noise
    dev = 'cuda:0'
    tab = model.embed_fn.params

    def draws():
        gen = torch.Generator(device=dev).manual_seed(5)
        return lambda shape, like: torch.rand(shape, device=like.device,
                                              dtype=like.dtype, generator=gen)

    # --- alone -------------------------------------------------------------
    tab.grad = None
    model._rand = draws()
    ref = model.smoothness(cfg.trainging_smooth_pts, cfg.trainging_smooth_vox,
                           cfg.trainging_smooth_margin) * w
    ref.backward()
    g_ref = tab.grad.clone()
    tab.grad = None
    model._rand = draws()
    model._render_bwd_pending = False
    got = ec.smoothness(model, cfg.trainging_smooth_pts - 1,
                        cfg.trainging_smooth_vox, cfg.trainging_smooth_margin,
                        w)
    got.backward()
    torch.cuda.synchronize()
    assert abs(float(got) - float(ref)) < 1e-5 * abs(float(ref))
    assert float(g_ref.abs().max()) > 0
    err = float((tab.grad - g_ref).abs().max() / g_ref.abs().max())
    assert err < 1e-4, err
    # --- next to the fused render: one scatter for both ---------------------
    n = 512
    ro, rd, depth, color = [t.to(dev) for t in cg.office0_inputs(n, 3)]
    inp = {'rays_o': ro, 'rays_d': rd, 'target_d': depth, 'target_s': color,
           'first': False}
    res = {}
    for fused in (False, True):
        model.fused_smoothness = fused
        model.fused_losses = True
        for p_ in model.parameters():
            p_.grad = None
        tab.grad = None
        model._rand = draws()
        out = model.get_outputs(inp)
        ld = model.get_loss_dict(out, inp, True, 0)
        assert 'smooth_loss' in ld
        sum(ld.values()).backward()
        torch.cuda.synchronize()
        res[fused] = (float(ld['smooth_loss']), tab.grad.clone())
        if fused:       # the lattice left with the render's scatter
            assert model._smooth_stash is None
            assert not model._render_bwd_pending
    del model._rand
    model.fused_smoothness = True
    assert abs(res[True][0] - res[False][0]) < 1e-5 * abs(res[False][0])
    err = float((res[True][1] - res[False][1]).abs().max() /
                res[False][1].abs().max())
    assert err < 1e-4, err


def test_fused_tracking_iteration_matches_generic_hooks():
    """CoSLAM.get_loss through the fused launches (sampling, render, loss)
    equals the generic get_model_input -> model -> get_loss_dict path on the
    same random draws: loss value and pose gradient"""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.common.frame import Frame
    from xrdslam_amd.slam.configs.input_config import coslam_config
    bound = [[-3, 3], [-4, 2.5], [-2, 2.5]]
    cam = Camera(fx=150., fy=150., cx=79.5, cy=59.5, width=160, height=120)
    cfg = coslam_config(bound)
    cfg.tracking_Wedge = cfg.tracking_Hedge = 5
    torch.manual_seed(0)
    algo = cfg.setup(camera=cam, device='cuda:0')
    with torch.no_grad():
        algo.model.embed_fn.params.normal_(0, 0.05)
    data = SyntheticRoom(bound, H=120, W=160, fx=150., fy=150., cx=79.5,
                         cy=59.5, n_frames=4, device='cuda:0')
    d = data[1]
    res = {}
    for fused in (True, False):
        algo.fused_iteration = fused
        f = Frame(fid=1, rgb=d['rgb'], depth=d['depth'],
                  gt_pose=d['c2w'].astype(np.float32),
                  init_pose=data[0]['c2w'].astype(np.float32),
                  separate_LR=True, rot_rep='axis_angle', device='cuda:0')
        torch.manual_seed(5)
        loss = algo.get_loss([f], False, 0, 10)
        loss.backward()
        res[fused] = (float(loss.detach()),
                      [p.grad.clone() for p in f.get_params()])
    assert abs(res[True][0] - res[False][0]) < 1e-4 * abs(res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert (a - b).abs().max() < 2e-4 * b.abs().max()


def test_sharded_loss_kernels_equal_the_unsharded_call():
    """xrd_coslam_loss_stats + xrd_coslam_loss_grads with all-reduced totals
    (multi-GPU mapping) on two halves of a batch = xrd_coslam_loss on the whole
    batch: same loss terms, same gradients for every ray"""
    from xrdslam_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    n, S = 301, 43
    dev = 'cuda:0'
    maps = torch.rand(n, 8, generator=g).to(dev)
    z = torch.sort(torch.rand(n, S, generator=g) * 4, dim=1).values.to(dev)
    raw = torch.randn(n, S, 4, generator=g).to(dev)
    td = (1.0 + 2 * torch.rand(n, generator=g))
    td[::9] = 0.0
    td = td.to(dev)
    tc = torch.rand(n, 3, generator=g).to(dev)
    cfg = (5.0, 0.1, 1000.0, 10.0, 0.1, 100.0, 0.05)
    st = _lib.stream_ptr(torch.device(dev))
    P = _lib.ptr

    def full():
        l5 = torch.empty(5, device=dev)
        gm, gr = torch.empty(n, 8, device=dev), torch.empty(n, S, 4, device=dev)
        ws = torch.empty(n * 8, device=dev)
        _lib.check(lib.xrd_coslam_loss(n, S, *cfg, P(maps), P(z), P(raw),
                                       P(td), P(tc), P(l5), P(gm), P(gr),
                                       P(ws), st), 'loss')
        return l5, gm, gr

    l5, gm, gr = full()
    cut = 140
    parts = [(0, cut), (cut, n)]
    stats = []
    for lo, hi in parts:
        s_ = torch.empty(hi - lo, 8, device=dev)
        _lib.check(lib.xrd_coslam_loss_stats(
            hi - lo, S, *cfg[4:], P(maps[lo:hi].contiguous()),
            P(z[lo:hi].contiguous()), P(raw[lo:hi].contiguous()),
            P(td[lo:hi].contiguous()), P(tc[lo:hi].contiguous()), P(s_), st),
            'stats')
        stats.append(s_)
    totals = sum(s_[:, :7].double().sum(0) for s_ in stats).contiguous()
    for (lo, hi), s_ in zip(parts, stats):
        l5s = torch.empty(5, device=dev)
        gms = torch.empty(hi - lo, 8, device=dev)
        grs = torch.empty(hi - lo, S, 4, device=dev)
        _lib.check(lib.xrd_coslam_loss_grads(
            hi - lo, S, *cfg, P(maps[lo:hi].contiguous()),
            P(z[lo:hi].contiguous()), P(raw[lo:hi].contiguous()),
            P(td[lo:hi].contiguous()), P(tc[lo:hi].contiguous()), P(s_),
            P(totals), n, P(l5s), P(gms), P(grs), st), 'grads')
        assert torch.allclose(l5s, l5, rtol=1e-5)
        assert torch.allclose(gms, gm[lo:hi], rtol=1e-5, atol=1e-9)
        assert torch.allclose(grs, gr[lo:hi], rtol=1e-5, atol=1e-9)


def test_axis_angle_pose_kernel_matches_torch_formula():
    """xrd_pose_aa_fwd/bwd against the torch restatement of
    OptimizablePose.matrix (checked on the CPU against the per-frame module
    in tests/test_coslam_host.py)"""
    from xrdslam_amd.slam.utils.opt_pose import (
        axis_angle_translation_to_matrix)
    g = torch.Generator().manual_seed(0)
    rot = torch.randn(70, 3, generator=g) * 0.9
    rot[3] = 0.0
    rot[5] *= 1e-3
    trans = torch.randn(70, 3, generator=g)
    w = torch.randn(70, 4, 4, generator=g)
    ra, ta = rot.clone().requires_grad_(True), trans.clone().requires_grad_(True)
    Ma = axis_angle_translation_to_matrix(ra, ta)         # torch ops, CPU
    (Ma * w).sum().backward()
    rb = rot.cuda().requires_grad_(True)
    tb = trans.cuda().requires_grad_(True)
    Mb = axis_angle_translation_to_matrix(rb, tb)         # fused kernels
    (Mb * w.cuda()).sum().backward()
    assert torch.allclose(Ma, Mb.cpu(), atol=2e-6)
    assert torch.allclose(ra.grad, rb.grad.cpu(), atol=2e-5, rtol=1e-4)
    assert torch.allclose(ta.grad, tb.grad.cpu(), atol=1e-6)


def test_sample_distinct_is_a_random_subset():
    from xrdslam_amd.engine import slam_ops
    torch.manual_seed(1)
    for total, n in ((307200, 2048), (15360 * 7, 2048), (5000, 5000), (3, 1)):
        idx = slam_ops.sample_distinct(total, n, 'cuda:0').cpu().numpy()
        assert idx.min() >= 0 and idx.max() < total
        assert len(np.unique(idx)) == n          # without replacement
    a = slam_ops.sample_distinct(307200, 4096, 'cuda:0').cpu().numpy()
    b = slam_ops.sample_distinct(307200, 4096, 'cuda:0').cpu().numpy()
    assert len(np.intersect1d(a, b)) < 200       # fresh keys per call
    # spread: each decile of the range gets its share
    hist = np.histogram(a, bins=10, range=(0, 307200))[0]
    assert hist.min() > 300 and hist.max() < 520


def test_pose_rays_matches_reference_gather():
    """rays_d = sum(dir * R[ids]), rays_o = t[ids] (coslam.py:196-204) and
    the gradient w.r.t. the poses"""
    from xrdslam_amd.engine import slam_ops
    g = torch.Generator().manual_seed(0)
    n_pose, n = 7, 3000
    c2w = torch.randn(n_pose, 4, 4, generator=g).cuda()
    rows = torch.randn(n, 7, generator=g).cuda()
    ids = torch.randint(0, n_pose, (n, ), generator=g).cuda()
    ids[-500:] = n_pose - 1
    wo = torch.randn(n, 3, generator=g).cuda()
    wd = torch.randn(n, 3, generator=g).cuda()
    a = c2w.clone().requires_grad_(True)
    ro, rd = slam_ops.PoseRaysFn.apply(a, rows, ids)
    ((ro * wo).sum() + (rd * wd).sum()).backward()
    b = c2w.clone().requires_grad_(True)
    rd_ref = (rows[:, None, :3] * b[ids, :3, :3]).sum(-1)
    ro_ref = b[ids, :3, 3]
    ((ro_ref * wo).sum() + (rd_ref * wd).sum()).backward()
    assert torch.allclose(ro, ro_ref) and torch.allclose(rd, rd_ref,
                                                         atol=1e-5)
    assert (a.grad - b.grad).abs().max() < 1e-4 * b.grad.abs().max()


def test_coslam_loop_tracks_synthetic_room():
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import cadence, coslam_config
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    bound = [[-3, 3], [-4, 2.5], [-2, 2.5]]
    cam = Camera(fx=150., fy=150., cx=79.5, cy=59.5, width=160, height=120)
    cfg = coslam_config(bound)
    cfg.mapping_first_n_iters = 100
    cfg.tracking_Wedge = cfg.tracking_Hedge = 5
    # the bank holds 5 % of 160x120 = 960 rays per keyframe: like python's
    # random.sample, a batch larger than the bank is an error
    cfg.mapping_sample = 768
    algo = cfg.setup(camera=cam, device='cuda:0')
    data = SyntheticRoom(bound, H=120, W=160, fx=150., fy=150., cx=79.5,
                         cy=59.5, n_frames=200, device='cuda:0')
    cad = cadence['co-slam']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device='cuda:0')
    for k in range(12):
        slam.step(k)
    assert len(algo.keyframe_graph) == 3
    assert algo.rays.shape == (3 * algo.num_rays_to_save, 7)
    ate = slam.ate_rmse()
    # constant-velocity init alone drifts by centimetres on this trajectory
    assert ate < 0.02, ate
    with torch.no_grad():
        _, depth = algo.render_img(algo.get_estimate_c2w_list()[10].to(
            'cuda:0'), gt_depth=data[10]['depth'])
    gt = np.asarray(data[10]['depth'].cpu() if torch.is_tensor(
        data[10]['depth']) else data[10]['depth'])
    err = np.abs(depth - gt)[gt > 0].mean()
    assert err < 0.1, err
    # IMAGE level: the fused renderer against the modular path (the mirror of
    # JointEncoding.render_rays on the HIP encodings + torch MLPs, itself held
    # to goldens made by the reference's model), same pose, same draws
    pose = algo.get_estimate_c2w_list()[10].to('cuda:0')
    imgs = {}
    for fused in (True, False):
        algo.model.use_fused = fused
        algo.model._fused_ok = None
        torch.manual_seed(123)
        with torch.no_grad():
            imgs[fused] = algo.render_img(pose, gt_depth=data[10]['depth'])
    algo.model.use_fused = True
    for name, a, b in (('color', imgs[True][0], imgs[False][0]),
                       ('depth', imgs[True][1], imgs[False][1])):
        scale = max(float(np.abs(b).max()), 1e-30)
        dev_px = np.abs(a - b).reshape(120 * 160, -1).max(1) / scale
        line = (f'co-slam render_img 160x120 {name}: fused vs modular max '
                f'{dev_px.max():.2e}, pixels > 1e-4: '
                f'{float((dev_px > 1e-4).mean()):.3%}')
        rep = os.environ.get('XRD_PARITY_REPORT')
        if rep:
            with open(rep, 'a') as f:
                f.write(line + '\n')
        assert (dev_px > 1e-4).mean() <= 0.002 and dev_px.max() < 1e-2, line


def test_live_count_loss_equals_the_trimmed_batch():
    """xrd_coslam_loss_live on a capacity batch (live count on the device) =
    xrd_coslam_loss on the first n_live rows: loss terms, gradients of the
    live rows; zero gradients behind them"""
    from xrdslam_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(4)
    n, S, live = 333, 43, 250
    dev = 'cuda:0'
    maps = torch.rand(n, 8, generator=g).to(dev)
    z = torch.sort(torch.rand(n, S, generator=g) * 4, dim=1).values.to(dev)
    raw = torch.randn(n, S, 4, generator=g).to(dev)
    td = (1.0 + 2 * torch.rand(n, generator=g))
    td[::9] = 0.0
    td = td.to(dev)
    tc = torch.rand(n, 3, generator=g).to(dev)
    cfg = (5.0, 0.1, 1000.0, 10.0, 0.1, 100.0, 0.05)
    st = _lib.stream_ptr(torch.device(dev))
    P = _lib.ptr
    l5 = torch.empty(5, device=dev)
    gm, gr = torch.empty(live, 8, device=dev), \
        torch.empty(live, S, 4, device=dev)
    ws = torch.empty(n * 8, device=dev)
    _lib.check(lib.xrd_coslam_loss(
        live, S, *cfg, P(maps[:live].contiguous()), P(z[:live].contiguous()),
        P(raw[:live].contiguous()), P(td[:live].contiguous()),
        P(tc[:live].contiguous()), P(l5), P(gm), P(gr), P(ws), st), 'loss')
    nl = torch.tensor([live], dtype=torch.int32, device=dev)
    l5b = torch.empty(5, device=dev)
    gmb = torch.full((n, 8), 7.0, device=dev)
    grb = torch.full((n, S, 4), 7.0, device=dev)
    _lib.check(lib.xrd_coslam_loss_live(
        n, S, *cfg, P(maps), P(z), P(raw), P(td), P(tc), P(nl), P(l5b),
        P(gmb), P(grb), P(ws), st), 'loss_live')
    assert torch.allclose(l5b, l5, rtol=1e-6)
    assert torch.allclose(gmb[:live], gm, rtol=1e-6, atol=1e-12)
    assert torch.allclose(grb[:live], gr, rtol=1e-6, atol=1e-12)
    assert float(gmb[live:].abs().max()) == 0.0
    assert float(grb[live:].abs().max()) == 0.0


def test_sample_distinct_dev_equals_host_sized_call():
    from xrdslam_amd.engine import slam_ops
    for total, n in ((15360 * 7, 2048), (307200, 2048), (5000, 5000)):
        torch.manual_seed(3)
        a = slam_ops.sample_distinct(total, n, 'cuda:0')
        torch.manual_seed(3)
        b = slam_ops.sample_distinct_dev(
            torch.tensor([total], dtype=torch.int64, device='cuda:0'), n,
            'cuda:0')
        assert torch.equal(a, b)


def test_coslam_slot_prewarm_leaves_the_run_unchanged():
    """pre-warming the capacity slots (two eager + two captured iterations per
    bucket on the call's data) restores the model, its optimiser state and
    the random streams EXACTLY (checked around the warm-up itself); the
    mapping call that follows therefore starts from the same state and draws
    the same batches as without the warm-up — the maps after the call agree
    up to what the float-atomic order of the table gradient and Adam make of
    it (a chaotic quantity: only its bulk is compared)"""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import cadence, coslam_config
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    bound = [[-3, 3], [-4, 2.5], [-2, 2.5]]
    cam = Camera(fx=150., fy=150., cx=79.5, cy=59.5, width=160, height=120)
    tables, checked = [], []
    for prewarm in (False, True):
        torch.manual_seed(0)
        np.random.seed(0)
        cfg = coslam_config(bound)
        cfg.mapping_first_n_iters = 60
        cfg.tracking_Wedge = cfg.tracking_Hedge = 5
        cfg.mapping_sample = 768
        algo = cfg.setup(camera=cam, device='cuda:0')
        algo.use_graphs = True
        algo.prewarm_slots = prewarm
        inner = algo._prewarm_slots

        def watched(*a, _algo=algo, _inner=inner, **kw):
            def opt_tensors():
                return [v for o in _algo.model_optimizers.optimizers.values()
                        for stt in o.state.values() for v in stt.values()
                        if torch.is_tensor(v)]
            params = [p for grp in _algo.model_optimizers.parameters.values()
                      for p in grp]
            p0 = [p.detach().clone() for p in params]
            s0 = [(v, v.detach().clone()) for v in opt_tensors()]
            r0 = (torch.cuda.get_rng_state('cuda:0'), torch.get_rng_state())
            _inner(*a, **kw)
            assert all(torch.equal(p.detach(), c) for p, c in zip(params, p0))
            known = {id(v) for v, _ in s0}
            assert all(torch.equal(v, c) for v, c in s0)
            # (state created by the warm-up must read as freshly initialised)
            assert all(float(v.abs().max()) == 0 for v in opt_tensors()
                       if id(v) not in known)
            assert torch.equal(r0[0], torch.cuda.get_rng_state('cuda:0'))
            assert torch.equal(r0[1], torch.get_rng_state())
            checked.append(len(_algo._pslots))
        algo._prewarm_slots = watched
        data = SyntheticRoom(bound, H=120, W=160, fx=150., fy=150., cx=79.5,
                             cy=59.5, n_frames=200, device='cuda:0')
        cad = cadence['co-slam']
        slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                              keyframe_every=cad.keyframe_every,
                              pose_device='cuda:0')
        for k in range(6):          # frame 5: the first slot call
            slam.step(k)
        assert len(algo._pslots) == (5 if prewarm else 1)
        tables.append(algo.model.embed_fn.params.detach().clone())
    assert checked == [5]           # the warm-up ran once, in the second run
    a, b = tables
    diff = (a - b).abs()
    assert float(diff.mean()) < 1e-3 * float(a.abs().mean()), \
        (float(diff.mean()), float(a.abs().mean()))
    assert float((diff > 1e-2).float().mean()) < 1e-3


@pytest.mark.parametrize('persistent', [False, True])
def test_coslam_mapping_graph_slot(persistent):
    """Co-SLAM with mapping through the persistent capacity slot (pose stacks,
    bank and current-frame part at fixed addresses, sizes read on the device;
    replay-only calls from the second call of a bucket on) tracks like the
    per-call path; the two draw the same batches (same RNG consumption, the
    live rows are the per-call batch)."""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import cadence, coslam_config
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    bound = [[-3, 3], [-4, 2.5], [-2, 2.5]]
    cam = Camera(fx=150., fy=150., cx=79.5, cy=59.5, width=160, height=120)
    cfg = coslam_config(bound)
    cfg.mapping_first_n_iters = 100
    cfg.tracking_Wedge = cfg.tracking_Hedge = 5
    cfg.mapping_sample = 768
    algo = cfg.setup(camera=cam, device='cuda:0')
    algo.use_graphs = True
    algo.persistent_map = persistent
    data = SyntheticRoom(bound, H=120, W=160, fx=150., fy=150., cx=79.5,
                         cy=59.5, n_frames=200, device='cuda:0')
    cad = cadence['co-slam']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device='cuda:0')
    for k in range(41):
        slam.step(k)
    assert len(algo.keyframe_graph) == 9
    if persistent:
        slots = algo._pslots
        # every bucket is built and captured when the first slot is needed
        # (prewarm_slots): both iteration kinds of all five
        assert set(slots) == {128, 256, 512, 1024, 2048}
        assert all(len(s['graphs']) == 2 for s in slots.values())
    ate = slam.ate_rmse()
    # (float-atomic order makes the ATE of this ~60 cm path vary from run to
    # run: below 2 cm in all but one of the round's ~15 suite runs, 2.11 cm
    # in that one; a pose that never moves scores tens of centimetres)
    assert ate < 0.03, ate
    # bundle adjustment wrote the keyframe poses back
    kf = algo.keyframe_graph[3]
    gt = torch.as_tensor(data[kf.fid]['c2w'])[:3, 3]
    assert (kf.get_pose().detach().cpu()[:3, 3] - gt).norm() < 0.03


def test_tracking_reads_the_decoder_through_a_static_pack():
    """tracking calls (decoder frozen) read the packed decoder from a static
    buffer: it follows every weight change torch can see, and the hook that
    ends a mapping call invalidates it for changes torch cannot see (captured
    optimiser steps, raw-pointer Adam); results equal the per-call pack"""
    from xrdslam_amd.engine import coslam as ec
    g = np.load(cg.GOLDEN)
    model = cg.build_model(g, 'cuda:0')
    tab = model._fused_tables('cuda:0')
    ro = torch.from_numpy(g['rays_o']).cuda()
    rd = torch.from_numpy(g['rays_d']).cuda()
    td = torch.from_numpy(g['target_d']).cuda()
    rnd = torch.rand(ro.shape[0], 43, device='cuda:0')

    def both():
        a = ec.render(model, tab, ro, rd, td, rnd, train_map=False)['_maps']
        b = ec.render(model, tab, ro, rd, td, rnd, train_map=True)['_maps']
        return a.detach(), b.detach()
    a, b = both()
    assert torch.equal(a, b)
    buf = model._track_pack
    w = model.decoder.sdf_net.model[0].weight
    with torch.no_grad():
        w.mul_(1.5)                       # a change torch's counter sees
    a2, b2 = both()
    assert torch.equal(a2, b2) and not torch.equal(a2, a)
    assert model._track_pack is buf       # same static buffer, re-packed
    w.data.mul_(0.5)                      # a change it does not see
    stale, fresh = both()
    assert not torch.equal(stale, fresh)
    model._track_pack_key = None          # what CoSLAM.after_mapping_update does
    a3, b3 = both()
    assert torch.equal(a3, b3)
