"""GPU parity of the fused Vox-Fusion ray pipeline (csrc/vox_rays.hip through
engine/vox.py: octree traversal -> hit sort -> inverse-CDF sampling -> point
compaction -> voxel features + decoder -> compositing + losses, and the
backward of all of it, with static capacities and no host sync):

(1) against the golden made from the REFERENCE's own SparseVoxel
    (tests/golden/voxfusion_render.npz, oracle/make_golden_voxfusion.py): sample
    depths, rendered depth/colour, the four loss terms and every gradient;
(2) against the modular path (get_outputs + get_loss_dict on compat.grid, itself
    pinned to the same golden) on a synthetic room view with 2048 rays and a
    ragged, multi-chunk sample distribution;
(3) capacity handling: a batch that does not fit is reported and the
    capacities grow; (4) the VoxFusion loop through captured graphs."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import parity  # noqa: E402
import voxfusion_golden_util as vg  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _noise_by_ray(noise_rows, hit, s_cap):
    """reference-layout noise (one row per HIT ray, in rank order) -> one row
    per ray of the batch"""
    n = hit.shape[0]
    out = torch.full((n, s_cap), 0.5, device=hit.device)
    rows = noise_rows[:int(hit.sum())]
    w = min(rows.shape[1], s_cap)
    out[hit.bool(), :w] = rows[:, :w].to(hit.device)
    return out


@pytest.mark.parametrize('which', ['small', 'office0'])
def test_fused_iteration_vs_reference_golden(which):
    """'office0' = BASELINE configs[2] shapes: 1024 rays of a 640x480 camera,
    3546 leaf voxels, ragged rows of up to 121 samples (golden made by the
    reference's SparseVoxel, oracle/make_golden_voxfusion.py office0)"""
    from xrdslam_amd.engine import vox as ev
    g = vg.Golden(vg.GOLDEN if which == 'small' else vg.GOLDEN_OFFICE0)
    model = vg.build_model(g, DEV)
    model.insert_points(torch.from_numpy(g['points']).to(DEV))
    noise = torch.from_numpy(g['noise'])           # [G, R, max_steps]
    hit = torch.from_numpy(g['out/ray_mask']).to(DEV)
    rows = noise.reshape(-1, noise.shape[-1])
    model.noise_fn = lambda shape, like: _noise_by_ray(rows, hit, shape[1])
    ro = torch.from_numpy(g['rays_o']).to(DEV).requires_grad_(True)
    rd = torch.from_numpy(g['rays_d']).to(DEV).requires_grad_(True)
    inp = {'rays_o': ro, 'rays_d': rd,
           'target_s': torch.from_numpy(g['target_s']).to(DEV),
           'target_d': torch.from_numpy(g['target_d']).to(DEV)}
    loss, terms = model.fused_loss(inp, True)
    loss.backward()
    ws = model._last_ws
    sizes = model.check_capacity()
    assert not sizes['grown'], sizes
    # structure: hit rays, row length, sample depths
    assert torch.equal(ws.hit.bool(), hit)
    zg = g['out/z_vals']
    assert sizes['s_max'] == zg.shape[1], (sizes, zg.shape)
    z = ws.s_depth[hit][:, :zg.shape[1]].cpu().numpy()
    assert np.abs(z - zg).max() < 1e-5
    names = ['rgb_loss', 'depth_loss', 'sdf_loss', 'fs_loss']
    pairs = [(f'loss/{k}', terms[i], torch.from_numpy(g[f'loss/{k}']))
             for i, k in enumerate(names)]
    pairs += [('depth', ws.depth, torch.from_numpy(g['out/depth'])),
              ('rgb', ws.rgb, torch.from_numpy(g['out/rgb'])),
              ('g_rays_o', ro.grad, torch.from_numpy(g['g_rays_o'])),
              ('g_rays_d', rd.grad, torch.from_numpy(g['g_rays_d'])),
              ('g_embeddings', model.embeddings.grad,
               torch.from_numpy(g['g_embeddings']))]
    for k, p in model.decoder.named_parameters():
        pairs.append((f'g_dec/{k}', p.grad, torch.from_numpy(g[f'g_dec/{k}'])))
    for name, a, b in pairs:
        err = vg.rel_err(a.detach().cpu().numpy().reshape(-1),
                         np.asarray(b).reshape(-1))
        assert err < 1e-4, (name, err)
    # inference entry: same depth / colour, plus the padded weights and z_min
    zmin = torch.zeros(ws.n, device=DEV)
    wts = torch.zeros(ws.n, ws.s_cap, device=DEV)
    with torch.no_grad():
        d2, c2 = ev.render(model.decoder, ws, model.map_states, model.config,
                           ro.detach(), rd.detach(), model.draw_noise(ws),
                           z_min=zmin, weights=wts)
    assert vg.rel_err(d2.cpu().numpy(), g['out/depth']) < 1e-4
    assert vg.rel_err(c2.cpu().numpy(), g['out/rgb']) < 1e-4
    assert vg.rel_err(wts[hit][:, :zg.shape[1]].cpu().numpy(),
                      g['out/weights']) < 1e-4
    assert vg.rel_err(zmin[hit].cpu().numpy(),
                      g['out/z_min'].reshape(-1)) < 1e-5


def _room_model(n_rays, seed=0):
    """a SparseVoxel that has seen one synthetic 160x120 room view"""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import voxfusion_config
    torch.manual_seed(seed)
    np.random.seed(seed)
    cam = Camera(fx=150., fy=150., cx=79.5, cy=59.5, width=160, height=120)
    algo = voxfusion_config().setup(camera=cam, device=DEV)
    data = SyntheticRoom([[-3, 3], [-4, 2.5], [-2, 2.5]], H=120, W=160,
                         fx=150., fy=150., cx=79.5, cy=59.5, n_frames=40,
                         device=DEV)
    from xrdslam_amd.slam.common.frame import Frame
    fr = data[3]
    pose = np.array(fr['c2w'], dtype=np.float32).copy()
    pose[:3, 3] += 10.0   # init_pose_offset: keeps voxel coordinates positive
    depth = fr['depth'].cpu().numpy() if torch.is_tensor(fr['depth']) \
        else np.asarray(fr['depth'])
    rgb = fr['rgb'].cpu().numpy() if torch.is_tensor(fr['rgb']) \
        else np.asarray(fr['rgb'])
    f = Frame(0, rgb, depth, init_pose=pose, device=DEV)
    algo.create_voxels(f)
    algo.config.mapping_sample = n_rays
    with torch.no_grad():
        inp = algo.get_model_input([f], True)
    # train the map a little so that the sdf has sign changes to composite
    return algo, f, {k: (v.detach() if torch.is_tensor(v) else v)
                     for k, v in inp.items()}


def test_fused_iteration_vs_modular_path_room():
    algo, f, inp = _room_model(2048)
    model = algo.model
    with torch.no_grad():
        model.embeddings.mul_(30.0)   # sdf values of both signs
    n = inp['rays_o'].shape[0]
    base = torch.rand(n, 1024, generator=torch.Generator().manual_seed(5))
    base = base.to(DEV)

    def run_fused():
        model.zero_grad(set_to_none=True)
        ro = inp['rays_o'].clone().requires_grad_(True)
        rd = inp['rays_d'].clone().requires_grad_(True)
        model.noise_fn = lambda shape, like: base[:, :shape[1]].contiguous()
        loss, terms = model.fused_loss({**inp, 'rays_o': ro, 'rays_d': rd},
                                       True)
        loss.backward()
        res = {'loss': loss.detach().clone(), 'terms': terms[:4].clone(),
               'g_o': ro.grad.clone(), 'g_d': rd.grad.clone(),
               'g_emb': model.embeddings.grad.clone()}
        for k, p in model.decoder.named_parameters():
            res['g_' + k] = p.grad.clone()
        return res

    got = run_fused()
    ws = model._last_ws
    sizes = model.check_capacity()
    assert not sizes['grown'], sizes
    hit = ws.hit.bool().clone()
    cnt = ws.cnt.clone()
    depth, rgb = ws.depth.clone(), ws.rgb.clone()
    assert int(hit.sum()) > 0.9 * n

    def fed(shape, like):
        G, R, S = shape
        rows = base[hit][:, :S]
        pad = rows[:1].expand(G * R - rows.shape[0], S)
        return torch.cat([rows, pad]).reshape(G, R, S).contiguous()

    model.zero_grad(set_to_none=True)
    model.noise_fn = fed
    ro = inp['rays_o'].clone().requires_grad_(True)
    rd = inp['rays_d'].clone().requires_grad_(True)
    inp2 = {**inp, 'rays_o': ro, 'rays_d': rd}
    out = model.get_outputs(inp2)
    ld = model.get_loss_dict(out, inp2, True, 0)
    sum(ld.values()).backward()
    assert torch.equal(out['ray_mask'], hit)
    # identical sample structure (a row may differ when the float32 sum of
    # its chord lengths, taken in a different order, crosses an integer)
    ref_cnt = out['z_vals'].ne(10.0).sum(-1)   # MAX_DEPTH marks padding
    same = ref_cnt == cnt[hit]
    assert float(same.float().mean()) > 0.998, float(same.float().mean())
    assert sizes['s_max'] == out['z_vals'].shape[1]
    ref = {'loss': sum(ld.values()).detach(),
           'terms': torch.stack([ld['rgb_loss'], ld['depth_loss'],
                                 ld['sdf_loss'], ld['fs_loss']]).detach(),
           'g_o': ro.grad, 'g_d': rd.grad, 'g_emb': model.embeddings.grad}
    for k, p in model.decoder.named_parameters():
        ref['g_' + k] = p.grad
    errs = {k: vg.rel_err(got[k].cpu().numpy(), ref[k].cpu().numpy())
            for k in ref}
    errs['depth'] = vg.rel_err(depth.cpu().numpy(),
                               out['depth'].detach().cpu().numpy())
    errs['rgb'] = vg.rel_err(rgb.cpu().numpy(),
                             out['rgb'].detach().cpu().numpy())
    bad = {k: v for k, v in errs.items() if not v < 1e-4}
    assert not bad, (bad, errs)


def test_capacity_overflow_is_reported_and_grows():
    algo, f, inp = _room_model(512)
    model = algo.model
    model.s_cap, model.pts_per_ray = 64, 8   # far too small
    model.noise_fn = None
    loss, _ = model.fused_loss(inp, False)
    assert torch.isfinite(loss)
    v0 = model.capacity_version
    sizes = model.check_capacity()
    assert sizes['grown'] and model.capacity_version == v0 + 1
    assert model.s_cap >= sizes['row_len'] and model.s_cap % 64 == 0
    loss2, _ = model.fused_loss(inp, False)
    sizes2 = model.check_capacity()
    assert not sizes2['grown'], sizes2
    assert sizes2['n_pts'] <= 512 * model.pts_per_ray
    assert torch.isfinite(loss2)


def test_overflow_of_an_earlier_iteration_is_not_lost():
    """the size record is reset by every launch; its overflow bits are folded
    into sticky slots, so ONE read per optimize_update still sees a batch that
    did not fit in an earlier iteration (advisor finding, round 2)"""
    algo, f, inp = _room_model(512)
    model = algo.model
    model.s_cap, model.pts_per_ray = 64, 8   # far too small
    model.noise_fn = None
    model.fused_loss(inp, False)             # overflows
    # a second iteration on the SAME workspace that fits: rays that leave
    # the mapped view (nothing hit -> no samples)
    away = dict(inp)
    away['rays_d'] = -inp['rays_d']
    model.fused_loss(away, False)
    ws = model._last_ws
    last = ws.meta.tolist()
    assert last[5] == 0 and last[11] != 0, last   # last launch fits, sticky set
    with pytest.warns(UserWarning, match='truncated'):
        sizes = model.check_capacity()
    assert sizes['grown']
    assert model.s_cap >= sizes['row_len'] > 64


def test_deferred_capacity_check_reports_one_call_late():
    """check_capacity_deferred (the frame loop with the pose chain on the
    device): no device->host wait on the call just issued; the record of the
    previous call is evaluated instead, overflow bits of every launch of that
    call included (sticky slots snapshot, then cleared behind the copy)"""
    algo, f, inp = _room_model(512)
    model = algo.model
    model.s_cap, model.pts_per_ray = 64, 8   # far too small
    model.noise_fn = None
    model.fused_loss(inp, False)             # call 1: overflows
    away = dict(inp)
    away['rays_d'] = -inp['rays_d']
    model.fused_loss(away, False)            # ... then a launch that fits
    v0 = model.capacity_version
    assert model.check_capacity_deferred() is None      # nothing to report yet
    assert model.capacity_version == v0
    ws = model._last_ws
    torch.cuda.synchronize()
    assert ws.meta[11:14].tolist() == [0, 0, 0]         # sticky slots restarted
    model.fused_loss(away, False)            # call 2 fits
    with pytest.warns(UserWarning, match='truncated'):
        rec = model.check_capacity_deferred()           # call 1's record
    assert rec['grown'] and model.capacity_version == v0 + 1
    assert model.s_cap >= rec['row_len'] > 64
    loss, _ = model.fused_loss(inp, False)   # call 3 on the grown capacities
    assert model.check_capacity_deferred() is None or True
    sizes = model.check_capacity()
    assert not sizes['grown'] and torch.isfinite(loss)


def test_voxfusion_loop_through_graphs():
    """tracking + mapping of the synthetic room with every iteration inside a
    captured hipGraph (persistent tracking graph, mapping graphs kept across
    calls, map arrays updated in place)"""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import (cadence,
                                                       voxfusion_config)
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    cam = Camera(fx=150., fy=150., cx=79.5, cy=59.5, width=160, height=120)
    algo = voxfusion_config().setup(camera=cam, device=DEV)
    algo.use_graphs = True
    data = SyntheticRoom([[-3, 3], [-4, 2.5], [-2, 2.5]], H=120, W=160,
                         fx=150., fy=150., cx=79.5, cy=59.5, n_frames=200,
                         device=DEV)
    cad = cadence['vox-fusion']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device=DEV,
                          use_relative_pose=cad.use_relative_pose,
                          init_pose_offset=cad.init_pose_offset)
    for k in range(12):
        slam.step(k)
    assert getattr(algo, '_track_slot', None) is not None
    assert algo._track_slot['graph'] is not None
    assert any(s.get('graphs') for s in algo._map_slots.values())
    assert algo.last_batch_sizes['n_hit_rays'] > 0
    ate = slam.ate_rmse()
    assert ate < 0.03, ate
    # render_img: the fused ray pipeline against the modular operators on the
    # same chunks and the same (fixed 0.5) sampler noise
    pose = algo.get_estimate_c2w_list()[10].to(DEV)
    algo.model.noise_fn = None
    rgb_f, d_f = algo.render_img(pose)
    algo.fused_iteration = False
    algo.model.noise_fn = lambda shape, like: like.new_full(shape, 0.5)
    rgb_m, d_m = algo.render_img(pose)
    assert (d_m > 0).mean() > 0.5
    tol_d = 1e-4 * np.abs(d_m).max()
    close = (np.abs(d_f - d_m) < tol_d) & \
        (np.abs(rgb_f - rgb_m).max(-1) < 1e-4)
    # a ray may gain or lose one sample where the float32 sum of its chord
    # lengths is taken in a different order; everything else agrees
    assert close.mean() > 0.999, close.mean()
    assert np.abs(d_f - d_m).max() < 0.05
    # mesh of the mapped room: vertices inside the allocated voxels, colours
    algo.fused_iteration = True
    mesh = algo.get_mesh()
    assert mesh is not None and mesh.faces.shape[0] > 100
    assert mesh.faces.max() < mesh.vertices.shape[0]
    assert mesh.vertex_colors.shape == (mesh.vertices.shape[0], 3)
    ms = algo.model.map_states
    leaf = ~ms['voxel_vertex_idx'].eq(-1).any(-1)
    c = ms['voxel_center_xyz'][leaf].cpu().numpy()
    lo, hi = c.min(0) - 0.1001, c.max(0) + 0.1001
    assert (mesh.vertices >= lo).all() and (mesh.vertices <= hi).all()


def test_sharded_fused_mapping_adds_up():
    """multi-GPU mapping through the fused pipeline (ranks played in turn on
    this GPU), deterministic sharding: every rank draws the SAME batch, passes
    ALL of it to the ray pipeline with a mask of its own slice
    (xrd_vox_sample_rays_shard) — intersection, the sampler's [200, R, P]
    regrouping of the hit rays and the size record (hit rays, longest row,
    free-space / band counts, usable depths = the loss normalisers) are the
    whole batch's on every rank, without an exchange — and evaluates the
    points of its slice.  The per-rank losses and gradients then add up to
    the single-process iteration EXACTLY, 1e-4."""
    from xrdslam_amd.engine import dist as xd
    algo, f, _ = _room_model(1024)
    model = algo.model
    with torch.no_grad():
        model.embeddings.mul_(30.0)
    model.noise_fn = None          # fixed 0.5: the same samples in every run

    def run():
        model.zero_grad(set_to_none=True)
        for p in f.get_params():
            p.grad = None
        torch.manual_seed(11)
        loss = algo.get_loss([f], True)
        loss.backward()
        ws = model._last_ws
        return {'loss': float(loss.detach()),
                'g_emb': model.embeddings.grad.clone(),
                'g_dec': torch.cat([p.grad.reshape(-1)
                                    for p in model.decoder.parameters()]),
                'g_pose': torch.cat([p.grad.reshape(-1)
                                     for p in f.get_params()]),
                'meta': ws.meta.clone(), 'pts': int(ws.meta[4])}

    single = run()
    st = xd.state
    saved = (st.enabled, st.rank, st.world, st.deterministic)
    try:
        for world in (2, 3):
            parts = []
            for r in range(world):
                st.enabled, st.rank, st.world, st.deterministic = \
                    True, r, world, True
                parts.append(run())
            # every rank saw the whole batch's size record ...
            for p in parts:
                for k in (0, 1, 2, 3, 6, 7, 8):
                    assert int(p['meta'][k]) == int(single['meta'][k]), k
            # ... and evaluated only its own points
            assert sum(p['pts'] for p in parts) == single['pts']
            assert max(p['pts'] for p in parts) < 0.7 * single['pts']
            tot = {k: sum(p[k] for p in parts) for k in
                   ('loss', 'g_emb', 'g_dec', 'g_pose')}
            assert abs(tot['loss'] - single['loss']) < \
                1e-4 * abs(single['loss'])
            for k in ('g_emb', 'g_dec', 'g_pose'):
                err = float((tot[k] - single[k]).abs().max() /
                            single[k].abs().max())
                assert err < 1e-4, (world, k, err)
    finally:
        st.enabled, st.rank, st.world, st.deterministic = saved
