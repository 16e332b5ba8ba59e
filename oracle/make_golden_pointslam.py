"""TEST INFRASTRUCTURE ONLY.  Point-SLAM golden vectors: executes the
REFERENCE's own ``ConvOnet2`` / ``NeuralPointCloud`` / ``POINT`` decoders
(imported from /root/reference) on the CPU with oracle/faiss_standin.py (exact
brute-force kNN) and stores: point-cloud growth over two frames, renders in
the geometry and colour stages, tracking and mapping losses, all gradients,
and every random draw (feature initialisation, empty-neighbour features) in
tests/golden/pointslam_render.npz.

    python oracle/make_golden_pointslam.py          # the small case
    python oracle/make_golden_pointslam.py tum      # BASELINE configs[4]
    python oracle/make_golden_pointslam.py tum64    # its f64 referee
    python oracle/make_golden_pointslam.py small64  # f64 referee, small case

``tum``: the shapes of the reference's point-slam configuration
(slam/configs/input_config.py:297-340: 1500 tracking rays, 5000 mapping rays,
5 samples a ray) on a TUM-fr1-like 640x480 camera, with a cloud of > 15 000
neural points grown over two frames.  Inputs and feature draws are regenerated
from seeds on both sides (tests/pointslam_golden_util.tum_inputs); the file
holds the reference's outputs (gradients of the point features as the rows of
a seeded subset + column sums) -> tests/golden/pointslam_tum.npz.

``tum64``: the f64 REFEREE of the ``tum`` case.  The same reference classes,
the same cloud (grown in f32, draw for draw), the same feature draws — but the
three queries (geometry mapping, colour mapping, tracking) are evaluated in
float64: parameters and features promoted, ``torch.set_default_dtype(float64)``
and the reference's hard-coded ``.float()`` / ``dtype=torch.float`` casts
redirected to float64 FOR THE DURATION OF THE QUERIES (patched here, in the
harness; the reference tree is untouched), exact kNN on the f64 sample
positions.  Two f32 evaluations of this path (torch on the CPU = the golden,
the HIP kernels) differ from each other on the rays whose samples sit on a
ReLU kink of the 32-wide decoder or on the query-radius cut; the referee tells
which of the two is closer to the exact value and bounds the kernels' distance
by the reference's own (tests/test_pointslam_hip.py) ->
tests/golden/pointslam_tum_f64.npz (+ the f32 reference's distance to it).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import faiss_standin  # noqa: E402
import ref_harness  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def frame_rays(k, n, g):
    """rays of a camera in front of a wall at z = -2 (OpenGL: looks down -z)"""
    o = torch.tensor([0.1 * k, 0.0, 0.05 * k]).repeat(n, 1)
    d = torch.stack([(torch.rand(n, generator=g) - 0.5) * 1.0,
                     (torch.rand(n, generator=g) - 0.5) * 0.8,
                     -torch.ones(n)], -1)
    depth = (2.0 + 0.3 * d[:, 0] + 0.05 * k) * (1 + 0.01 * torch.randn(
        n, generator=g))
    color = torch.rand(n, 3, generator=g)
    r_add = 0.02 + 0.06 * torch.rand(n, generator=g).double()
    return o.float(), d.float(), depth.float(), color.float(), r_add


def main():
    ref_harness.install()
    sys.modules['faiss'] = faiss_standin.module()
    import slam.model_components.neural_point_cloud as npc_mod
    npc_mod.faiss = sys.modules['faiss']
    from slam.common.camera import Camera
    from slam.models.conv_onet_pointslam import ConvOnet2, ConvOnet2Config
    ConvOnet2.load_pretrain = lambda self: None  # LFS pointer only
    torch.manual_seed(0)
    model = ConvOnet2(ConvOnet2Config(mapping_pixels_based_on_color_grad=40),
                      Camera(40., 40., 31.5, 23.5, 64, 48))
    out = {}
    for k, v in model.decoder.state_dict().items():
        out[f'dec/{k}'] = v.numpy().copy()
    # non-learnable Fourier matrix of the colour decoder is a plain attribute
    out['dec_attr/color_decoder.embedder._B'] = \
        model.decoder.color_decoder.embedder._B.numpy().copy()

    draws = []
    gen = torch.Generator().manual_seed(21)
    real_normal = torch.Tensor.normal_

    def rec_normal(self, mean=0, std=1, **k):
        real_normal(self, mean=mean, std=std, generator=gen)
        draws.append(self.clone())
        return self

    g = torch.Generator().manual_seed(4)
    torch.Tensor.normal_ = rec_normal
    try:
        for k in range(2):
            o, d, depth, color, r_add = frame_rays(k, 150, g)
            o2, d2, depth2, color2, r_add2 = frame_rays(k, 40, g)
            for name, v in (('o', o), ('d', d), ('depth', depth),
                            ('color', color), ('r', r_add), ('o2', o2),
                            ('d2', d2), ('depth2', depth2),
                            ('color2', color2), ('r2', r_add2)):
                out[f'add{k}/{name}'] = v.numpy()
            model.model_update({
                'batch_rays_o': o, 'batch_rays_d': d, 'batch_gt_depth': depth,
                'batch_gt_color': color, 'batch_dynamic_r': r_add,
                'batch_rays_o_grad': o2, 'batch_rays_d_grad': d2,
                'batch_gt_depth_grad': depth2, 'batch_gt_color_grad': color2,
                'batch_dynamic_r_grad': r_add2})
            npc = model.neural_point_cloud
            out[f'add{k}/cloud'] = np.array(npc._cloud_pos, np.float32)
            out[f'add{k}/n_input'] = np.int64(len(npc._input_pos))
        out['geo_feats'] = npc.geo_feats.detach().numpy().copy()
        out['col_feats'] = npc.col_feats.detach().numpy().copy()
        # frustum mask: a third of the points frozen
        fm = torch.ones(npc.pts_num(), dtype=torch.bool)
        fm[::3] = False
        model.masked_indices = fm
        out['frustum_mask'] = fm.numpy()
        model.get_param_groups()  # applies the mask like the mapper does

        o, d, depth, color, r_add = frame_rays(1, 90, g)
        depth[5:12] = 0.0       # pixels without sensor depth
        d[80:, 2] = 1.0         # rays that look away from every point
        rq = 2 * r_add
        out.update({'q/o': o.numpy(), 'q/d': d.numpy(),
                    'q/depth': depth.numpy(), 'q/color': color.numpy(),
                    'q/r': rq.numpy()})
        for tag, stage, is_mapping in (('map_geo', 'geometry', True),
                                       ('map_col', 'color', True),
                                       ('track', 'color', False)):
            for p in model.parameters():
                p.grad = None
            npc.geo_feats.grad = npc.col_feats.grad = None
            ro = o.clone().requires_grad_(True)
            rd = d.clone().requires_grad_(True)
            n0 = len(draws)
            inp = {'rays_o': ro, 'rays_d': rd, 'target_s': color,
                   'target_d': depth.reshape(-1, 1), 'stage': stage,
                   'batch_dynamic_r': rq}
            res = model.get_outputs(inp)
            ld = model.get_loss_dict(res, inp, is_mapping, stage)
            sum(ld.values()).backward()
            out[f'{tag}/n_draws'] = np.int64(len(draws) - n0)
            for k2 in ('rgb', 'depth', 'uncertainty', 'valid_ray_mask'):
                out[f'{tag}/{k2}'] = res[k2].detach().numpy()
            for k2, v in ld.items():
                out[f'{tag}/loss_{k2}'] = v.detach().numpy()
            out[f'{tag}/g_rays_o'] = ro.grad.numpy()
            out[f'{tag}/g_rays_d'] = rd.grad.numpy()
            out[f'{tag}/g_geo'] = npc.geo_feats.grad.numpy().copy()
            if npc.col_feats.grad is not None:
                out[f'{tag}/g_col'] = npc.col_feats.grad.numpy().copy()
            for k2, p in model.decoder.named_parameters():
                if p.grad is not None:
                    out[f'{tag}/g_dec/{k2}'] = p.grad.numpy().copy()
    finally:
        torch.Tensor.normal_ = real_normal
    for i, dr in enumerate(draws):
        out[f'draw{i}'] = dr.numpy()
    out['n_draws'] = np.int64(len(draws))
    path = os.path.join(GOLD, 'pointslam_render.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB; points',
          out['add0/cloud'].shape[0], '->', out['add1/cloud'].shape[0],
          'draws', len(draws), 'valid rays',
          int(out['map_col/valid_ray_mask'].sum()),
          {k2: float(v.detach()) for k2, v in ld.items()})


def main_tum():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import pointslam_golden_util as pg
    ref_harness.install()
    sys.modules['faiss'] = faiss_standin.module()
    import slam.model_components.neural_point_cloud as npc_mod
    npc_mod.faiss = sys.modules['faiss']
    from slam.common.camera import Camera
    from slam.models.conv_onet_pointslam import ConvOnet2, ConvOnet2Config
    ConvOnet2.load_pretrain = lambda self: None  # LFS pointer only
    torch.manual_seed(0)
    model = ConvOnet2(ConvOnet2Config(), Camera(*pg.TUM_CAM))
    out = {}
    for k, v in model.decoder.state_dict().items():
        out[f'dec/{k}'] = v.numpy().copy()
    out['dec_attr/color_decoder.embedder._B'] = \
        model.decoder.color_decoder.embedder._B.numpy().copy()
    gen = torch.Generator().manual_seed(pg.TUM_DRAW_SEED)
    real_normal = torch.Tensor.normal_
    shapes = []

    def rec_normal(self, mean=0, std=1, **k):
        real_normal(self, mean=mean, std=std, generator=gen)
        shapes.append(tuple(self.shape))
        return self

    torch.Tensor.normal_ = rec_normal
    try:
        for k in range(2):
            model.model_update(pg.tum_add_inputs(k))
            npc = model.neural_point_cloud
            out[f'add{k}/n_cloud'] = np.int64(len(npc._cloud_pos))
            out[f'add{k}/n_input'] = np.int64(len(npc._input_pos))
        cloud = np.array(npc._cloud_pos, np.float32)
        out['cloud_sum'] = cloud.astype(np.float64).sum(0)
        out['cloud_rows'] = cloud[pg.subset(cloud.shape[0])]
        fm = pg.tum_frustum_mask(npc.pts_num())
        model.masked_indices = fm
        model.get_param_groups()
        for tag, stage, is_mapping in (('map_geo', 'geometry', True),
                                       ('map_col', 'color', True),
                                       ('track', 'color', False)):
            q = pg.tum_query(is_mapping)
            for p in model.parameters():
                p.grad = None
            npc.geo_feats.grad = npc.col_feats.grad = None
            ro = q['o'].clone().requires_grad_(True)
            rd = q['d'].clone().requires_grad_(True)
            n0 = len(shapes)
            inp = {'rays_o': ro, 'rays_d': rd, 'target_s': q['color'],
                   'target_d': q['depth'].reshape(-1, 1), 'stage': stage,
                   'batch_dynamic_r': q['r']}
            res = model.get_outputs(inp)
            ld = model.get_loss_dict(res, inp, is_mapping, stage)
            sum(ld.values()).backward()
            out[f'{tag}/n_draws'] = np.int64(len(shapes) - n0)
            for k2 in ('rgb', 'depth', 'uncertainty', 'valid_ray_mask'):
                out[f'{tag}/{k2}'] = res[k2].detach().numpy()
            for k2, v in ld.items():
                out[f'{tag}/loss_{k2}'] = v.detach().numpy()
            out[f'{tag}/g_rays_o'] = ro.grad.numpy()
            out[f'{tag}/g_rays_d'] = rd.grad.numpy()
            for name, t in (('g_geo', npc.geo_feats.grad),
                            ('g_col', npc.col_feats.grad)):
                if t is None:
                    continue
                a = t.numpy()
                out[f'{tag}/{name}/rows'] = a[pg.subset(a.shape[0])].copy()
                out[f'{tag}/{name}/colsum'] = a.astype(np.float64).sum(0)
                out[f'{tag}/{name}/abssum'] = np.abs(a.astype(
                    np.float64)).sum(1)[pg.subset(a.shape[0], 7, 4000)]
            for k2, p in model.decoder.named_parameters():
                if p.grad is not None:
                    out[f'{tag}/g_dec/{k2}'] = p.grad.numpy().copy()
    finally:
        torch.Tensor.normal_ = real_normal
    out['draw_shapes'] = np.array([list(sh) + [0] * (2 - len(sh))
                                   for sh in shapes], np.int64)
    path = os.path.join(GOLD, 'pointslam_tum.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB; points',
          int(out['add0/n_cloud']), '->', int(out['add1/n_cloud']), 'draws',
          len(shapes), 'valid rays (map_col)',
          int(out['map_col/valid_ray_mask'].sum()), 'of',
          out['map_col/valid_ray_mask'].shape[0], '(track)',
          int(out['track/valid_ray_mask'].sum()),
          {k2: float(v.detach()) for k2, v in ld.items()})


class _Index64(faiss_standin.IndexIVFFlat):
    """exact kNN on f64 queries, squared distances returned in f64"""

    def search(self, x, k):
        x = np.asarray(x, np.float64).reshape(-1, 3)
        pts = self.pts.astype(np.float64)
        m = x.shape[0]
        D = np.full((m, k), 3.4028235e38, np.float64)
        ids = np.full((m, k), -1, np.int64)
        for a in range(0, m, 2048):
            d2 = ((x[a:a + 2048, None, :] - pts[None, :, :])**2).sum(-1)
            order = np.argsort(d2, axis=1, kind='stable')[:, :k]
            D[a:a + 2048, :order.shape[1]] = np.take_along_axis(d2, order, 1)
            ids[a:a + 2048, :order.shape[1]] = order
        return D, ids


def main_tum64():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import pointslam_golden_util as pg
    ref_harness.install()
    sys.modules['faiss'] = faiss_standin.module()
    import slam.model_components.neural_point_cloud as npc_mod
    npc_mod.faiss = sys.modules['faiss']
    from slam.common.camera import Camera
    from slam.models.conv_onet_pointslam import ConvOnet2, ConvOnet2Config
    ConvOnet2.load_pretrain = lambda self: None  # LFS pointer only
    torch.manual_seed(0)
    model = ConvOnet2(ConvOnet2Config(), Camera(*pg.TUM_CAM))
    gen = torch.Generator().manual_seed(pg.TUM_DRAW_SEED)
    real_normal = torch.Tensor.normal_

    def rec_normal(self, mean=0, std=1, **k):
        # the SAME f32 draw whatever the tensor's dtype
        t = torch.empty(self.shape, dtype=torch.float32)
        real_normal(t, mean=mean, std=std, generator=gen)
        self.copy_(t)
        return self

    g32 = np.load(os.path.join(GOLD, 'pointslam_tum.npz'))
    out = {}
    undo = None
    torch.Tensor.normal_ = rec_normal
    try:
        for k in range(2):          # the cloud: f32, exactly the golden's
            model.model_update(pg.tum_add_inputs(k))
        npc = model.neural_point_cloud
        assert len(npc._cloud_pos) == int(g32['add1/n_cloud'])
        model.masked_indices = pg.tum_frustum_mask(npc.pts_num())
        model.get_param_groups()
        # the queries are drawn under the f32 default dtype (torch.rand under
        # a float64 default draws different numbers)
        queries = {m: pg.tum_query(m) for m in (True, False)}
        # ---- from here on: float64 ------------------------------------
        undo = _enter_f64(model, npc)
        for tag, stage, is_mapping in (('map_geo', 'geometry', True),
                                       ('map_col', 'color', True),
                                       ('track', 'color', False)):
            q = queries[is_mapping]
            for p in model.parameters():
                p.grad = None
            npc.geo_feats.grad = npc.col_feats.grad = None
            ro = q['o'].double().requires_grad_(True)
            rd = q['d'].double().requires_grad_(True)
            inp = {'rays_o': ro, 'rays_d': rd, 'target_s': q['color'].double(),
                   'target_d': q['depth'].double().reshape(-1, 1),
                   'stage': stage, 'batch_dynamic_r': q['r'].double()}
            res = model.get_outputs(inp)
            ld = model.get_loss_dict(res, inp, is_mapping, stage)
            sum(ld.values()).backward()
            assert res['depth'].dtype == torch.float64
            assert ro.grad.dtype == torch.float64
            vm = res['valid_ray_mask'].numpy()
            out[f'{tag}/valid_ray_mask'] = vm
            for k2 in ('rgb', 'depth', 'uncertainty'):
                out[f'{tag}/{k2}'] = res[k2].detach().numpy()
            for k2, v in ld.items():
                out[f'{tag}/loss_{k2}'] = v.detach().numpy()
            out[f'{tag}/g_rays_o'] = ro.grad.numpy()
            out[f'{tag}/g_rays_d'] = rd.grad.numpy()
            for name, t in (('g_geo', npc.geo_feats.grad),
                            ('g_col', npc.col_feats.grad)):
                if t is None or f'{tag}/{name}/rows' not in g32.files:
                    continue
                a = t.numpy()
                assert a.dtype == np.float64
                out[f'{tag}/{name}/rows'] = a[pg.subset(a.shape[0])].copy()
                out[f'{tag}/{name}/colsum'] = a.sum(0)
                out[f'{tag}/{name}/abssum'] = np.abs(a).sum(1)[
                    pg.subset(a.shape[0], 7, 4000)]
            for k2, p in model.decoder.named_parameters():
                if p.grad is not None and f'{tag}/g_dec/{k2}' in g32.files:
                    out[f'{tag}/g_dec/{k2}'] = p.grad.numpy().copy()
    finally:
        torch.Tensor.normal_ = real_normal
        if undo is not None:
            undo()
    # the f32 reference's own distance to the referee, for the record
    for k in sorted(out):
        if k in g32.files and out[k].dtype == np.float64 and out[k].size:
            a, b = np.asarray(g32[k], np.float64), out[k]
            print(f'  f32 reference vs f64: {k:50s} '
                  f'{np.abs(a - b).max() / max(np.abs(b).max(), 1e-30):.3e}')
        elif k in g32.files:
            print(f'  {k}: equal = {bool(np.array_equal(g32[k], out[k]))}')
    path = os.path.join(GOLD, 'pointslam_tum_f64.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def _enter_f64(model, npc):
    """promote the reference model to float64 and redirect its hard-coded
    float32 casts; returns the undo function"""
    real_float, real_f = torch.Tensor.float, torch.float
    torch.set_default_dtype(torch.float64)
    torch.Tensor.float = lambda self: self.double()
    torch.float = torch.float64
    model.double()
    cd = model.decoder.color_decoder
    cd.embedder._B = cd.embedder._B.double()
    npc.double()                # geo_feats / col_feats are Parameters
    assert npc.geo_feats.dtype == torch.float64
    idx64 = _Index64(None, 3, 1)
    idx64.pts, idx64.is_trained = npc.index.pts, True
    npc.index = idx64

    def undo():
        torch.Tensor.float, torch.float = real_float, real_f
        torch.set_default_dtype(torch.float32)
    return undo


def main_small64():
    """f64 referee of the small case (pointslam_render.npz): inputs and draws
    are read back from that file -> tests/golden/pointslam_render_f64.npz"""
    ref_harness.install()
    sys.modules['faiss'] = faiss_standin.module()
    import slam.model_components.neural_point_cloud as npc_mod
    npc_mod.faiss = sys.modules['faiss']
    from slam.common.camera import Camera
    from slam.models.conv_onet_pointslam import ConvOnet2, ConvOnet2Config
    ConvOnet2.load_pretrain = lambda self: None  # LFS pointer only
    g32 = np.load(os.path.join(GOLD, 'pointslam_render.npz'))
    T = lambda k: torch.from_numpy(g32[k])  # noqa: E731
    torch.manual_seed(0)
    model = ConvOnet2(ConvOnet2Config(mapping_pixels_based_on_color_grad=40),
                      Camera(40., 40., 31.5, 23.5, 64, 48))
    for k, v in model.decoder.state_dict().items():
        assert np.array_equal(v.numpy(), g32[f'dec/{k}']), k
    draws = iter([T(f'draw{i}') for i in range(int(g32['n_draws']))])
    real_normal = torch.Tensor.normal_

    def replay_normal(self, mean=0, std=1, **k):
        self.copy_(next(draws))     # the golden's own draws, in call order
        return self

    out = {}
    undo = None
    torch.Tensor.normal_ = replay_normal
    try:
        for k in range(2):
            model.model_update({
                'batch_rays_o': T(f'add{k}/o'), 'batch_rays_d': T(f'add{k}/d'),
                'batch_gt_depth': T(f'add{k}/depth'),
                'batch_gt_color': T(f'add{k}/color'),
                'batch_dynamic_r': T(f'add{k}/r'),
                'batch_rays_o_grad': T(f'add{k}/o2'),
                'batch_rays_d_grad': T(f'add{k}/d2'),
                'batch_gt_depth_grad': T(f'add{k}/depth2'),
                'batch_gt_color_grad': T(f'add{k}/color2'),
                'batch_dynamic_r_grad': T(f'add{k}/r2')})
        npc = model.neural_point_cloud
        assert np.array_equal(np.array(npc._cloud_pos, np.float32),
                              g32['add1/cloud'])
        model.masked_indices = T('frustum_mask')
        model.get_param_groups()
        undo = _enter_f64(model, npc)
        for tag, stage, is_mapping in (('map_geo', 'geometry', True),
                                       ('map_col', 'color', True),
                                       ('track', 'color', False)):
            for p in model.parameters():
                p.grad = None
            npc.geo_feats.grad = npc.col_feats.grad = None
            ro = T('q/o').double().requires_grad_(True)
            rd = T('q/d').double().requires_grad_(True)
            inp = {'rays_o': ro, 'rays_d': rd,
                   'target_s': T('q/color').double(),
                   'target_d': T('q/depth').double().reshape(-1, 1),
                   'stage': stage, 'batch_dynamic_r': T('q/r').double()}
            res = model.get_outputs(inp)
            ld = model.get_loss_dict(res, inp, is_mapping, stage)
            sum(ld.values()).backward()
            assert res['depth'].dtype == torch.float64
            for k2 in ('rgb', 'depth', 'uncertainty', 'valid_ray_mask'):
                out[f'{tag}/{k2}'] = res[k2].detach().numpy()
            for k2, v in ld.items():
                out[f'{tag}/loss_{k2}'] = v.detach().numpy()
            out[f'{tag}/g_rays_o'] = ro.grad.numpy()
            out[f'{tag}/g_rays_d'] = rd.grad.numpy()
            out[f'{tag}/g_geo'] = npc.geo_feats.grad.numpy().copy()
            if npc.col_feats.grad is not None:
                out[f'{tag}/g_col'] = npc.col_feats.grad.numpy().copy()
            for k2, p in model.decoder.named_parameters():
                if p.grad is not None:
                    out[f'{tag}/g_dec/{k2}'] = p.grad.numpy().copy()
    finally:
        torch.Tensor.normal_ = real_normal
        if undo is not None:
            undo()
    worst = {}
    for k in sorted(out):
        if k in g32.files and out[k].dtype == np.float64 and out[k].size:
            a, b = np.asarray(g32[k], np.float64), out[k]
            worst[k] = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    for k in sorted(worst, key=worst.get)[-8:]:
        print(f'  f32 reference vs f64: {k:55s} {worst[k]:.3e}')
    path = os.path.join(GOLD, 'pointslam_render_f64.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'small64':
        main_small64()
    elif len(sys.argv) > 1 and sys.argv[1] == 'tum64':
        main_tum64()
    elif len(sys.argv) > 1 and sys.argv[1] == 'tum':
        main_tum()
    else:
        main()
