"""TEST INFRASTRUCTURE ONLY.  Co-SLAM golden vectors: executes the REFERENCE's
own ``JointEncoding`` (slam/models/joint_encoding.py, imported from
/root/reference through ref_harness) on the CPU with the oracle encodings
(oracle/tcnn_oracle.py) standing in for the unvendored tiny-cuda-nn, and stores
inputs, recorded random draws, outputs, loss terms and gradients in
tests/golden/coslam_render.npz.

    python oracle/make_golden_coslam.py            # coslam_render.npz
    python oracle/make_golden_coslam.py variants   # coslam_variants.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ref_harness  # noqa: E402
import tcnn_standin  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def main():
    ref_harness.install()
    tcnn_mod = tcnn_standin.module()
    sys.modules['tinycudann'] = tcnn_mod
    import slam.model_components.encodings_coslam as enc
    enc.tcnn = tcnn_mod
    from slam.common.camera import Camera
    from slam.models.joint_encoding import JointEncoding, JointEncodingConfig

    torch.manual_seed(0)
    bb = torch.from_numpy(np.array([[-1.0, 1.1], [-1.2, 0.9], [-0.8, 1.0]]))
    cfg = JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True,
                              hashsize=12, trainging_smooth_pts=8)
    model = JointEncoding(cfg, Camera(40., 40., 31.5, 23.5, 64, 48), bb)
    with torch.no_grad():  # non-trivial table and weights
        model.embed_fn.params.copy_(torch.randn_like(model.embed_fn.params) *
                                    0.05)
    out = {'bound': bb.numpy(), 'hash_params': model.embed_fn.params.detach()
           .numpy().copy(),
           'hash_cfg': np.array([model.resolution_sdf, cfg.hashsize],
                                dtype=np.int64)}
    for k, v in model.decoder.state_dict().items():
        out[f'dec/{k}'] = v.numpy().copy()

    g = torch.Generator().manual_seed(3)
    n = 80
    rays_o = (torch.rand(n, 3, generator=g) - 0.5) * 0.4
    rays_d = torch.randn(n, 3, generator=g)
    rays_d = rays_d / rays_d.norm(dim=1, keepdim=True)
    depth = 0.3 + 0.8 * torch.rand(n, 1, generator=g)
    depth[torch.rand(n, 1, generator=g) < 0.12] = 0.0
    color = torch.rand(n, 3, generator=g)
    out.update(rays_o=rays_o.numpy(), rays_d=rays_d.numpy(),
               target_d=depth.numpy(), target_s=color.numpy())

    real_rand = torch.rand
    for tag, is_mapping, first in (('track', False, False),
                                   ('map', True, False),
                                   ('map_first', True, True)):
        draws = []
        gen = torch.Generator().manual_seed(11)

        def rec_rand(*shape, **kw):
            shp = shape[0] if len(shape) == 1 and not isinstance(
                shape[0], int) else shape
            t = real_rand(tuple(shp), generator=gen)
            draws.append(t.clone())
            return t

        torch.rand = rec_rand
        try:
            for p in model.parameters():
                p.grad = None
            ro = rays_o.clone().requires_grad_(True)
            rd = rays_d.clone().requires_grad_(True)
            inp = {'rays_o': ro, 'rays_d': rd, 'target_s': color,
                   'target_d': depth, 'first': first}
            res = model.get_outputs(inp)
            ld = model.get_loss_dict(res, inp, is_mapping, 0)
            loss = sum(ld.values())
            loss.backward()
        finally:
            torch.rand = real_rand
        for i, d in enumerate(draws):
            out[f'{tag}/rand{i}'] = d.numpy()
        out[f'{tag}/n_rand'] = np.int64(len(draws))
        for k in ('rgb', 'depth', 'depth_var', 'acc_map', 'z_vals', 'raw'):
            out[f'{tag}/{k}'] = res[k].detach().numpy()
        for k, v in ld.items():
            out[f'{tag}/loss_{k}'] = v.detach().numpy()
        out[f'{tag}/g_rays_o'] = ro.grad.numpy()
        out[f'{tag}/g_rays_d'] = rd.grad.numpy()
        out[f'{tag}/g_hash'] = model.embed_fn.params.grad.numpy().copy()
        for k, p in model.decoder.named_parameters():
            out[f'{tag}/g_dec/{k}'] = p.grad.numpy().copy()
    path = os.path.join(GOLD, 'coslam_render.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def main_variants():
    """non-default model options of the reference (JointEncodingConfig:
    oneGrid=False -> a second, colour-only hash grid and ColorSDFNet;
    training_n_importance>0 -> the second, inverse-CDF sampling pass), stored
    in tests/golden/coslam_variants.npz"""
    ref_harness.install()
    tcnn_mod = tcnn_standin.module()
    sys.modules['tinycudann'] = tcnn_mod
    import slam.model_components.encodings_coslam as enc
    enc.tcnn = tcnn_mod
    from slam.common.camera import Camera
    from slam.models.joint_encoding import JointEncoding, JointEncodingConfig

    bb = torch.from_numpy(np.array([[-1.0, 1.1], [-1.2, 0.9], [-0.8, 1.0]]))
    out = {'bound': bb.numpy()}
    g = torch.Generator().manual_seed(5)
    n = 64
    rays_o = (torch.rand(n, 3, generator=g) - 0.5) * 0.4
    rays_d = torch.randn(n, 3, generator=g)
    rays_d = rays_d / rays_d.norm(dim=1, keepdim=True)
    depth = 0.3 + 0.8 * torch.rand(n, 1, generator=g)
    depth[torch.rand(n, 1, generator=g) < 0.12] = 0.0
    color = torch.rand(n, 3, generator=g)
    out.update(rays_o=rays_o.numpy(), rays_d=rays_d.numpy(),
               target_d=depth.numpy(), target_s=color.numpy())
    real_rand = torch.rand
    import coslam_golden_util as cg
    for name, kw in cg.VARIANTS.items():
        torch.manual_seed(1)
        cfg = JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True,
                                  hashsize=10, trainging_smooth_pts=8, **kw)
        model = JointEncoding(cfg, Camera(40., 40., 31.5, 23.5, 64, 48), bb)
        grids = {'embed_fn': model.embed_fn}
        if not cfg.oneGrid:
            grids['embed_fn_color'] = model.embed_fn_color
        # tables and decoder weights come from seeds, the same code on the
        # test side (tests/coslam_golden_util.py::variant_state)
        cg.variant_state(model, grids)
        for tag, is_mapping in (('track', False), ('map', True)):
            draws = []
            gen = torch.Generator().manual_seed(13)

            def rec_rand(*shape, **kw_):
                shp = shape[0] if len(shape) == 1 and not isinstance(
                    shape[0], int) else shape
                t = real_rand(tuple(shp), generator=gen)
                draws.append(t.clone())
                return t

            torch.rand = rec_rand
            try:
                for p in model.parameters():
                    p.grad = None
                ro = rays_o.clone().requires_grad_(True)
                rd = rays_d.clone().requires_grad_(True)
                inp = {'rays_o': ro, 'rays_d': rd, 'target_s': color,
                       'target_d': depth, 'first': False}
                res = model.get_outputs(inp)
                ld = model.get_loss_dict(res, inp, is_mapping, 0)
                sum(ld.values()).backward()
            finally:
                torch.rand = real_rand
            pre = f'{name}/{tag}'
            for i, d in enumerate(draws):
                out[f'{pre}/rand{i}'] = d.numpy()
            out[f'{pre}/n_rand'] = np.int64(len(draws))
            for k in ('rgb', 'depth', 'depth_var', 'acc_map', 'z_vals', 'raw'):
                out[f'{pre}/{k}'] = res[k].detach().numpy()
            for k, v in ld.items():
                out[f'{pre}/loss_{k}'] = v.detach().numpy()
            out[f'{pre}/g_rays_o'] = ro.grad.numpy()
            out[f'{pre}/g_rays_d'] = rd.grad.numpy()
            for gname, grid in grids.items():
                out[f'{pre}/g_{gname}'] = grid.params.grad.numpy().copy()
            for k, p in model.decoder.named_parameters():
                out[f'{pre}/g_dec/{k}'] = p.grad.numpy().copy()
    path = os.path.join(GOLD, 'coslam_variants.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'variants':
        main_variants()
    else:
        main()
