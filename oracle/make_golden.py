"""TEST INFRASTRUCTURE ONLY.  Generates ``tests/golden/*.npz`` by executing the
REFERENCE's own modules (imported from /root/reference through
``oracle/ref_harness.py``) on seeded CPU inputs.  Runs only in the build
container (the reference tree does not exist on the GPU box); the produced
fixtures are committed.

    python oracle/make_golden.py [nice|coslam|vox|splatam|point|all]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def _np(t):
    return t.detach().cpu().numpy()


def _flat_sd(prefix, sd):
    return {f'{prefix}/{k}': _np(v) for k, v in sd.items()}


# ---------------------------------------------------------------------------
# NICE-SLAM
# ---------------------------------------------------------------------------
def synth_rays(n, bound, seed, H=48, W=64, fx=40., fy=40.):
    """Seeded rays from a pose inside ``bound`` + synthetic depth/colour."""
    g = torch.Generator().manual_seed(seed)
    ang = 0.3
    c2w = torch.tensor([[np.cos(ang), 0, np.sin(ang), 0.1],
                        [0, 1, 0, -0.05],
                        [-np.sin(ang), 0, np.cos(ang), 0.2],
                        [0, 0, 0, 1]], dtype=torch.float32)
    i = torch.randint(0, W, (n, ), generator=g).float()
    j = torch.randint(0, H, (n, ), generator=g).float()
    depth = 0.4 + 1.2 * torch.rand(n, 1, generator=g)
    depth[torch.rand(n, 1, generator=g) < 0.1] = 0.0  # invalid pixels
    color = torch.rand(n, 3, generator=g)
    return c2w, i, j, depth, color, (H, W, fx, fy, (W - 1) / 2, (H - 1) / 2)


def make_nice():
    ref_harness.install()
    from slam.common.camera import Camera
    from slam.common.common import get_rays_from_uv
    from slam.models.conv_onet import ConvOnet, ConvOnetConfig

    ConvOnet.load_pretrain = lambda self: None  # LFS pointers only
    torch.manual_seed(0)
    bb = torch.from_numpy(np.array([[-1.0, 1.1], [-1.2, 0.9], [-0.8, 1.0]]))
    n = 96
    c2w, i, j, depth, color, (H, W, fx, fy, cx, cy) = synth_rays(n, bb, 1)
    cam = Camera(fx, fy, cx, cy, W, H)
    model = ConvOnet(ConvOnetConfig(coarse=True), cam, bb)
    # NICE.forward hard-codes 'cuda:%d' (decoder_nice.py:388): on CPU call the
    # sub-decoders exactly as :389-414 does.
    dec = model.decoder

    def nice_forward_cpu(p, c_grid, stage='middle', **kw):
        if stage == 'coarse':
            occ = dec.coarse_decoder(p, c_grid).squeeze(0)
            raw = torch.zeros(occ.shape[0], 4)
            raw[..., -1] = occ
            return raw
        if stage == 'middle':
            occ = dec.middle_decoder(p, c_grid).squeeze(0)
            raw = torch.zeros(occ.shape[0], 4)
            raw[..., -1] = occ
            return raw
        if stage == 'fine':
            fine_occ = dec.fine_decoder(p, c_grid)
            raw = torch.zeros(fine_occ.shape[0], 4)
            middle_occ = dec.middle_decoder(p, c_grid).squeeze(0)
            raw[..., -1] = fine_occ + middle_occ
            return raw
        fine_occ = dec.fine_decoder(p, c_grid)
        raw = dec.color_decoder(p, c_grid)
        middle_occ = dec.middle_decoder(p, c_grid).squeeze(0)
        raw[..., -1] = fine_occ + middle_occ
        return raw

    dec.forward = nice_forward_cpu
    # give the grids a non-trivial scale so lookups matter
    for k in model.grid_c:
        model.grid_c[k] = (model.grid_c[k] * (50.0 if k == 'grid_fine' else
                                              5.0)).requires_grad_(True)

    out = {'bound': _np(model.bounding_box), 'c2w': _np(c2w), 'i': _np(i),
           'j': _np(j), 'gt_depth': _np(depth), 'gt_color': _np(color),
           'cam': np.array([fx, fy, cx, cy, W, H], dtype=np.float64)}
    for k, v in model.grid_c.items():
        out[k] = _np(v)
    for name in ('coarse', 'middle', 'fine', 'color'):
        out.update(_flat_sd(f'dec_{name}',
                            getattr(dec, f'{name}_decoder').state_dict()))

    c2w_p = c2w.clone().requires_grad_(True)
    for stage in ('coarse', 'middle', 'fine', 'color'):
        for is_mapping in (True, False):
            if not is_mapping and stage != 'color':
                continue
            for p in model.parameters():
                p.grad = None
            for k in model.grid_c:
                model.grid_c[k].grad = None
            c2w_p.grad = None
            rays_o, rays_d = get_rays_from_uv(i, j, c2w_p, fx, fy, cx, cy,
                                              'cpu')
            rays_o = rays_o.float()
            rays_d = rays_d.float()
            rays_o.retain_grad()
            rays_d.retain_grad()
            inp = {'rays_o': rays_o, 'rays_d': rays_d, 'target_s': color,
                   'target_d': depth, 'stage': stage}
            res = model.get_outputs(inp)
            ld = model.get_loss_dict(res, inp, is_mapping, stage)
            loss = sum(ld.values())
            loss.backward()
            tag = f'{stage}_{"map" if is_mapping else "track"}'
            out[f'{tag}/depth'] = _np(res['depth'])
            out[f'{tag}/uncertainty'] = _np(res['uncertainty'])
            out[f'{tag}/rgb'] = _np(res['rgb'])
            out[f'{tag}/loss'] = _np(loss)
            out[f'{tag}/g_rays_o'] = _np(rays_o.grad)
            out[f'{tag}/g_rays_d'] = _np(rays_d.grad)
            out[f'{tag}/g_c2w'] = _np(c2w_p.grad)
            for k in model.grid_c:
                g = model.grid_c[k].grad
                if g is not None:
                    out[f'{tag}/g_{k}'] = _np(g)
            for name in ('coarse', 'middle', 'fine', 'color'):
                for pn, p in getattr(dec,
                                     f'{name}_decoder').named_parameters():
                    if p.grad is not None and p.grad.abs().sum() > 0:
                        out[f'{tag}/g_dec_{name}/{pn}'] = _np(p.grad)
    # no-depth render (render_img without gt depth)
    with torch.no_grad():
        rays_o, rays_d = get_rays_from_uv(i, j, c2w, fx, fy, cx, cy, 'cpu')
        res = model.get_outputs({'rays_o': rays_o.float(),
                                 'rays_d': rays_d.float(), 'target_s': None,
                                 'target_d': None, 'stage': 'color'})
        out['color_nodepth/depth'] = _np(res['depth'])
        out['color_nodepth/rgb'] = _np(res['rgb'])
        out['color_nodepth/uncertainty'] = _np(res['uncertainty'])
    path = os.path.join(GOLD, 'nice_render.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


# ---------------------------------------------------------------------------
# Vox-Fusion octree: golden from the COMPILED reference (oracle/_ref)
# ---------------------------------------------------------------------------
def _octree_scenario(which):
    """voxel batches for one scenario (seeded)"""
    rng = np.random.default_rng(100 + which)
    if which == 0:
        # clustered random voxels, three batches with duplicates
        base = rng.integers(40, 90, size=(2000, 3))
        b0 = base[:900]
        b1 = np.concatenate([base[600:1500], base[:50]])
        b2 = base[1200:]
        return [b0, b1, b2]
    # a wall seen by a 'camera': many duplicate voxels per batch + offset 10 m
    u, v = np.meshgrid(np.arange(0, 64), np.arange(0, 48))
    pts = np.stack([50 + (u * 0.13).astype(int), 50 + (v * 0.17).astype(int),
                    np.full_like(u, 62) + (u // 40)], -1).reshape(-1, 3)
    return [pts, pts[::-1].copy(), pts + np.array([3, 0, 1])]


def _run_octree_scenario(which, out_path):
    import build_ref_octree
    Octree = build_ref_octree.load()
    tree = Octree()
    tree.init(256, 16, 0.2)
    out = {}
    batches = _octree_scenario(which)
    rng = np.random.default_rng(7 + which)
    for bi, b in enumerate(batches):
        bt = torch.from_numpy(b.astype(np.int32)).contiguous()
        out[f'batch{bi}'] = b.astype(np.int32)
        out[f'try{bi}'] = np.float64(tree.try_insert(bt))
        tree.insert(bt)
        vox, ch, ft = tree.get_centres_and_children()
        out[f'voxels{bi}'] = vox.numpy()
        out[f'children{bi}'] = ch.numpy()
        out[f'features{bi}'] = ft.numpy()
        out[f'count{bi}'] = np.int64(tree.count_nodes())
        out[f'leaves{bi}'] = np.int64(tree.count_leaf_nodes())
    q = np.concatenate([batches[0][:20] + rng.integers(-1, 2, size=(20, 3)),
                        rng.integers(0, 255, size=(20, 3))]).astype(np.int32)
    out['query'] = q
    out['has'] = np.array([tree.has_voxel(torch.from_numpy(x)) for x in q])
    out['get_voxels'] = tree.get_voxels().numpy()
    out['get_leaf_voxels'] = tree.get_leaf_voxels().numpy()
    np.savez_compressed(out_path, **out)


def make_octree():
    import subprocess
    for which in (0, 1):
        path = os.path.join(GOLD, f'octree_{which}.npz')
        # one process per scenario: node ids come from a process-global counter
        subprocess.check_call([sys.executable, os.path.abspath(__file__),
                               '_octree_scenario', str(which), path])
        print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


MAKERS = {'nice': make_nice, 'octree': make_octree}

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '_octree_scenario':
        _run_octree_scenario(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    os.makedirs(GOLD, exist_ok=True)
    for k, fn in MAKERS.items():
        if which in ('all', k):
            fn()
