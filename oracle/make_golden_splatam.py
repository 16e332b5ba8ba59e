"""TEST INFRASTRUCTURE ONLY.  SplaTAM golden vectors: executes the REFERENCE's
own ``GaussianSplatting`` model / ``GaussianCloud`` (imported from
/root/reference) on the CPU with oracle/gs_standin.py as the rasteriser, on a
tiny two-frame scene, and stores every stage in tests/golden/splatam_render.npz:
seeded parameters, tracking loss + pose gradient, growth on the second frame,
mapping loss + Gaussian gradients, pruning.

    python oracle/make_golden_splatam.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import gs_standin  # noqa: E402
import ref_harness  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
H, W, FX = 24, 32, 30.0


def scene(seed=0):
    """two RGB-D frames of a slanted plane, camera looking down +z"""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(),
                            indexing='ij')
    frames = []
    for k in range(2):
        depth = 1.5 + 0.01 * xs + 0.005 * ys + 0.05 * k
        depth[:3, :4] = 0.0  # invalid pixels
        rgb = torch.stack([xs / W, ys / H, 0.5 + 0.2 * torch.sin(xs / 3 + k)],
                          -1) + 0.02 * torch.rand(H, W, 3, generator=g)
        c2w = torch.eye(4)
        c2w[0, 3] = 0.15 * k  # the second view sees a new strip
        c2w[2, 3] = -0.02 * k
        frames.append((rgb.float().numpy(), depth.float().numpy(),
                       c2w.numpy()))
    return frames


class F:  # minimal Frame stand-in with the attributes the model reads
    def __init__(self, rgb, depth, c2w):
        self.rgb, self.depth = rgb, depth
        self._pose = torch.from_numpy(c2w).float()

    def get_pose(self):
        return self._pose


def main():
    ref_harness.install()
    sys.modules['diff_gaussian_rasterization'] = gs_standin.module()
    # the reference allocates with device='cuda' in several helpers
    for name in ('zeros', 'ones', 'zeros_like', 'ones_like', 'tensor'):
        real = getattr(torch, name)

        def wrap(*a, _real=real, **k):
            if str(k.get('device', '')).startswith('cuda'):
                k['device'] = 'cpu'
            return _real(*a, **k)

        setattr(torch, name, wrap)
    from slam.common.camera import Camera
    from slam.models.gaussian_splatting import (GaussianSplatting,
                                                GaussianSplattingConfig)
    cam = Camera(FX, FX, (W - 1) / 2, (H - 1) / 2, W, H)
    model = GaussianSplatting(GaussianSplattingConfig(), cam, None)
    f0, f1 = (F(*x) for x in scene())
    out = {'cam': np.array([FX, FX, (W - 1) / 2, (H - 1) / 2, W, H])}
    for i, f in enumerate((f0, f1)):
        out[f'f{i}/rgb'], out[f'f{i}/depth'] = f.rgb, f.depth
        out[f'f{i}/c2w'] = f._pose.numpy()
    # 1. seeding
    model.model_update(f0)
    gc = model.gaussian_cloud
    for k, v in gc.params.items():
        out[f'init/{k}'] = v.detach().numpy().copy()
    out['init/scene_radius'] = np.float32(gc.variables['scene_radius'])
    # perturb so that renders are not a perfect fit
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        gc.params['means3D'] += 0.004 * torch.randn(gc.params['means3D'].shape,
                                                    generator=g)
        gc.params['rgb_colors'] += 0.05 * torch.randn(
            gc.params['rgb_colors'].shape, generator=g)
        gc.params['logit_opacities'] += torch.randn(
            gc.params['logit_opacities'].shape, generator=g)
        gc.params['logit_opacities'][::7] = -6.0  # prunable
    for k, v in gc.params.items():
        out[f'pert/{k}'] = v.detach().numpy().copy()
    # 2. tracking loss of frame 1 from a slightly wrong pose
    c2w = f1._pose.clone()
    c2w[0, 3] += 0.01
    c2w.requires_grad_(True)
    inp = {'w2c': torch.inverse(c2w), 'target_s': f1.rgb, 'target_d': f1.depth,
           'is_mapping': False, 'retain_grad': True}
    res = model.get_outputs(inp)
    ld = model.get_loss_dict(res, inp, False)
    sum(ld.values()).backward()
    out['track/c2w'] = c2w.detach().numpy()
    out['track/rgb'] = res['rgb'].detach().numpy()
    out['track/depth_sil'] = res['depth_sil'].detach().numpy()
    out['track/loss_depth'] = ld['depth'].detach().numpy()
    out['track/loss_rgb'] = ld['rgb'].detach().numpy()
    out['track/g_c2w'] = c2w.grad.numpy().copy()
    # 3. growth on frame 1
    n0 = gc.params['means3D'].shape[0]
    model.model_update(f1)
    out['grow/n_before'], out['grow/n_after'] = np.int64(n0), np.int64(
        gc.params['means3D'].shape[0])
    for k, v in gc.params.items():
        out[f'grow/{k}'] = v.detach().numpy().copy()
    # 4. mapping loss on frame 1
    for v in gc.params.values():
        v.grad = None
    inp = {'w2c': torch.inverse(f1._pose), 'target_s': f1.rgb,
           'target_d': f1.depth, 'is_mapping': True, 'retain_grad': True}
    res = model.get_outputs(inp)
    ld = model.get_loss_dict(res, inp, True)
    sum(ld.values()).backward()
    out['map/rgb'] = res['rgb'].detach().numpy()
    out['map/depth_sil'] = res['depth_sil'].detach().numpy()
    out['map/loss_depth'] = ld['depth'].detach().numpy()
    out['map/loss_rgb'] = ld['rgb'].detach().numpy()
    for k, v in gc.params.items():
        out[f'map/g_{k}'] = v.grad.numpy().copy()
    # 5. pruning at iteration 0 through real Adam optimisers
    opt = {k: torch.optim.Adam([v], lr=1e-3) for k, v in gc.params.items()}
    for o in opt.values():
        o.step()
    for k, v in gc.params.items():
        out[f'step/{k}'] = v.detach().numpy().copy()
    model.post_processing(0, opt)
    out['prune/n_after'] = np.int64(gc.params['means3D'].shape[0])
    for k, v in gc.params.items():
        out[f'prune/{k}'] = v.detach().numpy().copy()
    # reference behaviour: remove_points re-keys the sliced moments to the OLD
    # parameter object before swapping in the new one, so the pruned
    # parameter starts with an empty Adam state
    out['prune/new_param_state_len'] = np.int64(
        len(opt['means3D'].state[gc.params['means3D']]))
    path = os.path.join(GOLD, 'splatam_render.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB; gaussians',
          n0, '->', int(out['grow/n_after']), '->', int(out['prune/n_after']),
          {k: float(v.detach()) for k, v in ld.items()})


if __name__ == '__main__':
    main()
