"""Drives the host mirror of SplaTAM's GaussianSplatting / GaussianCloud
through the stages of tests/golden/splatam_render.npz (made by
oracle/make_golden_splatam.py from the reference's own model)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden',
                      'splatam_render.npz')
KEYS = ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities',
        'log_scales')


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def se3_tangent(c2w, g):
    """gradient of a rigid 4x4 projected on the pose's tangent space
    (rotation: skew part of R^T G_R; translation: G_t).  Two extensions of a
    function off SE(3) — torch.inverse of the full matrix vs the rigid
    inverse [R^T | -R^T t] — have different raw 4x4 gradients and the same
    projection; the pose parameters only ever see the projection."""
    c2w = np.asarray(c2w, np.float64)
    g = np.asarray(g, np.float64)
    A = c2w[:3, :3].T @ g[:3, :3]
    S = A - A.T
    return np.concatenate([[S[2, 1], S[0, 2], S[1, 0]], g[:3, 3]])


class Frame:
    def __init__(self, g, i, device):
        self.rgb, self.depth = g[f'f{i}/rgb'], g[f'f{i}/depth']
        self._pose = torch.from_numpy(g[f'f{i}/c2w']).float().to(device)

    def get_pose(self):
        return self._pose


def run(g, device, make_adam, c2w_input=False):
    """-> dict of relative errors / exact flags for every stage.
    ``c2w_input``: hand the model the frame's c2w (rigid inverse taken in
    the preparation kernel) instead of torch.inverse(c2w), and the frame
    object (device-resident targets) like SplaTAM.get_model_input does"""
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.models.gaussian_splatting import (
        GaussianSplatting, GaussianSplattingConfig)
    fx, fy, cx, cy, W, H = g['cam']
    cam = Camera(float(fx), float(fy), float(cx), float(cy), int(W), int(H))
    model = GaussianSplatting(GaussianSplattingConfig(), cam, None).to(device)
    f0, f1 = Frame(g, 0, device), Frame(g, 1, device)
    errs = {}
    model.model_update(f0)
    gc = model.gaussian_cloud
    for k in KEYS:
        errs[f'init/{k}'] = rel_err(gc.params[k].detach().cpu(),
                                    g[f'init/{k}'])
    errs['init/scene_radius'] = rel_err(
        float(gc.variables['scene_radius']), g['init/scene_radius'])
    with torch.no_grad():
        for k in KEYS:
            gc.params[k].copy_(torch.from_numpy(g[f'pert/{k}']))
    # tracking
    c2w = torch.from_numpy(g['track/c2w']).to(device).requires_grad_(True)
    inp = {'target_s': f1.rgb, 'target_d': f1.depth,
           'is_mapping': False, 'retain_grad': True}
    if c2w_input:
        inp['c2w'] = c2w
    else:
        inp['w2c'] = torch.inverse(c2w)
    res = model.get_outputs(inp)
    ld = model.get_loss_dict(res, inp, False)
    sum(ld.values()).backward()
    errs['track/rgb'] = rel_err(res['rgb'].detach().cpu(), g['track/rgb'])
    errs['track/depth_sil'] = rel_err(res['depth_sil'].detach().cpu(),
                                      g['track/depth_sil'])
    errs['track/loss_depth'] = rel_err(ld['depth'].detach().cpu(),
                                       g['track/loss_depth'])
    errs['track/loss_rgb'] = rel_err(ld['rgb'].detach().cpu(),
                                     g['track/loss_rgb'])
    if c2w_input:
        errs['track/g_c2w'] = rel_err(
            se3_tangent(g['track/c2w'], c2w.grad.cpu()),
            se3_tangent(g['track/c2w'], g['track/g_c2w']))
    else:
        errs['track/g_c2w'] = rel_err(c2w.grad.cpu(), g['track/g_c2w'])
    # growth
    model.model_update(f1)
    errs['grow/count'] = abs(gc.params['means3D'].shape[0] -
                             int(g['grow/n_after']))
    if errs['grow/count'] == 0:
        for k in KEYS:
            errs[f'grow/{k}'] = rel_err(gc.params[k].detach().cpu(),
                                        g[f'grow/{k}'])
    # mapping
    inp = {'target_s': f1.rgb, 'target_d': f1.depth, 'is_mapping': True,
           'retain_grad': True}
    if c2w_input:
        inp['c2w'] = f1.get_pose()
    else:
        inp['w2c'] = torch.inverse(f1.get_pose())
    res = model.get_outputs(inp)
    ld = model.get_loss_dict(res, inp, True)
    sum(ld.values()).backward()
    errs['map/rgb'] = rel_err(res['rgb'].detach().cpu(), g['map/rgb'])
    errs['map/depth_sil'] = rel_err(res['depth_sil'].detach().cpu(),
                                    g['map/depth_sil'])
    errs['map/loss_depth'] = rel_err(ld['depth'].detach().cpu(),
                                     g['map/loss_depth'])
    errs['map/loss_rgb'] = rel_err(ld['rgb'].detach().cpu(),
                                   g['map/loss_rgb'])
    for k in KEYS:
        gold = g[f'map/g_{k}']
        got = gc.params[k].grad.cpu().numpy()
        if np.abs(gold).max() < 1e-7:
            # isotropic Gaussians: the rotation gradient is zero up to
            # rounding — compare absolutely
            errs[f'map/g_{k}'] = float(np.abs(got - gold).max()) * 1e-2
        else:
            errs[f'map/g_{k}'] = rel_err(got, gold)
    # one Adam step + pruning
    opt = {k: make_adam([v]) for k, v in gc.params.items()}
    for o in opt.values():
        o.step()
    for k in KEYS:
        errs[f'step/{k}'] = rel_err(gc.params[k].detach().cpu(),
                                    g[f'step/{k}'])
    model.post_processing(0, opt)
    errs['prune/count'] = abs(gc.params['means3D'].shape[0] -
                              int(g['prune/n_after']))
    if errs['prune/count'] == 0:
        for k in KEYS:
            errs[f'prune/{k}'] = rel_err(gc.params[k].detach().cpu(),
                                         g[f'prune/{k}'])
    errs['prune/state'] = abs(len(opt['means3D'].state[gc.params['means3D']])
                              - int(g['prune/new_param_state_len']))
    return errs
