"""SplaTAM helpers (reference: slam/model_components/slam_helpers_splatam.py,
slam_external_splatam.py): transform of the isotropic Gaussians into a frame,
the two render-variable dictionaries (colour pass; depth / silhouette / depth^2
pass), SSIM, rotation matrices and the densification statistics."""
from math import exp

import torch
import torch.nn.functional as F


def l1_loss_v1(x, y):
    return torch.abs(x - y).mean()


def build_rotation(q):
    """unit-normalised quaternion (r,x,y,z) -> [n,3,3]"""
    q = q / torch.sqrt((q * q).sum(1))[:, None]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)
    ], -1).reshape(-1, 3, 3)


def _gaussian_window(size, channel, sigma=1.5):
    g = torch.tensor([exp(-(x - size // 2)**2 / float(2 * sigma**2))
                      for x in range(size)])
    g = (g / g.sum()).unsqueeze(1)
    w2d = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2d.expand(channel, 1, size, size).contiguous()


def calc_ssim(img1, img2, window_size=11, size_average=True):
    """structural similarity with an 11x11 Gaussian window (sigma 1.5),
    zero-padded depth-wise convolutions (slam_external_splatam.py:59-96)"""
    if img1.is_cuda and img1.dim() == 3 and window_size == 11 and \
            size_average and img1.dtype == torch.float32:
        # MI355X: one fused launch (and one for the backward) instead of five
        # depth-wise convolutions and their autograd graph
        from ...engine import slam_ops
        return slam_ops.SsimMapFn.apply(img1, img2).mean()
    ch = img1.size(-3)
    win = _gaussian_window(window_size, ch).to(img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, win, padding=pad, groups=ch)
    mu2 = F.conv2d(img2, win, padding=pad, groups=ch)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, win, padding=pad, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, win, padding=pad, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, win, padding=pad, groups=ch) - mu12
    c1, c2 = 0.01**2, 0.03**2
    ssim_map = ((2 * mu12 + c1) * (2 * s12 + c2)) / \
        ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return ssim_map.mean() if size_average else \
        ssim_map.mean(1).mean(1).mean(1)


def accumulate_mean2d_gradient(variables):
    seen = variables['seen'] if 'seen' in variables else \
        variables['radius'] > 0
    variables['means2D_gradient_accum'][seen] += torch.norm(
        variables['means2D'].grad[seen, :2], dim=-1)
    variables['denom'][seen] += 1
    return variables


def transform_to_frame(means3D, w2c, gaussians_grad, camera_grad):
    """world -> camera-frame centres; which side carries gradient is chosen by
    the caller (tracking: pose only; mapping: Gaussians only), :263-292"""
    rel = w2c if camera_grad else w2c.detach()
    pts = means3D if gaussians_grad else means3D.detach()
    pts4 = torch.cat((pts, torch.ones_like(pts[:, :1])), 1)
    return (rel @ pts4.T).T[:, :3]


def get_depth_and_silhouette(pts_3D, w2c):
    """per-Gaussian 'colour' (z, 1, z^2) of the depth/silhouette pass, z in
    the frame of ``w2c`` (:205-222)"""
    pts4 = torch.cat((pts_3D, torch.ones_like(pts_3D[:, :1])), -1)
    z = (w2c @ pts4.transpose(0, 1)).transpose(0, 1)[:, 2]
    return torch.stack([z, torch.ones_like(z), torch.square(z)], -1).float()


def _common_rendervar(params, transformed_pts):
    return {
        'means3D': transformed_pts,
        'rotations': F.normalize(params['unnorm_rotations']),
        'opacities': torch.sigmoid(params['logit_opacities']),
        'scales': torch.exp(torch.tile(params['log_scales'], (1, 3))),
        'means2D': torch.zeros_like(params['means3D'], requires_grad=True)
        + 0,
    }


def transformed_params2rendervar(params, transformed_pts):
    rv = _common_rendervar(params, transformed_pts)
    rv['colors_precomp'] = params['rgb_colors']
    return rv


def transformed_params2depthplussilhouette(params, w2c, transformed_pts):
    rv = _common_rendervar(params, transformed_pts)
    rv['colors_precomp'] = get_depth_and_silhouette(transformed_pts, w2c)
    return rv
