"""TEST INFRASTRUCTURE ONLY (the parity oracle) — not part of the product path.

CPU (torch, fp32/fp64 exactly as the reference mixes them) restatement of the
NICE-SLAM render path of openxrlab/xrdslam.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this file, and only as the checker.

Pinned against the reference itself: ``oracle/make_golden.py`` executes the
reference's own modules (through ``oracle/ref_harness.py``) on seeded inputs and
stores the results under ``tests/golden/nice_*.npz``;
``tests/test_oracle_nice.py`` checks this restatement against those vectors.

Each function cites the reference file:line it restates (paths relative to the
reference root).  Decoder parameters are passed as plain ``state_dict``-style
dicts with the reference's key names (``fc_c.i.weight`` …).
"""
from __future__ import annotations

from typing import Dict, Optional

import contextlib

import torch
import torch.nn.functional as F

# compute type of the "f32" casts of the reference.  float32 restates the
# reference; ``with high_precision():`` evaluates the same formulas in float64
# throughout — the yardstick for terms where two correct float32 evaluations
# (torch on the CPU, torch on CUDA, the fused kernels) differ by more than 1e-4
# because the term itself is ill-conditioned (tests/test_nice_hip.py).
_FT = torch.float32


@contextlib.contextmanager
def high_precision():
    global _FT
    old, _FT = _FT, torch.float64
    try:
        yield
    finally:
        _FT = old


# --------------------------------------------------------------------------
# rays
# --------------------------------------------------------------------------
def rays_from_uv(i, j, c2w, fx, fy, cx, cy):
    """slam/common/common.py:39-53 (get_rays_from_uv): OpenGL pinhole, rays
    NOT normalised; rays_d = sum(dirs * c2w[:3,:3], -1), rays_o = c2w[:3,3]."""
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)],
                       -1)
    dirs = dirs.reshape(-1, 1, 3)
    rays_d = torch.sum(dirs * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


# --------------------------------------------------------------------------
# feature grids + decoders
# --------------------------------------------------------------------------
def normalize_3d(p, bound):
    """slam/common/common.py:16-31 (normalize_3d_coordinate)."""
    p = p.reshape(-1, 3).clone()
    for a in range(3):
        p[:, a] = ((p[:, a] - bound[a, 0]) /
                   (bound[a, 1] - bound[a, 0])) * 2 - 1.0
    return p


def sample_grid(p, grid, bound):
    """slam/model_components/decoder_nice.py:195-205 (sample_grid_feature):
    f64 normalisation -> f32 -> F.grid_sample(bilinear, border,
    align_corners=True) on a [1,C,Z,Y,X] grid; returns [P,C]."""
    p_nor = normalize_3d(p, bound).unsqueeze(0)
    vgrid = p_nor[:, :, None, None].to(_FT)
    c = F.grid_sample(grid, vgrid, padding_mode='border', align_corners=True,
                      mode='bilinear').squeeze(-1).squeeze(-1)
    return c.transpose(1, 2).squeeze(0)


def mlp_forward(sd: Dict[str, torch.Tensor], p, c, skips=(2, ), n_blocks=5):
    """slam/model_components/decoder_nice.py:207-234 (MLP.forward) with the
    Gaussian-Fourier embedding of :33-38 (sin(p @ B), p cast to f32)."""
    p = p.to(_FT)
    emb = torch.sin(p @ sd['embedder._B'])
    h = emb
    for i in range(n_blocks):
        h = F.linear(h, sd[f'pts_linears.{i}.weight'],
                     sd[f'pts_linears.{i}.bias'])
        h = F.relu(h)
        h = h + F.linear(c, sd[f'fc_c.{i}.weight'], sd[f'fc_c.{i}.bias'])
        if i in skips:
            h = torch.cat([emb, h], -1)
    out = F.linear(h, sd['output_linear.weight'], sd['output_linear.bias'])
    return out


def mlp_no_xyz_forward(sd, c, skips=(2, ), n_blocks=5):
    """slam/model_components/decoder_nice.py:308-320 (MLP_no_xyz.forward)."""
    h = c
    for i in range(n_blocks):
        h = F.linear(h, sd[f'pts_linears.{i}.weight'],
                     sd[f'pts_linears.{i}.bias'])
        h = F.relu(h)
        if i in skips:
            h = torch.cat([c, h], -1)
    return F.linear(h, sd['output_linear.weight'],
                    sd['output_linear.bias'])


def nice_forward(p, grids, decoders, bound, stage, coarse_enlarge=2):
    """slam/model_components/decoder_nice.py:386-414 (NICE.forward).
    ``grids``: dict grid_{coarse,middle,fine,color} -> [1,32,Z,Y,X];
    ``decoders``: dict {coarse,middle,fine,color} -> state dict."""
    P = p.shape[0]
    raw = torch.zeros(P, 4, dtype=_FT, device=p.device)
    if stage == 'coarse':
        c = sample_grid(p, grids['grid_coarse'], bound * coarse_enlarge)
        raw[:, 3] = mlp_no_xyz_forward(decoders['coarse'], c).squeeze(-1)
        return raw

    def middle():
        c = sample_grid(p, grids['grid_middle'], bound)
        return mlp_forward(decoders['middle'], p, c).squeeze(-1)

    def fine():
        c = sample_grid(p, grids['grid_fine'], bound)
        with torch.no_grad():  # decoder_nice.py:215-217
            cm = sample_grid(p, grids['grid_middle'], bound)
        return mlp_forward(decoders['fine'], p,
                           torch.cat([c, cm], 1)).squeeze(-1)

    if stage == 'middle':
        raw[:, 3] = middle()
        return raw
    if stage == 'fine':
        f = fine()
        raw[:, 3] = f + middle()
        return raw
    if stage == 'color':
        f = fine()
        c = sample_grid(p, grids['grid_color'], bound)
        rawc = mlp_forward(decoders['color'], p, c)
        occ = f + middle()
        return torch.cat([rawc[:, :3], occ[:, None]], 1)
    raise ValueError(stage)


def eval_points(p, grids, decoders, bound, stage):
    """slam/models/conv_onet.py:339-375: decoder + occupancy logit 100 for
    points outside the (un-enlarged) bound (strict inequalities)."""
    mask = ((p[:, 0] < bound[0][1]) & (p[:, 0] > bound[0][0]) &
            (p[:, 1] < bound[1][1]) & (p[:, 1] > bound[1][0]) &
            (p[:, 2] < bound[2][1]) & (p[:, 2] > bound[2][0]))
    raw = nice_forward(p, grids, decoders, bound, stage)
    occ = torch.where(mask, raw[:, 3], torch.full_like(raw[:, 3], 100.0))
    return torch.cat([raw[:, :3], occ[:, None]], 1)


# --------------------------------------------------------------------------
# sampling + compositing
# --------------------------------------------------------------------------
def sample_z(rays_o, rays_d, bound, gt_depth: Optional[torch.Tensor],
             n_samples=32, n_surface=16):
    """slam/models/conv_onet.py:391-484 (render_batch_ray, sampling part,
    lindisp=False, perturb=0).  Returns z_vals [N,S] float64 (sorted)."""
    if gt_depth is None:
        n_surface = 0
        near = 0.01
    else:
        gt_depth = gt_depth.reshape(-1, 1)
        near = gt_depth.repeat(1, n_samples) * 0.01
    with torch.no_grad():
        o = rays_o.detach().unsqueeze(-1)
        d = rays_d.detach().unsqueeze(-1)
        t = (bound.unsqueeze(0) - o) / d
        far_bb, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
        far_bb = far_bb.unsqueeze(-1) + 0.01
    if gt_depth is not None:
        far = torch.clamp(far_bb, 0, torch.max(gt_depth * 1.2))
    else:
        far = far_bb
    if n_surface > 0:
        nz = (gt_depth > 0).squeeze(-1)
        ts = torch.linspace(0., 1., steps=n_surface,
                            device=gt_depth.device).double()
        dnz = gt_depth[nz].reshape(-1, 1).repeat(1, n_surface)
        z_nz = 0.95 * dnz * (1. - ts) + 1.05 * dnz * ts
        z_surf = torch.zeros(gt_depth.shape[0], n_surface,
                             device=gt_depth.device).double()
        z_surf[nz, :] = z_nz
        z_zero = 0.001 * (1. - ts) + torch.max(gt_depth) * ts
        z_surf[~nz, :] = z_zero
    t_vals = torch.linspace(0., 1., steps=n_samples, device=rays_o.device)
    z_vals = near * (1. - t_vals) + far * t_vals
    if n_surface > 0:
        z_vals, _ = torch.sort(torch.cat([z_vals, z_surf.double()], -1), -1)
    return z_vals


def composite(raw, z_vals, coef=10.0):
    """slam/model_components/utils.py:189-244 (raw2outputs_nerf_color,
    occupancy=True): alpha = sigmoid(coef*occ); w = alpha * excl-cumprod(
    1-alpha+1e-10); rgb/depth/var sums. depth/var come out float64 because
    z_vals is float64."""
    rgb = raw[..., :3]
    alpha = torch.sigmoid(coef * raw[..., 3]).to(_FT)
    ones = torch.ones((alpha.shape[0], 1), dtype=_FT,
                      device=alpha.device)
    weights = alpha * torch.cumprod(
        torch.cat([ones, (1. - alpha + 1e-10).to(_FT)], -1), -1)[:, :-1]
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    tmp = z_vals - depth_map.unsqueeze(-1)
    depth_var = torch.sum(weights * tmp * tmp, dim=1)
    return depth_map, depth_var, rgb_map, weights


def render_batch_ray(rays_o, rays_d, gt_depth, grids, decoders, bound, stage,
                     n_samples=32, n_surface=16):
    """slam/models/conv_onet.py:377-524 (render_batch_ray)."""
    if stage == 'coarse':
        gt_depth = None
    z_vals = sample_z(rays_o, rays_d, bound, gt_depth, n_samples, n_surface)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    raw = eval_points(pts.reshape(-1, 3), grids, decoders, bound, stage)
    raw = raw.reshape(rays_o.shape[0], z_vals.shape[1], 4)
    depth, var, rgb, weights = composite(raw, z_vals)
    return {'rgb': rgb, 'depth': depth, 'uncertainty': var,
            'weights': weights, 'z_vals': z_vals}


def loss_dict(outputs, target_d, target_rgb, is_mapping, stage,
              w_color_track=0.5, w_color_map=0.2):
    """slam/models/conv_onet.py:145-185 (get_loss_dict)."""
    target_d = target_d.squeeze()
    depth, color = outputs['depth'], outputs['rgb']
    unc = outputs['uncertainty'].detach()
    out = {}
    if not is_mapping:
        tmp = torch.abs(target_d - depth) / torch.sqrt(unc + 1e-10)
        mask = (tmp < 10 * tmp.median()) & (target_d > 0)
        out['depth_loss'] = tmp[mask].sum()
        out['rgb_loss'] = w_color_track * torch.abs(target_rgb -
                                                    color)[mask].sum()
    else:
        mask = target_d > 0
        out['depth_loss'] = torch.abs(target_d[mask] - depth[mask]).sum()
        if stage == 'color':
            out['rgb_loss'] = w_color_map * torch.abs(target_rgb -
                                                      color).sum()
    return out


def inside_mask(rays_o, rays_d, gt_depth, bound):
    """slam/algorithms/nice_slam.py:181-194: keep rays whose bbox exit
    distance >= gt depth."""
    o = rays_o.detach().unsqueeze(-1)
    d = rays_d.detach().unsqueeze(-1)
    t = (bound.unsqueeze(0) - o) / d
    t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
    return t >= gt_depth.squeeze(-1)
