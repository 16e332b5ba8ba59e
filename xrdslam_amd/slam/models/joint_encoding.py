"""``JointEncoding`` — Co-SLAM's scene model behind the Model plugin surface
(reference: slam/models/joint_encoding.py).  Hash-grid + OneBlob encodings run
on the HIP kernels (csrc/encodings.hip through the tinycudann-compatible
module); the two 2-layer MLPs, the SDF-bell compositing and the losses are the
reference's formulas (:94-147 losses, :165-197 smoothness, :250-344 sampling
and rendering, :346-406 sdf2weights / raw2outputs, :426-507 queries).

All random draws go through ``self._rand`` so that tests can feed the same
numbers to the reference-generated golden and to this model."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Type, Union

import torch
from torch.nn import Parameter

from ..model_components.decoder_coslam import ColorSDFNet, ColorSDFNet_v2
from ..model_components.encodings_coslam import get_encoder
from ..model_components.utils import (compute_loss, coordinates,
                                      get_sdf_loss)
from .base_model import Model, ModelConfig


@dataclass
class JointEncodingConfig(ModelConfig):
    _target: Type = field(default_factory=lambda: JointEncoding)
    voxel_sdf: float = 0.02
    voxel_color: float = 0.08
    enc: str = 'HashGrid'
    pos_enc: str = 'OneBlob'
    pos_nbins: int = 16
    hashsize: int = 16
    oneGrid: bool = True
    geo_feat_dim: int = 15
    hidden_dim: int = 32
    num_layers: int = 2
    num_layers_color: int = 2
    hidden_dim_color: int = 32
    tcnn_network: bool = False
    tcnn_encoding: bool = False
    trainging_rgb_weight: float = 5.0
    trainging_depth_weight: float = 0.1
    trainging_sdf_weight: float = 1000
    trainging_fs_weight: float = 10
    trainging_smooth_weight: float = 0.000001
    trainging_smooth_pts: int = 32
    trainging_smooth_vox: float = 0.1
    trainging_smooth_margin: float = 0.05
    training_n_samples: int = 256
    training_n_sample_d: int = 32
    training_range_d: float = 0.1
    training_n_range_d: int = 11
    training_n_importance: int = 0
    training_perturb: int = 1
    training_white_bkgd: bool = False
    training_trunc: float = 0.1
    training_rgb_missing: float = 0.05
    data_sc_factor: int = 1
    data_translation: int = 0
    cam_near: float = 0.0
    cam_far: float = 5.0
    cam_depth_trunc: float = 100.0
    mesh_render_color: bool = False


class JointEncoding(Model):
    config: JointEncodingConfig

    def populate_modules(self):
        super().populate_modules()
        cfg = self.config
        dim_max = (self.bounding_box[:, 1] - self.bounding_box[:, 0]).max()
        self.resolution_sdf = cfg.voxel_sdf if cfg.voxel_sdf > 10 else \
            int(dim_max / cfg.voxel_sdf)
        self.resolution_color = cfg.voxel_color if cfg.voxel_color > 10 else \
            int(dim_max / cfg.voxel_color)
        self.embedpos_fn, self.input_ch_pos = get_encoder(
            cfg.pos_enc, n_bins=cfg.pos_nbins)
        self.embed_fn, self.input_ch = get_encoder(
            cfg.enc, log2_hashmap_size=cfg.hashsize,
            desired_resolution=self.resolution_sdf)
        if not cfg.oneGrid:
            self.embed_fn_color, self.input_ch_color = get_encoder(
                cfg.enc, log2_hashmap_size=cfg.hashsize,
                desired_resolution=self.resolution_color)
            self.decoder = ColorSDFNet(cfg, input_ch=self.input_ch,
                                       input_ch_pos=self.input_ch_pos)
        else:
            self.decoder = ColorSDFNet_v2(cfg, input_ch=self.input_ch,
                                          input_ch_pos=self.input_ch_pos)
        self._bb_dev = None

    # random numbers (hookable for parity tests)
    def _rand(self, shape, like):
        return torch.rand(shape, device=like.device, dtype=like.dtype)

    def _bbox(self, device):
        if self._bb_dev is None or self._bb_dev.device != torch.device(device):
            self._bb_dev = self.bounding_box.to(device)
        return self._bb_dev

    # -- plugin surface -------------------------------------------------------
    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        groups = {'decoder': list(self.decoder.parameters()),
                  'embed_fn': list(self.embed_fn.parameters())}
        if not self.config.oneGrid:
            groups['embed_fn_color'] = list(self.embed_fn_color.parameters())
        return groups

    def get_outputs(self, input) -> Dict[str, Union[torch.Tensor, List]]:
        return self.render_rays(input['rays_o'], input['rays_d'],
                                target_d=input['target_d'])

    def get_loss_dict(self, outputs, inputs, is_mapping,
                      stage=None) -> Dict[str, torch.Tensor]:
        cfg = self.config
        target_d, target_rgb = inputs['target_d'], inputs['target_s']
        from ...engine import dist as _dist
        sharded = bool(inputs.get('sharded', False)) and _dist.state.enabled
        # the smoothness term is evaluated identically on every rank (shared
        # RNG stream): 1/world of it per rank sums to one copy
        smooth_scale = 1.0 / _dist.state.world if sharded else 1.0
        if '_maps' in outputs and getattr(self, 'fused_losses', False):
            # fused renderer output + fused loss: the four data terms come
            # as ONE differentiable total ('data_loss'); the individual terms
            # are kept (detached) in self.last_loss_terms
            from ...engine import coslam as ec
            total, l5 = ec.loss(self, outputs, target_d, target_rgb,
                                sharded=sharded, n_live=inputs.get('n_live'))
            self.last_loss_terms = l5
            losses = {'data_loss': total}
            if is_mapping and not inputs['first']:
                if self.fused_smoothness and cfg.tcnn_encoding:
                    # lattice points, hash features, TV loss and its feature
                    # gradient as three launches; the table gradient leaves
                    # with the render backward's scatter (engine/coslam.py)
                    losses['smooth_loss'] = ec.smoothness(
                        self, cfg.trainging_smooth_pts - 1,
                        cfg.trainging_smooth_vox, cfg.trainging_smooth_margin,
                        cfg.trainging_smooth_weight * smooth_scale)
                else:
                    losses['smooth_loss'] = self.smoothness(
                        cfg.trainging_smooth_pts, cfg.trainging_smooth_vox,
                        cfg.trainging_smooth_margin) * \
                        (cfg.trainging_smooth_weight * smooth_scale)
            return losses
        if sharded:
            losses = self._sharded_data_losses(outputs, inputs)
            if is_mapping and not inputs['first']:
                losses['smooth_loss'] = self.smoothness(
                    cfg.trainging_smooth_pts, cfg.trainging_smooth_vox,
                    cfg.trainging_smooth_margin) * \
                    (cfg.trainging_smooth_weight * smooth_scale)
            return losses
        td = target_d.squeeze()
        valid = (td > 0.) * (td < cfg.cam_depth_trunc)
        # reference quirk kept on purpose (joint_encoding.py:106-107): the
        # weight tensor is BOOLEAN, so writing training_rgb_missing (0.05)
        # into the invalid-depth entries stores True — every pixel ends up
        # with colour weight 1
        rgb_w = valid.clone().unsqueeze(-1)
        rgb_w[rgb_w == 0] = cfg.training_rgb_missing
        rgb_loss = compute_loss(outputs['rgb'] * rgb_w, target_rgb * rgb_w)
        if getattr(self, 'fixed_shape_losses', False):
            # same mean over the valid-depth rays without boolean compaction
            nv = valid.sum().clamp(min=1)
            depth_loss = (torch.where(valid, outputs['depth'].squeeze() - td,
                                      torch.zeros_like(td))**2).sum() / nv
        else:
            depth_loss = compute_loss(outputs['depth'].squeeze()[valid],
                                      td[valid])
        if 'rgb0' in outputs:
            # importance sampling: the first pass's maps are supervised too
            # (reference joint_encoding.py:115-119)
            rgb_loss = rgb_loss + compute_loss(outputs['rgb0'] * rgb_w,
                                               target_rgb * rgb_w)
            depth_loss = depth_loss + compute_loss(outputs['depth0'][valid],
                                                   td[valid])
        truncation = cfg.training_trunc * cfg.data_sc_factor
        fs_loss, sdf_loss = get_sdf_loss(outputs['z_vals'], target_d,
                                         outputs['raw'][..., -1], truncation,
                                         'l2', grad=None)
        losses = {'rgb_loss': rgb_loss * cfg.trainging_rgb_weight,
                  'depth_loss': depth_loss * cfg.trainging_depth_weight,
                  'sdf_loss': sdf_loss * cfg.trainging_sdf_weight,
                  'fs_loss': fs_loss * cfg.trainging_fs_weight}
        if is_mapping and not inputs['first']:
            losses['smooth_loss'] = self.smoothness(
                cfg.trainging_smooth_pts, cfg.trainging_smooth_vox,
                cfg.trainging_smooth_margin) * cfg.trainging_smooth_weight
        return losses

    def _sharded_data_losses(self, outputs, inputs):
        """the four data terms for a SHARD of the mapping batch (multi-GPU):
        local sums over this rank's rays divided by the batch-global
        normalisers (ray / valid-depth / mask counts all-reduced first), so
        that the per-rank losses add up to the single-GPU loss and the
        all-reduced gradient equals the single-GPU gradient"""
        import torch.distributed as dist
        cfg = self.config
        target_d, target_rgb = inputs['target_d'], inputs['target_s']
        td = target_d.squeeze(-1)
        valid = (td > 0.) & (td < cfg.cam_depth_trunc)
        w = (valid | bool(cfg.training_rgb_missing)).to(target_rgb.dtype)[:, None]
        rgb_sum = ((outputs['rgb'] * w - target_rgb * w)**2).sum()
        depth_sum = torch.where(valid, (outputs['depth'] - td)**2,
                                torch.zeros_like(td)).sum()
        z, sdf = outputs['z_vals'], outputs['raw'][..., -1]
        trunc = cfg.training_trunc * cfg.data_sc_factor
        front = (z < (target_d - trunc)).to(z.dtype)
        back = (z > (target_d + trunc)).to(z.dtype)
        m = (1.0 - front) * (1.0 - back) * (target_d > 0.0).to(z.dtype)
        fs_sum = ((sdf * front - front)**2).sum()
        sdf_sum = (((z + sdf * trunc) * m - target_d * m)**2).sum()
        counts = torch.stack([front.sum(), m.sum(), valid.sum().to(z.dtype),
                              torch.tensor(float(td.shape[0]), device=z.device,
                                           dtype=z.dtype)]).double()
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        n_fs, n_sdf, n_valid, n = [float(c) for c in counts]
        fs_w = 1.0 - n_fs / (n_fs + n_sdf)
        sdf_w = 1.0 - n_sdf / (n_fs + n_sdf)
        S = z.shape[1]
        return {
            'rgb_loss': rgb_sum / (3.0 * n) * cfg.trainging_rgb_weight,
            'depth_loss': depth_sum / max(n_valid, 1.0) *
            cfg.trainging_depth_weight,
            'sdf_loss': sdf_sum / (n * S) * sdf_w * cfg.trainging_sdf_weight,
            'fs_loss': fs_sum / (n * S) * fs_w * cfg.trainging_fs_weight}

    fused_smoothness = True   # the smoothness term on xrd_hashgrid_tv (CUDA)

    def smoothness(self, sample_points=256, voxel_size=0.1, margin=0.05):
        """total variation of the hash features on a random lattice"""
        dev = self.device
        bb = self._bbox(dev)
        volume = bb[:, 1] - bb[:, 0]
        grid_size = (sample_points - 1) * voxel_size
        offset_max = volume - grid_size - 2 * margin
        offset = self._rand((3, ), offset_max) * offset_max + margin
        coords = coordinates(sample_points - 1, dev, flatten=False).to(volume)
        pts = (coords + self._rand((1, 1, 1, 3), volume)) * voxel_size + \
            bb[:, 0] + offset
        if self.config.tcnn_encoding:
            pts = (pts - bb[:, 0]) / volume
        feat = self.query_sdf(pts, embed=True)
        tv = ((feat[1:] - feat[:-1])**2).sum() + \
            ((feat[:, 1:] - feat[:, :-1])**2).sum() + \
            ((feat[:, :, 1:] - feat[:, :, :-1])**2).sum()
        return tv / (sample_points**3)

    # -- rendering ----------------------------------------------------------
    def _fused_tables(self, device):
        """device tables of the fused renderer, or None when this model /
        device is outside what the fused kernels cover"""
        from ...engine import coslam as ec
        if getattr(self, '_fused_ok', None) is None:
            self._fused_ok = ec.supported(self)
        if not self._fused_ok or not getattr(self, 'use_fused', True) or \
                torch.device(device).type != 'cuda':
            return None
        t = getattr(self, '_fused_tab', None)
        if t is None or t.device != torch.device(device):
            t = self._fused_tab = ec.SceneTables(self, device)
        return t

    def render_rays(self, rays_o, rays_d, target_d=None):
        cfg = self.config
        n_rays = rays_o.shape[0]
        if target_d is not None:
            tab = self._fused_tables(rays_o.device)
            if tab is not None:
                from ...engine import coslam as ec
                S = cfg.training_n_range_d + cfg.training_n_sample_d
                rnd = self._rand((n_rays, S), rays_o) \
                    if cfg.training_perturb > 0. else None
                return ec.render(self, tab, rays_o, rays_d, target_d, rnd,
                                 getattr(self, 'map_trainable', True))
        if target_d is not None:
            lin = torch.linspace(-cfg.training_range_d, cfg.training_range_d,
                                 steps=cfg.training_n_range_d).to(target_d)
            z_samples = lin[None, :].repeat(n_rays, 1) + target_d
            far_lin = torch.linspace(cfg.cam_near, cfg.cam_far,
                                     steps=cfg.training_n_range_d).to(target_d)
            z_samples = torch.where(target_d.reshape(-1, 1) <= 0,
                                    far_lin[None, :], z_samples)
            if cfg.training_n_sample_d > 0:
                z_vals = torch.linspace(cfg.cam_near, cfg.cam_far,
                                        cfg.training_n_sample_d)[None, :] \
                    .repeat(n_rays, 1).to(rays_o)
                z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
            else:
                z_vals = z_samples
        else:
            z_vals = torch.linspace(cfg.cam_near, cfg.cam_far,
                                    cfg.training_n_samples).to(rays_o)
            z_vals = z_vals[None, :].repeat(n_rays, 1)
        if cfg.training_perturb > 0.:
            mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
            upper = torch.cat([mids, z_vals[..., -1:]], -1)
            lower = torch.cat([z_vals[..., :1], mids], -1)
            z_vals = lower + (upper - lower) * self._rand(z_vals.shape,
                                                         rays_o)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        raw = self.run_network(pts)
        rgb, disp, acc, weights, depth, depth_var = self.raw2outputs(
            raw, z_vals, cfg.training_white_bkgd)
        if cfg.training_n_importance > 0:
            # second pass (reference joint_encoding.py:303-325): inverse-CDF
            # draws from the first pass's interior weights, merged and sorted
            # with the first samples; the first pass's maps stay in the
            # outputs (suffix 0) and are supervised by get_loss_dict
            first_pass = {'rgb0': rgb, 'disp0': disp, 'acc0': acc,
                          'depth0': depth, 'depth_var0': depth_var}
            z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
            z_new = self._importance_samples(
                z_mid, weights[..., 1:-1], cfg.training_n_importance,
                cfg.training_perturb == 0.).detach()
            z_vals, _ = torch.sort(torch.cat([z_vals, z_new], -1), -1)
            pts = rays_o[..., None, :] + rays_d[..., None, :] * \
                z_vals[..., :, None]
            raw = self.run_network(pts)
            rgb, disp, acc, weights, depth, depth_var = self.raw2outputs(
                raw, z_vals, cfg.training_white_bkgd)
        ret = {'rgb': rgb, 'depth': depth, 'disp_map': disp, 'acc_map': acc,
               'depth_var': depth_var, 'z_vals': z_vals, 'raw': raw}
        if cfg.training_n_importance > 0:
            ret.update(first_pass)
            ret['z_std'] = torch.std(z_new, dim=-1, unbiased=False)
        return ret

    def _importance_samples(self, bins, weights, n_new, deterministic):
        """n_new depths per ray from the piecewise-linear inverse of the CDF
        of ``weights`` (one weight per interval of ``bins``); reference
        model_components/utils.py:31 (sample_pdf)"""
        pdf = weights + 1e-5
        pdf = pdf / pdf.sum(-1, keepdim=True)
        cdf = torch.nn.functional.pad(torch.cumsum(pdf, -1), (1, 0))
        lead = list(cdf.shape[:-1])
        if deterministic:
            half = 0.5 / n_new
            u = torch.linspace(half, 1. - half, steps=n_new,
                               device=cdf.device).expand(lead + [n_new])
        else:
            u = self._rand(tuple(lead + [n_new]), cdf)
        u = u.contiguous()
        hi = torch.searchsorted(cdf, u, right=True)
        lo = (hi - 1).clamp_min(0)
        hi = hi.clamp_max(cdf.shape[-1] - 1)
        c_lo, c_hi = cdf.gather(-1, lo), cdf.gather(-1, hi)
        b_lo, b_hi = bins.gather(-1, lo), bins.gather(-1, hi)
        width = c_hi - c_lo
        width = torch.where(width < 1e-5, torch.ones_like(width), width)
        return b_lo + (u - c_lo) / width * (b_hi - b_lo)

    def sdf2weights(self, sdf, z_vals):
        """bell-shaped weights sigma(s/tr) sigma(-s/tr), cut tr behind the
        first sign change, normalised"""
        tr = self.config.training_trunc
        w = torch.sigmoid(sdf / tr) * torch.sigmoid(-sdf / tr)
        crossing = (sdf[:, 1:] * sdf[:, :-1] < 0.0).to(sdf.dtype)
        inds = torch.argmax(crossing, axis=1)[..., None]
        z_min = torch.gather(z_vals, 1, inds)
        mask = (z_vals < z_min + self.config.data_sc_factor * tr).to(
            z_vals.dtype)
        w = w * mask
        return w / (torch.sum(w, axis=-1, keepdims=True) + 1e-8)

    def raw2outputs(self, raw, z_vals, white_bkgd=False):
        rgb = torch.sigmoid(raw[..., :3])
        weights = self.sdf2weights(raw[..., 3], z_vals)
        rgb_map = torch.sum(weights[..., None] * rgb, -2)
        depth_map = torch.sum(weights * z_vals, -1)
        depth_var = torch.sum(
            weights * torch.square(z_vals - depth_map.unsqueeze(-1)), dim=-1)
        acc = torch.sum(weights, -1)
        disp = 1. / torch.max(1e-10 * torch.ones_like(depth_map),
                              depth_map / acc)
        if white_bkgd:
            rgb_map = rgb_map + (1. - acc[..., None])
        return rgb_map, disp, acc, weights, depth_map, depth_var

    # -- point queries ---------------------------------------------------------
    def query_sdf(self, query_points, return_geo=False, embed=False):
        flat = query_points.reshape(-1, query_points.shape[-1])
        embedded = self.embed_fn(flat)
        if embed:
            return embedded.reshape(list(query_points.shape[:-1]) +
                                    [embedded.shape[-1]])
        out = self.decoder.sdf_net(torch.cat([embedded,
                                              self.embedpos_fn(flat)], -1))
        sdf = out[..., :1].reshape(list(query_points.shape[:-1]))
        if not return_geo:
            return sdf
        geo = out[..., 1:]
        return sdf, geo.reshape(list(query_points.shape[:-1]) +
                                [geo.shape[-1]])

    def query_color_sdf(self, query_points):
        flat = query_points.reshape(-1, query_points.shape[-1])
        embed = self.embed_fn(flat)
        embed_pos = self.embedpos_fn(flat)
        if not self.config.oneGrid:
            return self.decoder(embed, embed_pos, self.embed_fn_color(flat))
        return self.decoder(embed, embed_pos)

    def query_color(self, query_points):
        return torch.sigmoid(self.query_color_sdf(query_points)[..., :3])

    def run_network(self, inputs):
        flat = inputs.reshape(-1, inputs.shape[-1])
        if self.config.tcnn_encoding:
            bb = self._bbox(flat.device)  # float64 like the reference (B.15)
            flat = (flat - bb[:, 0]) / (bb[:, 1] - bb[:, 0])
        out = self.query_color_sdf(flat)
        return out.reshape(list(inputs.shape[:-1]) + [out.shape[-1]])

    def query_fn(self, pi):
        if self.config.tcnn_encoding:
            bb = self._bbox(pi.device)
            pi = (pi - bb[:, 0]) / (bb[:, 1] - bb[:, 0])
        return self.query_sdf(pi.unsqueeze(1))

    def color_func(self, pi):
        if self.config.tcnn_encoding:
            bb = self._bbox(pi.device)
            pi = (pi - bb[:, 0]) / (bb[:, 1] - bb[:, 0])
        return self.query_color(pi.unsqueeze(1))
