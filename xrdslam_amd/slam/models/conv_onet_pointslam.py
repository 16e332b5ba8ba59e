"""``ConvOnet2`` — the Point-SLAM model (reference:
slam/models/conv_onet_pointslam.py): ``rendering_n_surface`` samples per ray in
[0.98 d, 1.02 d] around the sensor depth (uniform in [near_end, far] for pixels
without depth), features interpolated from the neural point cloud, the POINT
decoders, occupancy compositing, L1 losses (tracking: uncertainty-normalised
with outlier rejection)."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Type, Union

import torch
from torch.nn import Parameter

from ..model_components.decoder_pointslam import POINT
from ..model_components.neural_point_cloud import NeuralPointCloud
from ..common.common import masked_lower_median
from ..model_components.utils import raw2outputs_nerf_color2
from .base_model import Model, ModelConfig


@dataclass
class ConvOnet2Config(ModelConfig):
    _target: Type = field(default_factory=lambda: ConvOnet2)
    use_dynamic_radius: bool = True
    points_batch_size: int = 50000
    cuda_id: int = 0
    pretrained_decoders_middle_fine: Optional[Path] = None
    # model
    model_c_dim: int = 32
    model_pos_embedding_method: str = 'fourier'
    model_use_view_direction: bool = False
    model_encode_rel_pos_in_col: bool = True
    model_encode_exposure: bool = False
    model_encode_viewd: bool = True
    model_exposure_dim: int = 8
    # point cloud
    pointcloud_nn_weighting: str = 'distance'
    pointcloud_nn_num: int = 8
    pointcloud_min_nn_num: int = 2
    pointcloud_radius_add: float = 0.04
    pointcloud_radius_min: float = 0.02
    pointcloud_radius_query: float = 0.08
    pointcloud_fix_interval_when_add_along_ray: bool = False
    pointcloud_n_add: int = 3
    # rendering
    rendering_n_surface: int = 5
    rendering_sample_near_pcl: bool = False
    rendering_near_end_surface: float = 0.98
    rendering_near_end: float = 0.3
    rendering_far_end_surface: float = 1.02
    rendering_sigmoid_coef_mapper: float = 0.1
    # losses
    tracking_w_color_loss: float = 0.5
    mapping_w_color_loss: float = 0.1
    tracking_handle_dynamic: bool = True
    tracking_use_color_in_tracking: bool = True
    mapping_fix_color_decoder: bool = False
    mapping_fix_geo_decoder: bool = True
    mapping_pixels_based_on_color_grad: int = 1000


class ConvOnet2(Model):
    # engine option (not a reference config field): see get_param_groups
    freeze_fixed_decoders = True

    config: ConvOnet2Config

    def __init__(self, config, camera, **kwargs) -> None:
        super().__init__(config=config, camera=camera, bounding_box=None,
                         **kwargs)

    def populate_modules(self):
        super().populate_modules()
        cfg = self.config
        self.decoder = POINT(
            use_dynamic_radius=cfg.use_dynamic_radius,
            pointcloud_nn_weighting=cfg.pointcloud_nn_weighting,
            pointcloud_min_nn_num=cfg.pointcloud_min_nn_num,
            rendering_n_surface=cfg.rendering_n_surface,
            model_encode_rel_pos_in_col=cfg.model_encode_rel_pos_in_col,
            model_encode_exposure=cfg.model_encode_exposure,
            model_encode_viewd=cfg.model_encode_viewd,
            model_exposure_dim=cfg.model_exposure_dim, c_dim=cfg.model_c_dim,
            pos_embedding_method=cfg.model_pos_embedding_method,
            use_view_direction=cfg.model_use_view_direction)
        self.load_pretrain()
        self.masked_indices = None
        self.neural_point_cloud = None
        self.knn_factory = None  # None = the HIP grid kNN

    def load_pretrain(self):
        """geometry decoder from the ConvONet checkpoint (:214-233).  The
        reference ships the checkpoint as a git-LFS pointer; without the file
        the decoders keep their initialisation."""
        path = self.config.pretrained_decoders_middle_fine
        if path is None or not os.path.exists(path) or \
                os.path.getsize(path) < 1024:
            return
        ckpt = torch.load(path, map_location='cpu')
        geo = {k[8 + 7:]: v for k, v in ckpt['model'].items()
               if 'decoder' in k and 'encoder' not in k and 'coarse' in k}
        self.decoder.geo_decoder.load_state_dict(geo, strict=False)

    # -- map update ---------------------------------------------------------------
    def model_update(self, input):
        cfg = self.config
        if self.neural_point_cloud is None:
            extra = {} if self.knn_factory is None else \
                {'knn_factory': self.knn_factory}
            self.neural_point_cloud = NeuralPointCloud(
                c_dim=cfg.model_c_dim, cuda_id=cfg.cuda_id,
                nn_num=cfg.pointcloud_nn_num,
                radius_add=cfg.pointcloud_radius_add,
                radius_min=cfg.pointcloud_radius_min,
                radius_query=cfg.pointcloud_radius_query,
                fix_interval_when_add_along_ray=cfg.
                pointcloud_fix_interval_when_add_along_ray,
                use_dynamic_radius=cfg.use_dynamic_radius,
                N_surface=cfg.rendering_n_surface, N_add=cfg.pointcloud_n_add,
                near_end_surface=cfg.rendering_near_end_surface,
                far_end_surface=cfg.rendering_far_end_surface,
                device=self.device, **extra)
        npc = self.neural_point_cloud
        npc.add_neural_points(
            batch_rays_o=input['batch_rays_o'],
            batch_rays_d=input['batch_rays_d'],
            batch_gt_depth=input['batch_gt_depth'],
            batch_gt_color=input['batch_gt_color'],
            dynamic_radius=input['batch_dynamic_r'])
        if cfg.mapping_pixels_based_on_color_grad > 0:
            npc.add_neural_points(
                batch_rays_o=input['batch_rays_o_grad'],
                batch_rays_d=input['batch_rays_d_grad'],
                batch_gt_depth=input['batch_gt_depth_grad'],
                batch_gt_color=input['batch_gt_color_grad'],
                dynamic_radius=input['batch_dynamic_r_grad'],
                is_pts_grad=True)

    # -- plugin surface -------------------------------------------------------------
    def get_outputs(self, input) -> Dict[str, Union[torch.Tensor, List]]:
        out = self.render_batch_ray(
            rays_d=input['rays_d'], rays_o=input['rays_o'],
            stage=input['stage'], gt_depth=input['target_d'],
            dynamic_r_query=input['batch_dynamic_r'],
            depth_positive=bool(input.get('depth_positive', False) or
                                input.get('static_shapes', False)),
            placed=(input['z_vals'], input['pts'], input.get('rq_pts'))
            if 'pts' in input else None)
        out['stage'] = input['stage']
        return out

    def get_loss_dict(self, outputs, inputs, is_mapping,
                      stage=None) -> Dict[str, torch.Tensor]:
        cfg = self.config
        target_d = inputs['target_d'].squeeze()
        target_rgb = inputs['target_s']
        depth, color = outputs['depth'], outputs['rgb']
        uncertainty = outputs['uncertainty']
        losses = {}
        # captured iterations keep the deselected rays in the batch: the
        # selection arrives as a mask (point_slam.get_model_input)
        ray_valid = inputs.get('ray_valid')
        if not is_mapping and depth.is_cuda and self.fused_track_loss and \
                depth.dtype == torch.float32:
            # median-rejected depth term + masked colour term and their
            # gradients as ONE launch (engine/point.track_loss) instead of
            # ~35 torch launches each way
            from ...engine import point as _pt
            geo, rgb = _pt.track_loss(
                depth, uncertainty, color, target_d, target_rgb, ray_valid,
                cfg.tracking_handle_dynamic,
                cfg.tracking_use_color_in_tracking, cfg.tracking_w_color_loss)
            losses['geo_loss'] = geo
            if cfg.tracking_use_color_in_tracking:
                losses['rgb_loss'] = rgb
            return losses
        if not is_mapping:
            uncertainty = uncertainty.detach()
            nan_mask = (~torch.isnan(depth)) & (~torch.isnan(uncertainty))
            err = torch.abs(target_d - depth)
            tmp = err / torch.sqrt(uncertainty + 1e-10) \
                if cfg.tracking_handle_dynamic else err
            if ray_valid is None:
                med = tmp.median()
            else:
                # (torch.median propagates a NaN in its input: so does this)
                med = masked_lower_median(tmp.detach(), ray_valid)
                med = torch.where((torch.isnan(tmp) & ray_valid).any(),
                                  torch.full_like(med, float('nan')), med)
                nan_mask = nan_mask & ray_valid
            mask = (tmp < 10 * med) & (target_d > 0) & nan_mask
            # masked sums without compaction; masked-out entries are replaced
            # BEFORE the arithmetic so that a NaN there cannot reach the
            # gradient through 0 * NaN
            depth_s = torch.where(mask, depth, target_d)
            unc_s = torch.where(mask, uncertainty, torch.ones_like(uncertainty))
            losses['geo_loss'] = torch.where(mask, torch.clamp(
                torch.abs(target_d - depth_s) / torch.sqrt(unc_s + 1e-10),
                min=0.0, max=1e3), torch.zeros_like(depth)).sum()
            if cfg.tracking_use_color_in_tracking:
                color_s = torch.where(mask[:, None], color, target_rgb)
                losses['rgb_loss'] = cfg.tracking_w_color_loss * \
                    torch.abs(target_rgb - color_s).sum()
        else:
            m = (target_d > 0) & outputs['valid_ray_mask'] & \
                (~torch.isnan(depth))
            if ray_valid is not None:
                m = m & ray_valid
            depth_s = torch.where(m, depth, target_d)
            losses['geo_loss'] = torch.abs(target_d - depth_s).sum()
            if outputs['stage'] == 'color':
                color_s = torch.where(m[:, None], color, target_rgb)
                losses['rgb_loss'] = cfg.mapping_w_color_loss * \
                    torch.abs(target_rgb - color_s).sum()
        return losses

    fused_composite = True   # compositing on xrd_point_composite_* (CUDA)
    fused_track_loss = True  # tracking loss on xrd_point_track_loss (CUDA)

    def fused_map_loss(self, input) -> torch.Tensor:
        """get_outputs + get_loss_dict of a MAPPING iteration whose rays all
        carry a sensor depth (or are masked by ``ray_valid``): the decoders as
        in render_batch_ray, then compositing, the loss and their backward as
        ONE launch (engine/point.map_loss) instead of ~100 small kernels each
        way.  Same arithmetic as the modular hooks
        (tests/test_pointslam_hip.py)."""
        from ...engine import point as _pt
        cfg, dev = self.config, self.device
        rays_o, rays_d = input['rays_o'], input['rays_d']
        S = cfg.rendering_n_surface
        d = input['target_d'].reshape(-1, 1).float()
        if 'pts' in input:
            # placed by the batch kernel (engine/point.batch)
            z_vals, pts = input['z_vals'], input['pts']
            rq = input['rq_pts'] if cfg.use_dynamic_radius \
                else input['batch_dynamic_r']
        else:
            t = torch.linspace(0.0, 1.0, steps=S, device=dev)
            z_vals = cfg.rendering_near_end_surface * d * (1. - t) + \
                cfg.rendering_far_end_surface * d * t
            pts = rays_o[..., None, :] + \
                rays_d[..., None, :] * z_vals[..., :, None]
            rq = input['batch_dynamic_r']
            if cfg.use_dynamic_radius:
                rq = rq.reshape(-1, 1).repeat_interleave(S, dim=0)
        raw, _, point_mask = self.eval_points(
            p=pts.reshape(-1, 3), stage=input['stage'], is_tracker=True,
            ray_pts_num=S, dynamic_r_query=rq)
        color = input['stage'] == 'color'
        return _pt.map_loss(
            raw, z_vals, d, input['target_s'] if color else None, point_mask,
            input.get('ray_valid'), cfg.rendering_sigmoid_coef_mapper,
            cfg.mapping_w_color_loss, int(S / 2 + 1))

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        cfg = self.config
        dec = []
        # a decoder no optimiser owns needs no gradients (the reference's
        # autograd fills them and never reads them); without them the
        # geometry path runs on its fused kernels (engine/point.py)
        self.decoder.geo_decoder.requires_grad_(
            not (cfg.mapping_fix_geo_decoder and self.freeze_fixed_decoders))
        if not cfg.mapping_fix_geo_decoder:
            dec += list(self.decoder.geo_decoder.parameters())
        if not cfg.mapping_fix_color_decoder:
            dec += list(self.decoder.color_decoder.parameters())
        if self.masked_indices is not None:
            self.neural_point_cloud.set_mask(self.masked_indices)
        return {'decoder': dec,
                'geometry': [self.neural_point_cloud.geo_feats],
                'color': [self.neural_point_cloud.col_feats]}

    # -- rendering ------------------------------------------------------------------
    def eval_points(self, p, stage='color', is_tracker=False, pts_views_d=None,
                    ray_pts_num=None, dynamic_r_query=None,
                    exposure_feat=None):
        """decoder over chunks of ``points_batch_size`` points (:235-300)"""
        rets, ray_masks, point_masks = [], [], []
        bs = self.config.points_batch_size
        for k, pi in enumerate(torch.split(p, bs)):
            sl = slice(k * bs, k * bs + pi.shape[0])
            ret, ray_mask, point_mask = self.decoder(
                p=pi.unsqueeze(0), npc=self.neural_point_cloud, stage=stage,
                pts_num=ray_pts_num, is_tracker=is_tracker,
                pts_views_d=None if pts_views_d is None else pts_views_d[sl],
                dynamic_r_query=None if dynamic_r_query is None
                else dynamic_r_query[sl], exposure_feat=exposure_feat)
            ret = ret.squeeze(0)
            if ret.dim() == 1 and ret.shape[0] == 4:
                ret = ret.unsqueeze(0)
            rets.append(ret)
            ray_masks.append(ray_mask)
            point_masks.append(point_mask)
        if len(rets) == 1:      # one chunk: nothing to concatenate
            return rets[0], ray_masks[0], point_masks[0]
        return torch.cat(rets, 0), torch.cat(ray_masks, 0), \
            torch.cat(point_masks, 0)

    def render_batch_ray(self, rays_d, rays_o, stage, gt_depth=None,
                         is_tracker=True, dynamic_r_query=None,
                         exposure_feat=None, depth_positive=False,
                         placed=None):
        """:302-461.  ``depth_positive``: the caller guarantees gt_depth > 0 for
        every ray (the optimisation batches are depth-filtered), which spares
        the size read-back that decides the no-depth branch.  Masks are
        applied with where / masked_fill instead of boolean indexing (same
        values, no compaction, no host synchronisation)."""
        cfg, dev = self.config, self.device
        n_rays, S = rays_o.shape[0], cfg.rendering_n_surface
        if placed is not None and depth_positive and gt_depth is not None \
                and not cfg.model_use_view_direction:
            # samples, points and per-point radii already placed by the
            # batch kernel (engine/point.batch): same values as below
            z_vals, pts, rq_pts = placed
            raw, valid_ray_mask, point_mask = self.eval_points(
                p=pts.reshape(-1, 3), stage=stage, is_tracker=is_tracker,
                pts_views_d=None, ray_pts_num=S,
                dynamic_r_query=rq_pts if cfg.use_dynamic_radius
                else dynamic_r_query, exposure_feat=exposure_feat)
            return self._composite(raw, z_vals, rays_d, point_mask,
                                   valid_ray_mask.to(dev), n_rays, S)
        if gt_depth is not None:
            far = torch.minimum(5 * gt_depth.mean(),
                                torch.max(gt_depth * 1.2)).repeat(
                                    n_rays, 1).float()
            gt_depth = gt_depth.reshape(-1, 1) if torch.numel(gt_depth) != 0 \
                else torch.zeros(n_rays, 1, device=dev)
        else:
            far = 10 * torch.ones((n_rays, 1), device=dev).float()
            gt_depth = torch.zeros(n_rays, 1, device=dev)
        nonzero = (gt_depth > 0).squeeze(-1)
        near_pcl = torch.ones(n_rays, device=dev).type(torch.bool)
        t = torch.linspace(0.0, 1.0, steps=S, device=dev)
        d = gt_depth.reshape(-1, 1).float()
        z_vals = cfg.rendering_near_end_surface * d * (1. - t) + \
            cfg.rendering_far_end_surface * d * t
        # (rays without depth sample at z = 0 either way: near * 0, far * 0)
        if not depth_positive and nonzero.sum() < n_rays:
            if cfg.rendering_sample_near_pcl:
                z0, not_near = self.neural_point_cloud.sample_near_pcl(
                    rays_o[~nonzero].detach().clone(),
                    rays_d[~nonzero].detach().clone(), cfg.rendering_near_end,
                    torch.max(far), S)
                if torch.sum(not_near.ravel()):
                    rows = torch.nonzero(~nonzero, as_tuple=True)[0][not_near]
                    near_pcl[rows] = False
                z_vals[~nonzero, :] = z0
            else:
                z_vals[~nonzero, :] = torch.linspace(
                    cfg.rendering_near_end, torch.max(far), steps=S,
                    device=dev).repeat((~nonzero).sum(), 1)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        views = rays_d.repeat_interleave(S, dim=0).reshape(-1, 3)
        if cfg.use_dynamic_radius:
            dynamic_r_query = dynamic_r_query.reshape(-1, 1) \
                .repeat_interleave(S, dim=0)
        raw, valid_ray_mask, point_mask = self.eval_points(
            p=pts.reshape(-1, 3), stage=stage, is_tracker=is_tracker,
            pts_views_d=views, ray_pts_num=S, dynamic_r_query=dynamic_r_query,
            exposure_feat=exposure_feat)
        out = self._composite(raw, z_vals, rays_d, point_mask,
                              valid_ray_mask.to(dev) & near_pcl, n_rays, S)
        if not cfg.rendering_sample_near_pcl and not depth_positive:
            out['depth'] = out['depth'].masked_fill(~nonzero, 0.0)
        return out

    def _composite(self, raw, z_vals, rays_d, point_mask, valid_ray_mask,
                   n_rays, S):
        cfg, dev = self.config, self.device
        if raw.is_cuda and self.fused_composite and raw.shape[-1] == 4 and \
                S <= 16:
            # compositing as one launch each way (engine/point.composite)
            from ...engine import point as _pt
            depth, uncertainty, color = _pt.composite(
                raw, z_vals, point_mask, cfg.rendering_sigmoid_coef_mapper)
        else:
            with torch.no_grad():
                raw[:, -1].masked_fill_(~point_mask, -100.0)
            raw = raw.reshape(n_rays, S, -1).to(dev)
            depth, uncertainty, color, _ = raw2outputs_nerf_color2(
                raw, z_vals, rays_d, device=dev,
                coef=cfg.rendering_sigmoid_coef_mapper)
        return {'rgb': color, 'depth': depth, 'uncertainty': uncertainty,
                'valid_ray_mask': valid_ray_mask}
