"""``GaussianSplatting`` — the SplaTAM model (reference:
slam/models/gaussian_splatting.py): wraps the ``GaussianCloud`` map; tracking
differentiates the renders w.r.t. the camera only, mapping w.r.t. the
Gaussians only; L1 depth + L1 colour (tracking: sums over the silhouette mask;
mapping: means, colour mixed with 0.2 (1 - SSIM))."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Type, Union

import torch
from torch.nn import Parameter

from ..model_components.gaussian_cloud_splatam import GaussianCloud
from ..model_components.slam_helpers_splatam import calc_ssim, l1_loss_v1
from .base_model import Model, ModelConfig


def _dict(**kw):
    return field(default_factory=lambda: dict(**kw))


@dataclass
class GaussianSplattingConfig(ModelConfig):
    _target: Type = field(default_factory=lambda: GaussianSplatting)
    # tracking
    tracking_use_sil_for_loss: bool = True
    tracking_sil_thres: float = 0.99
    tracking_use_l1: bool = True
    tracking_ignore_outlier_depth_loss: bool = False
    tracking_loss_weights: dict = _dict(rgb=0.5, depth=1.0)
    # mapping
    mapping_use_sil_for_loss: bool = False
    mapping_sil_thres: float = 0.5
    mapping_use_l1: bool = True
    mapping_ignore_outlier_depth_loss: bool = False
    mapping_loss_weights: dict = _dict(rgb=0.5, depth=1.0)
    mapping_do_ba: bool = False
    mapping_pruning_dict: dict = _dict(
        start_after=0, remove_big_after=0, stop_after=20, prune_every=20,
        removal_opacity_threshold=0.005,
        final_removal_opacity_threshold=0.005, reset_opacities=False,
        reset_opacities_every=500)
    mapping_use_gaussian_splatting_densification: bool = False
    mapping_densify_dict: dict = _dict(
        start_after=500, remove_big_after=3000, stop_after=5000,
        densify_every=100, grad_thresh=0.0002, num_to_split_into=2,
        removal_opacity_threshold=0.005,
        final_removal_opacity_threshold=0.005, reset_opacities_every=3000)
    mapping_mean_sq_dist_method: str = 'projective'


class GaussianSplatting(Model):
    config: GaussianSplattingConfig

    def populate_modules(self):
        super().populate_modules()
        self.gaussian_cloud = None

    def _frame_tensors(self, frame):
        dev = self.device
        return (torch.as_tensor(frame.rgb).to(dev),
                torch.as_tensor(frame.depth).to(dev),
                torch.inverse(frame.get_pose()).to(dev))

    def model_update(self, cur_frame):
        """first frame: seed the cloud; later: add Gaussians where the
        silhouette says the map is empty (:88-105)"""
        cfg = self.config
        rgb, depth, w2c = self._frame_tensors(cur_frame)
        if self.gaussian_cloud is None:
            self.gaussian_cloud = GaussianCloud(
                init_rgb=rgb, init_depth=depth, w2c=w2c, camera=self.camera,
                prune_dict=cfg.mapping_pruning_dict,
                densify_dict=cfg.mapping_densify_dict)
            self.gaussian_cloud.fused_passes = \
                not cfg.mapping_use_gaussian_splatting_densification
        else:
            self.gaussian_cloud.add_new_gaussians(
                gt_rgb=rgb, gt_depth=depth, curr_w2c=w2c,
                sil_thres=cfg.mapping_sil_thres,
                mean_sq_dist_method=cfg.mapping_mean_sq_dist_method)

    def post_processing(self, iter, optimizer=None):
        if optimizer is None:
            return
        with torch.no_grad():
            self.gaussian_cloud.prune_gaussians(iter, optimizer)
            if self.config.mapping_use_gaussian_splatting_densification:
                self.gaussian_cloud.densify(iter, optimizer)

    def get_outputs(self, input) -> Dict[str, Union[torch.Tensor, List]]:
        is_mapping, retain = input['is_mapping'], input['retain_grad']
        if not is_mapping:
            g_grad, c_grad = False, True       # pose only
        elif self.config.mapping_do_ba:
            g_grad, c_grad = True, True
        else:
            g_grad, c_grad = True, False       # Gaussians only
        if not retain:                         # plain image rendering
            g_grad = c_grad = False
        return self.gaussian_cloud.render(input.get('w2c'), g_grad, c_grad,
                                          retain, c2w=input.get('c2w'))

    def get_loss_dict(self, outputs, inputs, is_mapping,
                      stage=None) -> Dict[str, torch.Tensor]:
        cfg = self.config
        mode = 'mapping' if is_mapping else 'tracking'
        use_sil = getattr(cfg, f'{mode}_use_sil_for_loss')
        sil_thres = getattr(cfg, f'{mode}_sil_thres')
        use_l1 = getattr(cfg, f'{mode}_use_l1')
        ignore_outlier = getattr(cfg, f'{mode}_ignore_outlier_depth_loss')
        weights = getattr(cfg, f'{mode}_loss_weights')
        dev = self.device
        rgb, depth_sil = outputs['rgb'], outputs['depth_sil']
        if rgb.is_cuda and use_l1 and not ignore_outlier:
            return self._loss_dict_fused(outputs, inputs, is_mapping, use_sil,
                                         sil_thres, weights)
        target_d = torch.as_tensor(inputs['target_d']).to(dev).unsqueeze(0)
        target_rgb = torch.permute(
            torch.as_tensor(inputs['target_s']).to(dev), (2, 0, 1)).float()
        depth = depth_sil[0].unsqueeze(0)
        presence = depth_sil[1] > sil_thres
        uncertainty = (depth_sil[2].unsqueeze(0) - depth**2).detach()
        nan_mask = (~torch.isnan(depth)) & (~torch.isnan(uncertainty))
        if ignore_outlier:
            err = torch.abs(target_d - depth) * (target_d > 0)
            mask = (err < 10 * err.median()) & (target_d > 0)
        else:
            mask = target_d > 0
        mask = mask & nan_mask
        if not is_mapping and use_sil:
            mask = mask & presence
        losses = {}
        if use_l1:
            mask = mask.detach()
            d_err = torch.abs(target_d - depth)[mask]
            losses['depth'] = d_err.mean() if is_mapping else d_err.sum()
        if not is_mapping and (use_sil or ignore_outlier):
            cmask = torch.tile(mask, (3, 1, 1)).detach()
            losses['rgb'] = torch.abs(target_rgb - rgb)[cmask].sum()
        elif not is_mapping:
            losses['rgb'] = torch.abs(target_rgb - rgb).sum()
        else:
            losses['rgb'] = 0.8 * l1_loss_v1(rgb, target_rgb) + \
                0.2 * (1.0 - calc_ssim(rgb, target_rgb))
        return {k: v * weights[k] for k, v in losses.items()}

    def _loss_dict_fused(self, outputs, inputs, is_mapping, use_sil,
                         sil_thres, weights):
        """the same losses as one statistics launch + one gradient launch
        (csrc/gs_prepare.hip: xrd_gs_loss_*), targets read from the frame's
        device-resident images: no boolean-mask gathers (a host sync and a
        sort in their backward), no per-iteration upload / permute"""
        from ...engine.gs import GsLossFn
        from ...engine import slam_ops
        dev = self.device
        frame = inputs.get('frame')
        if frame is not None:
            target_d, target_rgb = frame.device_images(dev)
        else:
            target_d = torch.as_tensor(inputs['target_d']).to(dev).float()
            target_rgb = torch.as_tensor(inputs['target_s']).to(dev).float()
        rgb, depth_sil = outputs['rgb'], outputs['depth_sil']
        own = self._own_rows_mask(rgb) if is_mapping else None
        if own is not None:
            # tile-band sharding: this rank rendered its band (+ halo); only
            # the pixels it OWNS feed the L1 terms, the other ranks add theirs
            # (the normalisers stay the full image's: depth count from the
            # target, 3 H W for the colour mean).  Rows outside the rendered
            # band composite nothing, so their terms reach no Gaussian.
            rgb = rgb * own + rgb.detach() * (1.0 - own)
            depth_sil = depth_sil * own + depth_sil.detach() * (1.0 - own)
        ld, lc = GsLossFn.apply(
            rgb, depth_sil, target_d, target_rgb, is_mapping,
            (not is_mapping) and use_sil, sil_thres, weights['depth'],
            weights['rgb'], 0.8 if is_mapping else 1.0)
        if is_mapping:
            chw = frame.device_rgb_chw(dev) if frame is not None else \
                target_rgb.reshape(rgb.shape[1], rgb.shape[2], 3) \
                .permute(2, 0, 1).contiguous()
            ssim_map = slam_ops.SsimMapFn.apply(outputs['rgb'], chw)
            if own is not None:
                # the windows CENTRED on this rank's rows (they read up to 5
                # rendered rows of the neighbouring bands: the halo)
                ssim = (ssim_map * own).sum() / ssim_map.numel()
            else:
                ssim = ssim_map.mean()
            lc = lc + (0.2 * weights['rgb']) * (1.0 - ssim)
        return {'depth': ld, 'rgb': lc}

    # tile-band sharding of the mapping image (set by the algorithm when the
    # mapping work is spread over ranks): (row0, row1) this rank owns
    own_rows = None

    def _own_rows_mask(self, like):
        if self.own_rows is None:
            return None
        key = (tuple(self.own_rows), tuple(like.shape[-2:]), like.device)
        hit = getattr(self, '_own_mask', None)
        if hit is None or hit[0] != key:
            m = torch.zeros(1, like.shape[-2], like.shape[-1],
                            device=like.device)
            m[:, self.own_rows[0]:self.own_rows[1]] = 1.0
            hit = self._own_mask = (key, m)
        return hit[1]

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        return {k: [v.to(self.device)]
                for k, v in self.gaussian_cloud.params.items()}
