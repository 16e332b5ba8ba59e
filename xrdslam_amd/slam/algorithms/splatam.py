"""``SplaTAM`` algorithm plugin (reference: slam/algorithms/splatam.py): every
frame is tracked (pose only) and mapped (Gaussians only); each mapping
iteration renders ONE random frame of the overlap-selected keyframe window."""
from __future__ import annotations

import functools
from dataclasses import dataclass, field
from typing import Type

import numpy as np
import torch

from ..common.common import keyframe_selection_overlap
from ..models.gaussian_splatting import GaussianSplattingConfig
from .base_algorithm import Algorithm, AlgorithmConfig


@dataclass
class SplaTAMConfig(AlgorithmConfig):
    _target: Type = field(default_factory=lambda: SplaTAM)
    model: GaussianSplattingConfig = field(
        default_factory=GaussianSplattingConfig)
    mapping_sil_thres: float = 0.5
    render_mode: str = 'color'


class SplaTAM(Algorithm):
    config: SplaTAMConfig

    def __init__(self, config: SplaTAMConfig, camera, device: str) -> None:
        super().__init__(config, camera, device)
        self.model = config.model.setup(camera=camera, bounding_box=None)
        self.model.to(device)
        self.bundle_adjust = False

    def select_optimize_frames(self, cur_frame, keyframe_selection_method):
        frames = []
        if len(self.keyframe_graph) > 0:
            frames = keyframe_selection_overlap(
                camera=self.camera, cur_frame=cur_frame,
                keyframes_graph=self.keyframe_graph[:-1],
                k=self.config.mapping_window_size - 2,
                use_ray_sample=self.config.keyframe_use_ray_sample,
                device=self.device)
            frames = list(frames) + [self.keyframe_graph[-1]]
        if cur_frame is not None:
            frames = frames + [cur_frame]
        return frames

    def get_model_input(self, optimize_frames, is_mapping):
        f = optimize_frames[np.random.randint(0, len(optimize_frames))]
        inp = {'target_s': f.rgb, 'target_d': f.depth, 'frame': f,
               'is_mapping': is_mapping, 'retain_grad': True}
        pose = f.get_pose().to(self.device)
        if pose.is_cuda:
            # the rigid inverse is taken inside the preparation kernel
            # (csrc/gs_prepare.hip) — torch.inverse is an LU factorisation
            # with a host sync per iteration
            inp['c2w'] = pose
        else:
            inp['w2c'] = torch.inverse(pose)
        return inp

    def get_loss(self, optimize_frames, is_mapping, step=None, n_iters=None,
                 coarse=False):
        inp = self.get_model_input(optimize_frames, is_mapping)
        out = self.model(inp)
        losses = self.model.get_loss_dict(out, inp, is_mapping)
        return functools.reduce(torch.add, losses.values())

    def pre_precessing(self, cur_frame, is_mapping):
        if is_mapping:
            self.model.model_update(cur_frame)

    def post_processing(self, step, is_mapping, optimizer=None, coarse=False):
        if is_mapping:
            self.model.post_processing(step, optimizer)

    def optimizer_config_update(self, max_iters, coarse=False):
        pass

    def render_img(self, c2w, gt_depth=None, idx=None, use_sil_depth=True):
        with torch.no_grad():
            if isinstance(c2w, np.ndarray):
                c2w = torch.from_numpy(c2w)
            c2w = c2w.to(self.device)
            pose = {'c2w': c2w} if c2w.is_cuda else \
                {'w2c': torch.inverse(c2w)}
            out = self.model({**pose, 'is_mapping': True,
                              'retain_grad': False})
            rdepth = out['depth_sil'][0] if use_sil_depth \
                else out['depth'].squeeze(0)
            valid = torch.as_tensor(gt_depth > 0).to(self.device) & \
                (~torch.isnan(rdepth))
            return (out['rgb'] * valid).detach().cpu().permute(1, 2, 0) \
                .numpy(), (rdepth * valid).detach().cpu().numpy()

    def update_mesh(self):
        pass

    def get_mesh(self):
        raise NotImplementedError('mesh output is out of the hot-path scope')

    def get_cloud(self, c2w_np, gt_depth_np):
        raise NotImplementedError('point-cloud export (viewer) is out of the '
                                  'hot-path scope')
