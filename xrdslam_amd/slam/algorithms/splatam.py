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


class _FrameSlot:
    """static device buffers a captured mapping iteration reads its frame
    from; quacks like ``Frame`` for GaussianSplatting's fused loss"""

    def __init__(self, d, c, chw):
        self.d, self.c, self.chw = (torch.empty_like(d), torch.empty_like(c),
                                    torch.empty_like(chw))
        self.c2w = torch.empty(4, 4, device=d.device)

    def device_images(self, device):
        return self.d, self.c

    def device_rgb_chw(self, device):
        return self.chw


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

    # ---- hipGraph support (MI355X) ---------------------------------------
    # N changes with every frame (add_new_gaussians): no graph survives a
    # frame; within a frame the iterations between two pruning steps share
    # one captured graph.
    persistent_track_graph = False

    def graph_segment_key(self, is_mapping, step, n_iters, coarse=False):
        if not is_mapping:
            return 0
        cfg = self.model.config
        if cfg.mapping_use_gaussian_splatting_densification:
            return step             # statistics + surgery every iteration
        d = cfg.mapping_pruning_dict

        def surgery(it):
            if it > d['stop_after']:
                return False
            return (it >= d['start_after'] and it % d['prune_every'] == 0) \
                or (it > 0 and d['reset_opacities'] and
                    it % d['reset_opacities_every'] == 0)
        # an iteration that changes the number of Gaussians (or swaps
        # parameters) is a segment of its own; the ones after it start a new
        # one
        before = sum(1 for it in range(step) if surgery(it))
        return 2 * before + (1 if surgery(step) else 0)

    def host_pre_iteration(self, optimize_frames, is_mapping, step):
        """the frame this iteration renders (splatam.py:63: one random frame
        of the window per iteration).  Under graph capture / replay the
        chosen keyframe's pose and images are copied into static slot
        buffers the captured iteration reads."""
        f = optimize_frames[np.random.randint(0, len(optimize_frames))]
        self._chosen = f
        self._set_band(is_mapping)
        if not (getattr(self, 'fixed_shape_batches', False) and is_mapping):
            self._slot_live = False
            return
        dev = self.device
        d, c = f.device_images(dev)
        chw = f.device_rgb_chw(dev)
        slot = getattr(self, '_slot', None)
        if slot is None or slot.d.shape != d.shape:
            slot = self._slot = _FrameSlot(d, c, chw)
        with torch.no_grad():
            slot.d.copy_(d)
            slot.c.copy_(c)
            slot.chw.copy_(chw)
            slot.c2w.copy_(f.get_pose().detach().to(dev))
        self._slot_live = True

    def get_model_input(self, optimize_frames, is_mapping):
        f = getattr(self, '_chosen', None)
        self._chosen = None
        if f is None:     # called outside the optimiser loop
            f = optimize_frames[np.random.randint(0, len(optimize_frames))]
        inp = {'target_s': f.rgb, 'target_d': f.depth, 'frame': f,
               'is_mapping': is_mapping, 'retain_grad': True}
        if getattr(self, '_slot_live', False) and is_mapping and \
                not self.model.config.mapping_do_ba:
            inp['frame'] = self._slot
            inp['c2w'] = self._slot.c2w
            return inp
        pose = f.get_pose().to(self.device)
        if not is_mapping:
            # the iteration's best-pose bookkeeping reads this pose: one
            # quaternion -> matrix launch an iteration instead of two
            self._iter_c2w = pose.detach()
        if pose.is_cuda:
            # the rigid inverse is taken inside the preparation kernel
            # (csrc/gs_prepare.hip) — torch.inverse is an LU factorisation
            # with a host sync per iteration
            inp['c2w'] = pose
        else:
            inp['w2c'] = torch.inverse(pose)
        return inp

    def _graphs_ok(self, optimizers, is_mapping):
        # bundle adjustment optimises the pose of the frame an iteration
        # draws: a captured iteration would bake ONE frame's pose parameters
        # in (the static frame slot carries images and a detached pose only)
        if is_mapping and self.model.config.mapping_do_ba:
            return False
        return super()._graphs_ok(optimizers, is_mapping)

    def _reserve_pair_list(self, frames):
        """size the rasteriser's static (Gaussian, tile) pair list from the
        LARGEST count over every frame of the window (one preparation +
        preprocess launch per frame, one read-back) before the iterations are
        captured: a replay renders a different keyframe through the frozen
        list, and the eager iteration in front of the capture saw only one of
        them.  The reference sizes every pass exactly (num_rendered)."""
        cloud = self.model.gaussian_cloud
        if cloud is None or not frames:
            return
        from ...compat import diff_gaussian_rasterization as dgr
        dev = torch.device(self.device)
        with torch.no_grad():
            counts = [cloud.pair_count(f.get_pose().detach().to(dev))
                      for f in frames]
            total = int(torch.stack(counts).max().item())
        dgr._BIN.reserve(dev, cloud.params['means3D'].shape[0], total)

    @property
    def replicated_mapping(self):
        """Only the fused loss applies the own-rows mask, and the 3D-GS
        densification statistics (means2D gradients) are per-rank partial in
        band mode: with either option set every rank renders and scores the
        WHOLE image and the mapping all-reduce is off (replicated mapping:
        correct, not sharded)"""
        cfg = self.model.config
        return not (cfg.mapping_use_l1 and
                    not cfg.mapping_ignore_outlier_depth_loss and
                    not cfg.mapping_use_gaussian_splatting_densification and
                    torch.device(self.device).type == 'cuda')

    def _set_band(self, on):
        """multi-GPU mapping: one full frame per iteration ->
        every rank rasterises a band of tile rows (plus the SSIM halo) and
        owns the loss terms of its rows; the Gaussian gradients are summed by
        the mapping all-reduce (Optimizers.optimizer_step_all).  Tracking,
        growth and pruning stay replicated on the whole image."""
        from ...compat import diff_gaussian_rasterization as dgr
        from ...engine import dist as xdist
        st = xdist.state
        shardable = not self.replicated_mapping
        if on and st.enabled and st.world > 1 and shardable:
            band = xdist.tile_band(st.rank, st.world, self.camera.height)
            dgr.BAND = band['render_tiles']
            self.model.own_rows = band['own']
        else:
            dgr.BAND = None
            self.model.own_rows = None

    def optimize_update(self, n_iters, optimize_frames, is_mapping,
                        coarse=False):
        self._window = optimize_frames if is_mapping else None
        try:
            out = super().optimize_update(n_iters, optimize_frames,
                                          is_mapping, coarse=coarse)
        finally:
            self._set_band(False)
        self._window = None
        if self.use_graphs and torch.device(self.device).type == 'cuda':
            from ...compat import diff_gaussian_rasterization as dgr
            dgr._BIN.check_replays(torch.device(self.device))
        return out

    def get_loss(self, optimize_frames, is_mapping, step=None, n_iters=None,
                 coarse=False):
        inp = self.get_model_input(optimize_frames, is_mapping)
        out = self.model(inp)
        losses = self.model.get_loss_dict(out, inp, is_mapping)
        return functools.reduce(torch.add, losses.values())

    def pre_precessing(self, cur_frame, is_mapping):
        if is_mapping:
            self.model.model_update(cur_frame)
            if self.use_graphs and torch.device(self.device).type == 'cuda' \
                    and not self.model.config.mapping_do_ba:
                self._reserve_pair_list(getattr(self, '_window', None))

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
