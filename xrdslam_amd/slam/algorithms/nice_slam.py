"""``NiceSLAM`` algorithm plugin (reference: slam/algorithms/nice_slam.py):
stage schedule middle -> fine -> color over a mapping call, optional coarse
pass, per-frame pixel sampling with bbox pre-filtering, colour refinement on
the final frame, LR-factor plumbing into the stage schedulers.  The model call
inside ``get_loss`` is the fused HIP render."""
from __future__ import annotations

import functools
from dataclasses import dataclass, field
from typing import List, Type

import numpy as np
import torch

from ...engine import dist as _dist
from ..common.common import get_rays, get_samples
from ..models.conv_onet import ConvOnetConfig
from .base_algorithm import Algorithm, AlgorithmConfig


@dataclass
class NiceSLAMConfig(AlgorithmConfig):
    _target: Type = field(default_factory=lambda: NiceSLAM)
    model: ConvOnetConfig = field(default_factory=ConvOnetConfig)
    mapping_sample: int = 2048
    min_sample_pixels: int = 100
    tracking_sample: int = 1024
    ray_batch_size: int = 3000
    marching_cubes_bound: List[List[float]] = field(
        default_factory=lambda: [[-3.5, 3], [-3, 3], [-3, 3]])
    mapping_bound: List[List[float]] = field(
        default_factory=lambda: [[-3.5, 3], [-3, 3], [-3, 3]])
    tracking_Wedge: int = 100
    tracking_Hedge: int = 100
    mapping_middle_iter_ratio: float = 0.4
    mapping_fine_iter_ratio: float = 0.6
    mapping_lr_factor: float = 1.0
    mapping_lr_first_factor: float = 5.0
    mapping_color_refine: bool = True


class NiceSLAM(Algorithm):
    config: NiceSLAMConfig

    def __init__(self, config: NiceSLAMConfig, camera, device: str) -> None:
        super().__init__(config, camera, device)
        self.stage = 'color'
        self.marching_cube_bound = torch.from_numpy(
            np.array(config.marching_cubes_bound))
        self.bounding_box = torch.from_numpy(np.array(config.mapping_bound))
        config.model.coarse = config.coarse
        self.model = config.model.setup(camera=camera,
                                        bounding_box=self.bounding_box)
        self.model.to(device)
        self.cur_mesh = None

    def do_mapping(self, cur_frame):
        cfg = self.config
        n_iters = cfg.mapping_n_iters if self.is_initialized() \
            else cfg.mapping_first_n_iters
        outer = 1
        if cur_frame.is_final_frame and cfg.mapping_color_refine:
            # colour refinement on the last frame (nice_slam.py:79-87)
            outer = 5
            cfg.mapping_window_size *= 2
            cfg.mapping_middle_iter_ratio = 0.0
            cfg.mapping_fine_iter_ratio = 0.0
            self.model.config.mapping_fix_color = True
            self.model.config.mapping_frustum_feature_selection = False
        side = cfg.coarse and outer == 1 and self._coarse_side_ok()
        frames = None
        if side:
            # window selections in the usual order (they consume the host RNG)
            with torch.no_grad():
                frames = self.select_optimize_frames(
                    cur_frame, cfg.keyframe_selection_method)
            frames_c = self.select_optimize_frames(cur_frame, 'random')
            self._coarse_on_side_stream(n_iters, frames_c)
        for _ in range(outer):
            if frames is None:
                with torch.no_grad():
                    frames = self.select_optimize_frames(
                        cur_frame, cfg.keyframe_selection_method)
            self.optimize_update(n_iters, frames, is_mapping=True,
                                 coarse=False)
            frames = None
        if cfg.coarse and not side:
            frames = self.select_optimize_frames(cur_frame, 'random')
            self.optimize_update(n_iters, frames, is_mapping=True, coarse=True)
        if not self.is_initialized():
            self.set_initialized()

    # ---- the coarse mapper next to the mapper (MI355X) -----------------------
    # The reference runs the coarse-level mapper as a process of its own,
    # concurrently with the mapper (slam/pipeline: coarse_mapper).  Its work —
    # 60 iterations on grid_coarse alone, ~1/3 of a mapping call's device time
    # at 1000 rays (blocks of ONE ray: far too little to fill 256 CUs) —
    # touches nothing the mapper or the tracker reads: its replayed graphs
    # are enqueued on a second HIP stream BEFORE the mapper's and run under
    # them.  The host still issues the two calls one after the other.
    concurrent_coarse = True

    def _coarse_side_ok(self):
        return (self.concurrent_coarse and self.use_graphs and
                self.persistent_map_graph and self.is_initialized() and
                not _dist.state.enabled and
                torch.device(self.device).type == 'cuda')

    def _coarse_on_side_stream(self, n_iters, frames):
        dev = torch.device(self.device)
        side = self.__dict__.get('_coarse_stream')
        if side is None:
            side = self._coarse_stream = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # the frustum selection belongs to the mapper's grids (grid_coarse is
        # optimised whole) and the packed decoders to the mapper / tracker:
        # the coarse call must not rewrite either under their launches
        self.model.selection_frozen = True
        self._coarse_call = True
        try:
            with torch.cuda.stream(side):
                for f in frames:
                    for t in f.device_images(dev):
                        t.record_stream(side)
                self.optimize_update(n_iters, frames, is_mapping=True,
                                     coarse=True)
        finally:
            self.model.selection_frozen = False
            self._coarse_call = False

    def join_coarse(self):
        """make the current stream wait for the coarse mapper's work (before
        anything reads grid_coarse: rendering through the coarse stage,
        meshing, checkpoints)"""
        side = self.__dict__.get('_coarse_stream')
        if side is not None:
            torch.cuda.current_stream(torch.device(self.device)) \
                .wait_stream(side)

    # the mapping work of a call depends on the window only through data that
    # fits static buffers (images, poses, cell selection): keep its graphs
    persistent_map_graph = True

    def map_slot_key(self, n_iters, optimize_frames, coarse):
        cfg, m, f = self.config, self.model.config, optimize_frames[-1]
        ba = len(self.keyframe_graph) > 4 and not coarse and \
            len(optimize_frames) > 1
        return (len(optimize_frames), ba, n_iters, bool(coarse), f.h, f.w,
                f.separate_LR, f.rot_rep, cfg.mapping_middle_iter_ratio,
                cfg.mapping_fine_iter_ratio, cfg.mapping_sample,
                cfg.min_sample_pixels, cfg.mapping_lr_factor,
                m.mapping_fix_color, m.mapping_frustum_feature_selection)

    def after_mapping_update(self):
        # the tracking graph reads the packed decoder weights in place (the
        # coarse pass trains no decoder)
        if not getattr(self, '_coarse_call', False):
            self.model.sync_decoders(force=True)

    def optimizer_config_update(self, max_iters, coarse=False):
        """nice_slam.py:114-132: BA once >4 keyframes (never in the coarse
        pass); base LR = lr factor (x5 before initialisation, except poses);
        schedulers get the iteration budget and stage ratios."""
        cfg = self.config
        self.bundle_adjust = len(self.keyframe_graph) > 4 and not coarse
        for name, params in cfg.optimizers.items():
            sch = params.get('scheduler')
            if sch is None:
                continue
            first = not (self.is_initialized() or 'pose' in name)
            params['optimizer'].lr = cfg.mapping_lr_first_factor if first \
                else cfg.mapping_lr_factor
            sch.max_steps = max_iters
            sch.coarse = coarse
            sch.middle_iter_ratio = cfg.mapping_middle_iter_ratio
            sch.fine_iter_ratio = cfg.mapping_fine_iter_ratio

    def pre_precessing(self, cur_frame, is_mapping):
        if is_mapping:
            self.model.pre_precessing(cur_frame)
        else:
            self.model.scene()
            self.model.set_grids_trainable(False)

    def post_processing(self, step, is_mapping, optimizer=None, coarse=False):
        if is_mapping:
            self.model.post_processing(coarse)

    def get_model_input(self, optimize_frames, is_mapping):
        cfg = self.config
        dev = self.model.device
        n_pix, Hedge, Wedge = cfg.tracking_sample, cfg.tracking_Hedge, \
            cfg.tracking_Wedge
        if is_mapping:
            n_pix = max(cfg.mapping_sample // len(optimize_frames),
                        cfg.min_sample_pixels)
            Hedge = Wedge = 0
            gen = None
            if _dist.state.enabled and not _dist.state.deterministic:
                # this rank's own 1/world of the rays, from its own stream
                n_pix = _dist.state.shard_count(n_pix)
                gen = _dist.state.shard_generator
        else:
            gen = None
        det = is_mapping and _dist.state.enabled and \
            _dist.state.deterministic
        ro, rd, gd, gc = [], [], [], []
        for frame in optimize_frames:
            c2w = frame.get_pose()
            if is_mapping and not self.bundle_adjust:
                c2w = c2w.detach()
            o, d, dep, col = get_samples(self.camera, n_pix, c2w,
                                         frame.depth, frame.rgb, device=dev,
                                         Hedge=Hedge, Wedge=Wedge, frame=frame,
                                         generator=gen)
            ro.append(o.float())
            rd.append(d.float())
            gd.append(dep.float())
            gc.append(col.float())
        # poses that no optimiser owns need no gradient (BA off / other frames)
        rays_o, rays_d = torch.cat(ro), torch.cat(rd)
        depth, color = torch.cat(gd), torch.cat(gc)
        # drop rays whose sensor depth lies beyond the bound (nice_slam.py:181-194)
        with torch.no_grad():
            if getattr(self, '_bb_dev', None) is None or \
                    self._bb_dev.device != torch.device(dev):
                self._bb_dev = self.bounding_box.to(dev)
            bb = self._bb_dev
            t = (bb.unsqueeze(0) - rays_o.detach().unsqueeze(-1)) / \
                rays_d.detach().unsqueeze(-1)
            t_exit = t.max(dim=2)[0].min(dim=1)[0]
            keep = t_exit >= depth.squeeze(-1)
        extra = {}
        if det:
            # deterministic sharding: every rank drew the SAME batch (shared
            # RNG stream); this rank renders a contiguous slice of every
            # frame's rays.  max(gt_depth) over the kept rays of the WHOLE
            # batch bounds the sampling range (conv_onet.py:418,455), so it
            # is taken before slicing.
            with torch.no_grad():
                extra['dmax'] = torch.where(
                    keep, depth.squeeze(-1),
                    torch.zeros_like(depth.squeeze(-1))).max()
            sel = self._shard_rows(len(optimize_frames), n_pix, dev)
            rays_o, rays_d = rays_o[sel], rays_d[sel]
            depth, color, keep = depth[sel], color[sel], keep[sel]
        if getattr(self, 'fixed_shape_batches', False):
            # no compaction (no host sync, hipGraph friendly): the mask
            # travels with the batch and is applied in the loss
            return {'rays_o': rays_o, 'rays_d': rays_d, 'target_s': color,
                    'target_d': depth, 'stage': self.stage, 'ray_mask': keep,
                    **extra}
        return {'rays_o': rays_o[keep], 'rays_d': rays_d[keep],
                'target_s': color[keep], 'target_d': depth[keep],
                'stage': self.stage, **extra}

    def _shard_rows(self, n_frames, n_pix, dev):
        """row indices of this rank's slice [lo, hi) of every frame's n_pix
        rays in the frame-major batch (deterministic sharding)"""
        lo, hi = _dist.state.shard_slice(n_pix)
        key = (n_frames, n_pix, lo, hi, str(dev))
        cache = self.__dict__.setdefault('_shard_row_cache', {})
        if key not in cache:
            cache[key] = (torch.arange(lo, hi, device=dev).unsqueeze(0) +
                          n_pix * torch.arange(n_frames, device=dev)
                          .unsqueeze(1)).reshape(-1)
        return cache[key]

    def set_stage(self, is_mapping, step, n_iters, coarse=False):
        cfg = self.config
        if not is_mapping:
            self.stage = 'color'
        elif self.model.config.coarse and coarse:
            self.stage = 'coarse'
        elif step <= cfg.mapping_middle_iter_ratio * n_iters:
            self.stage = 'middle'
        elif step <= cfg.mapping_fine_iter_ratio * n_iters:
            self.stage = 'fine'
        else:
            self.stage = 'color'

    def graph_segment_key(self, is_mapping, step, n_iters, coarse=False):
        saved = self.stage
        self.set_stage(is_mapping, step, n_iters, coarse=coarse)
        key, self.stage = self.stage, saved
        return key

    def _fused_loss(self, optimize_frames, is_mapping):
        """get_model_input + model + get_loss_dict as five launches (sampling,
        render, loss and their backwards) instead of ~150 small kernels; same
        arithmetic as the generic hooks (tests/test_nice_loop_hip.py)."""
        from ...engine import nice as _en
        from ...engine import slam_ops
        cfg, cam, dev = self.config, self.camera, self.model.device
        mcfg = self.model.config
        n_pix, Hedge, Wedge = cfg.tracking_sample, cfg.tracking_Hedge, \
            cfg.tracking_Wedge
        gen = None
        if is_mapping:
            n_pix = max(cfg.mapping_sample // len(optimize_frames),
                        cfg.min_sample_pixels)
            Hedge = Wedge = 0
            if _dist.state.enabled and not _dist.state.deterministic:
                n_pix = _dist.state.shard_count(n_pix)
                gen = _dist.state.shard_generator
        det = is_mapping and _dist.state.enabled and \
            _dist.state.deterministic
        wcrop = cam.width - 2 * Wedge
        cnt = (cam.height - 2 * Hedge) * wcrop
        F = len(optimize_frames)
        if self.batched_draws:
            idx = torch.randint(cnt, (F, n_pix), device=dev, generator=gen)
        else:
            # one draw per frame, like get_model_input (the parity tests
            # compare both paths on equal draws)
            idx = torch.stack([torch.randint(cnt, (n_pix, ), device=dev,
                                             generator=gen)
                               for _ in optimize_frames])
        imgs = [f.device_images(dev) for f in optimize_frames]
        bound6 = self.bounding_box.reshape(-1).tolist()
        detach = is_mapping and not self.bundle_adjust
        quat = self._quat_pose_params(optimize_frames, dev, detach)
        if quat is not None and is_mapping and self.fused_map_launch and \
                mcfg.rendering_n_samples == 32 and \
                mcfg.rendering_n_surface == 16:
            return self._fused_map_step(idx, imgs, quat, bound6, det, n_pix,
                                        (Hedge, Wedge, wcrop), detach)
        if is_mapping and not mcfg.mapping_fix_fine:
            raise NotImplementedError(
                'mapping_fix_fine=False needs the one-launch mapping iteration '
                '(quaternion poses on the device, 32 + 16 samples a ray): the '
                'fine decoder\'s weight gradient is formed from its exported '
                'per-sample gradients')
        if quat is not None and not is_mapping and self.fused_map_launch \
                and not det and all(p.requires_grad for p in quat[1]):
            return self._fused_track_step(idx, imgs, quat, bound6,
                                          (Hedge, Wedge, wcrop))
        if quat is not None:
            # poses as parameters: matrices + sampling of all frames in ONE
            # launch (and one for the pose gradients)
            ro, rd, td, tc, keep, dmax = slam_ops.SampleRaysPosesFn.apply(
                idx, [i[0] for i in imgs], [i[1] for i in imgs], cam,
                (Hedge, Wedge, wcrop), bound6, quat[0], *quat[1])
        else:
            poses = []
            for f in optimize_frames:
                c2w = f.get_pose()
                if detach:
                    c2w = c2w.detach()
                poses.append(c2w.to(dev))
            c2ws = torch.stack(poses) if F > 1 else poses[0].unsqueeze(0)
            ro, rd, td, tc, keep, dmax = slam_ops.SampleRaysFn.apply(
                c2ws, idx, [i[0] for i in imgs], [i[1] for i in imgs], cam,
                (Hedge, Wedge, wcrop), bound6)
        if det:
            # same batch on every rank, this rank renders its slice of every
            # frame; dmax (max depth of the kept rays) stays the batch's
            sel = self._shard_rows(F, n_pix, dev)
            ro, rd = ro.index_select(0, sel), rd.index_select(0, sel)
            td, tc, keep = td[sel], tc[sel], keep[sel]
        stage = self.stage
        depth, var, rgb = _en.nice_render(
            self.model.scene(), stage, ro, rd,
            None if stage == 'coarse' else td, dmax=dmax)
        if is_mapping:
            use_color, w = stage == 'color', mcfg.mapping_w_color_loss
        else:
            use_color, w = mcfg.tracking_use_color_in_tracking, \
                mcfg.tracking_w_color_loss
        return slam_ops.NiceLossFn.apply(depth, var, rgb, td, tc, keep,
                                         is_mapping, use_color,
                                         mcfg.tracking_handle_dynamic, w)

    def _fused_map_step(self, idx, imgs, quat, bound6, det, n_pix, crop,
                        detach):
        """a mapping iteration's sampling, render, loss and EVERY gradient as
        four launches (sampling, the fused render+loss+backward, its
        finishing launch, the pose-parameter gradients): nothing goes through
        autograd — the gradients are assigned to ``.grad`` here and the
        returned loss carries no graph (``_iteration`` skips backward)."""
        from ...engine import nice as _en
        from ...engine import slam_ops
        cam, dev, mcfg = self.camera, self.model.device, self.model.config
        layout, params = quat
        (ro, rd, td, tc, keep, dmax), sctx = slam_ops.sample_rays_poses(
            idx, [i[0] for i in imgs], [i[1] for i in imgs], cam, crop,
            bound6, layout, params)
        F = len(imgs)
        sel = None
        if det:
            sel = self._shard_rows(F, n_pix, dev)
            ro_s, rd_s = ro.index_select(0, sel), rd.index_select(0, sel)
            td_s, tc_s, keep_s = td[sel], tc[sel], keep[sel]
        else:
            ro_s, rd_s, td_s, tc_s, keep_s = ro, rd, td, tc, keep
        stage = self.stage
        scene = self.model.scene()
        need_rays = (not detach) and stage != 'coarse' and \
            any(p.requires_grad for p in params)
        need_dec = stage == 'color' and \
            getattr(scene, 'decoder_trainable', True) and \
            scene.dec_flat.get('color') is not None and \
            scene.dec_flat['color'].requires_grad
        # mapping_fix_fine = False: the fine decoder trains in the stages that
        # evaluate it (its weight gradient from the exported per-sample
        # gradients, engine/nice.decoder_weight_grad)
        fine = self.model.decoder.fine_decoder.flat
        need_fine = (not mcfg.mapping_fix_fine) and \
            stage in ('fine', 'color') and fine.requires_grad
        export = {} if need_fine else None
        loss, g_o, g_d, g_flat = _en.nice_map_iter(
            scene, stage, ro_s, rd_s, td_s, dmax, tc_s, keep_s,
            mcfg.mapping_w_color_loss, need_rays, need_dec, export=export)
        if need_dec:
            scene.dec_flat['color'].grad = g_flat
        if need_fine:
            fine.grad = _en.decoder_weight_grad(
                scene, 'fine', fine, export['points'], export['g_occ'])
        if need_rays:
            if sel is not None:
                g_o = torch.zeros_like(ro).index_copy_(0, sel, g_o)
                g_d = torch.zeros_like(rd).index_copy_(0, sel, g_d)
            g7 = slam_ops.sample_rays_poses_bwd(sctx, g_o, g_d)
            for p, g in zip(params, slam_ops.pose_param_grads(g7, layout)):
                if p.requires_grad:
                    p.grad = g
        self._grads_assigned = True     # _iteration: no backward to run
        return loss

    def _fused_track_step(self, idx, imgs, quat, bound6, crop):
        """a tracking iteration without autograd: sampling, forward, robust
        loss, backward to the rays, pose-parameter gradients (five launches +
        the backward's finishing launch); ``.grad`` is assigned here"""
        from ...engine import nice as _en
        from ...engine import slam_ops
        mcfg = self.model.config
        layout, params = quat
        (ro, rd, td, tc, keep, dmax), sctx = slam_ops.sample_rays_poses(
            idx, [i[0] for i in imgs], [i[1] for i in imgs], self.camera,
            crop, bound6, layout, params)
        loss, g_o, g_d = _en.nice_track_iter(
            self.model.scene(), ro, rd, td, dmax, tc, keep,
            mcfg.tracking_use_color_in_tracking, mcfg.tracking_handle_dynamic,
            mcfg.tracking_w_color_loss)
        g7 = slam_ops.sample_rays_poses_bwd(sctx, g_o, g_d)
        for p, g in zip(params, slam_ops.pose_param_grads(g7, layout)):
            p.grad = g
        self._grads_assigned = True     # _iteration: no backward to run
        # the pose this loss was evaluated at (for the keep-the-best-pose
        # step): the sampling launch built the matrix already
        self._iter_c2w = sctx[8][-1]
        return loss

    fused_iteration = True  # use the fused launches when the batch shape is fixed
    # mapping iterations as ONE render launch (forward + loss + backward)
    fused_map_launch = True
    batched_draws = True    # one randint launch for the whole window

    @staticmethod
    def _quat_pose_params(frames, dev, detach):
        """(layout, parameter tensors) for SampleRaysPosesFn, or None when a
        frame's pose is not a float32 quaternion pose on ``dev``"""
        if len(frames) > 16:
            return None
        layout, params = [], []
        for f in frames:
            pose = f.pose
            if pose is None or pose.rot_rep != 'quat':
                return None
            ps = [pose.data_t, pose.data_q] if pose.separate_LR \
                else [pose.data]
            if any(p.device != torch.device(dev) or p.dtype != torch.float32
                   or not p.is_contiguous() for p in ps):
                return None
            layout.append('tq' if pose.separate_LR else '7')
            params += [p.detach() for p in ps] if detach else ps
        return tuple(layout), params

    def get_loss(self, optimize_frames, is_mapping, step, n_iters,
                 coarse=False):
        self.set_stage(is_mapping, step, n_iters, coarse=coarse)
        if is_mapping:
            self.model.grid_processing(coarse=coarse)
        on_gpu = torch.device(self.model.device).type == 'cuda'
        # mapping_fix_fine = False: the fine decoder's weight gradient comes
        # out of the one-launch iteration's export -> that path, captured or
        # not (un-compacted batch + keep mask)
        train_fine = is_mapping and not coarse and \
            not self.model.config.mapping_fix_fine
        if self.fused_iteration and on_gpu and \
                (getattr(self, 'fixed_shape_batches', False) or train_fine):
            return self._fused_loss(optimize_frames, is_mapping)
        if train_fine:
            raise NotImplementedError(
                'mapping_fix_fine=False runs on the fused mapping iteration '
                '(CUDA device, fused_iteration): the generic hooks do not '
                'produce the fine decoder\'s weight gradient')
        model_input = self.get_model_input(optimize_frames, is_mapping)
        outputs = self.model(model_input)
        losses = self.model.get_loss_dict(outputs, model_input, is_mapping,
                                          self.stage)
        return functools.reduce(torch.add, losses.values())

    def render_img(self, c2w, gt_depth=None, idx=None):
        """full-image render in ray_batch_size chunks (nice_slam.py:234-279);
        like the reference this runs under no_grad without taking the lock."""
        self.join_coarse()
        with torch.no_grad():
            dev = self.model.device
            rays_o, rays_d = get_rays(self.camera, c2w, device=dev)
            rays_o = rays_o.reshape(-1, 3).float().contiguous()
            rays_d = rays_d.reshape(-1, 3).float().contiguous()
            if gt_depth is not None:
                gt_depth = torch.as_tensor(gt_depth).to(dev).reshape(-1, 1)
            depths, colors = [], []
            bs = self.config.ray_batch_size
            for i in range(0, rays_d.shape[0], bs):
                out = self.model({
                    'rays_o': rays_o[i:i + bs], 'rays_d': rays_d[i:i + bs],
                    'target_s': None,
                    'target_d': None if gt_depth is None
                    else gt_depth[i:i + bs],
                    'stage': 'color'})
                depths.append(out['depth'].double())
                colors.append(out['rgb'])
            H, W = self.camera.height, self.camera.width
            depth = torch.cat(depths).reshape(H, W)
            color = torch.cat(colors).reshape(H, W, 3)
            return color.cpu().numpy(), depth.cpu().numpy()

    def get_mesh(self):
        """the fine-level occupancy level set over marching_cubes_bound,
        coloured by the colour decoder (nice_slam.py:281-288)"""
        self.join_coarse()
        with self.lock:
            self.model.sync_decoders(force=True)
            self.cur_mesh = self._mesher().get_mesh(
                keyframe_graph=self.keyframe_graph,
                query_fn=self.model.query_fn,
                color_func=self.model.color_func, device=self.device)
            return self.cur_mesh

    def _mesher(self):
        if getattr(self, 'mesher', None) is None:
            from ..common.mesher import MesherConfig
            cfg = getattr(self.config, 'mesher', None) or MesherConfig(
                resolution=256, points_batch_size=30000)
            self.mesher = cfg.setup(
                camera=self.camera, bounding_box=self.bounding_box,
                marching_cubes_bound=self.marching_cube_bound)
        return self.mesher
