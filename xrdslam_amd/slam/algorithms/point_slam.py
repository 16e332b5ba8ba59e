"""``PointSLAM`` algorithm plugin (reference: slam/algorithms/point_slam.py):
before each mapping call neural points are added along sampled rays of the new
frame (uniform pixels + strongest-colour-gradient pixels) with a per-pixel
radius that shrinks where the image has texture; mapping optimises the
features of the points inside the current frustum (geometry stage, then
colour), tracking optimises the pose with uncertainty-normalised losses."""
from __future__ import annotations

import functools
import os
from dataclasses import dataclass, field
from typing import Type

import numpy as np
import torch

from ...engine import dist as _dist
from ..common.common import (color_gradient_magnitude, get_rays, get_samples,
                             masked_lower_median,
                             get_samples_with_pixel_grad)
from ..models.conv_onet_pointslam import ConvOnet2Config
from .base_algorithm import Algorithm, AlgorithmConfig


@dataclass
class PointSLAMConfig(AlgorithmConfig):
    _target: Type = field(default_factory=lambda: PointSLAM)
    model: ConvOnet2Config = field(default_factory=ConvOnet2Config)
    use_dynamic_radius: bool = True
    pixels_adding: int = 6000
    mapping_sample: int = 2048
    min_sample_pixels: int = 100
    tracking_sample: int = 1024
    ray_batch_size: int = 3000
    tracking_sample_with_color_grad: bool = False
    tracking_Wedge: int = 100
    tracking_Hedge: int = 100
    mapping_geo_iter_ratio: float = 0.4
    mapping_pixels_based_on_color_grad: int = 0
    mapping_frustum_feature_selection: bool = True
    mapping_frustum_edge: int = -4
    mapping_BA: bool = False
    model_encode_exposure: bool = False
    pointcloud_radius_add_max: float = 0.08
    pointcloud_radius_add_min: float = 0.02
    pointcloud_radius_add: float = 0.04
    pointcloud_radius_query: float = 0.08
    pointcloud_radius_query_ratio: int = 2
    pointcloud_color_grad_threshold: float = 0.15
    clean_mesh: bool = True


class PointSLAM(Algorithm):
    config: PointSLAMConfig

    def __init__(self, config: PointSLAMConfig, camera, device: str) -> None:
        super().__init__(config, camera, device)
        self.stage = 'color'
        mc = config.model
        mc.model_encode_exposure = config.model_encode_exposure
        mc.use_dynamic_radius = config.use_dynamic_radius
        mc.mapping_pixels_based_on_color_grad = \
            config.mapping_pixels_based_on_color_grad
        self.model = mc.setup(camera=camera)
        self.model.to(device)
        self.dynamic_r_query_allkeyframe = {}
        if torch.device(device).type == 'cuda':
            # the decoders' GEMMs are tall and thin ([2e5, 52] x [52, 128],
            # [25 000, 128] x [128, 128]): rocBLAS picks better kernels for
            # them than hipBLASLt (measured +13 % frames/s); they stay library
            # GEMMs until the fused render kernels exist
            try:
                torch.backends.cuda.preferred_blas_library('cublas')
            except Exception:
                pass

    @property
    def _dev(self):
        return self.model.device

    # -- per-frame preparation ------------------------------------------------------
    def pre_precessing(self, cur_frame, is_mapping):
        cfg = self.config
        depth_np, color_np = cur_frame.depth, cur_frame.rgb
        # (detached: the map is built from the pose's VALUE; rays that kept
        # the pose's autograd graph would pin it inside the cloud's
        # bookkeeping tensors, with its accumulation node bound to this
        # stream, and break later hipGraph captures of the pose gradient)
        c2w, idx = cur_frame.get_pose().detach(), cur_frame.fid
        r_add = None
        if cfg.use_dynamic_radius:
            r_add, r_query = self.cal_dynamic_radius(color_np,
                                                     frame=cur_frame)
            self.dynamic_r_query_allkeyframe[np.array2string(
                np.asarray(idx))] = r_query
        if not is_mapping:
            return
        dev = self._dev
        depth = torch.as_tensor(depth_np).to(dev)
        n_add = cfg.pixels_adding
        if idx == 0:
            n_add = torch.clamp(
                cfg.pixels_adding * ((depth.median() / 2.5)**2),
                min=cfg.pixels_adding, max=cfg.pixels_adding * 3).int().item()
        ro, rd, gd, gc, i, j = get_samples(
            self.camera, n_add, c2w, depth_np, color_np, device=dev,
            depth_filter=True, return_index=True, frame=cur_frame)
        inp = {'batch_rays_o': ro, 'batch_rays_d': rd, 'batch_gt_depth': gd,
               'batch_gt_color': gc,
               'batch_dynamic_r': r_add[j, i] if r_add is not None else None}
        if cfg.mapping_pixels_based_on_color_grad > 0:
            ro2, rd2, gd2, gc2, i2, j2 = get_samples_with_pixel_grad(
                self.camera, cfg.mapping_pixels_based_on_color_grad, c2w,
                depth_np, color_np, device=dev, depth_filter=True,
                return_index=True)
            inp.update({'batch_rays_o_grad': ro2, 'batch_rays_d_grad': rd2,
                        'batch_gt_depth_grad': gd2,
                        'batch_gt_color_grad': gc2,
                        'batch_dynamic_r_grad': r_add[j2, i2]
                        if r_add is not None else None})
        self.model.model_update(inp)
        if cfg.mapping_frustum_feature_selection:
            self.model.masked_indices = self.get_mask_from_c2w(c2w, depth)
        # the per-frame radius cache served the tracking and the mapping call
        # of this frame; the query radii live on in dynamic_r_query_allkeyframe
        # (a keyframe must not pin 2.4 MB of add radii for the whole run)
        cur_frame._dynamic_radius = None

    def post_processing(self, step, is_mapping, optimizer=None, coarse=False):
        pass

    def optimizer_config_update(self, max_iters, coarse=False):
        cfg = self.config
        self.bundle_adjust = len(self.keyframe_graph) > 4 and cfg.mapping_BA
        for params in cfg.optimizers.values():
            if params.get('scheduler') is not None:
                params['optimizer'].lr = 1.0
                params['scheduler'].max_steps = max_iters
                params['scheduler'].geo_iter_ratio = cfg.mapping_geo_iter_ratio

    # -- batches --------------------------------------------------------------------
    def get_model_input(self, optimize_frames, is_mapping):
        cfg, dev = self.config, self._dev
        n, Hedge, Wedge = cfg.tracking_sample, cfg.tracking_Hedge, \
            cfg.tracking_Wedge
        gen = None
        if is_mapping:
            n = int(np.maximum(cfg.mapping_sample // len(optimize_frames),
                               cfg.min_sample_pixels))
            Hedge = Wedge = 0
            if _dist.state.enabled:
                # multi-GPU: the mapping losses are plain sums over rays
                # (conv_onet_pointslam.py:190-204), so each rank renders
                # 1/world of the rays from its own RNG stream and the
                # gradients are summed (Optimizers.optimizer_step_all)
                n = _dist.state.shard_count(n)
                gen = _dist.state.shard_generator
        grad_sampler = not is_mapping and cfg.tracking_sample_with_color_grad
        if self.batched_sampling and not grad_sampler and \
                torch.device(dev).type == 'cuda':
            if self.fused_batch and getattr(self, 'fixed_shape_batches',
                                            False):
                return self._fused_batch(optimize_frames, n, Hedge, Wedge,
                                         gen, is_mapping)
            ro, rd, gd, gc, rq = self._sample_window(
                optimize_frames, n, Hedge, Wedge, gen, is_mapping)
            return self._select_batch(ro, rd, gd, gc, rq)
        ro, rd, gd, gc, rq = [], [], [], [], []
        # the per-frame depth filter (depth > 0) and the batch filter below
        # are applied together, with ONE compaction (one size read-back per
        # iteration instead of one per frame and tensor); the rays kept and
        # their order are the reference's (get_samples(depth_filter=True) per
        # frame, then the ``inside`` selection)
        for f in optimize_frames:
            sampler = get_samples_with_pixel_grad if grad_sampler \
                else functools.partial(get_samples, frame=f, generator=gen)
            # keyframe poses only move under bundle adjustment: without it
            # they are constants of the mapping iteration (no gradient to
            # compute through the rays, the samples and the neighbour weights)
            pose = f.get_pose()
            if is_mapping and not getattr(self, 'bundle_adjust', False):
                pose = pose.detach()
            o, d, dep, col, i, j = sampler(
                self.camera, n, pose, f.depth, f.rgb, device=dev,
                Hedge=Hedge, Wedge=Wedge, depth_filter=grad_sampler,
                return_index=True)
            ro.append(o.float())
            rd.append(d.float())
            gd.append(dep.float().reshape(-1))
            gc.append(col.float())
            if cfg.use_dynamic_radius:
                rq.append(self.dynamic_r_query_allkeyframe[np.array2string(
                    np.asarray(f.fid))][j, i])
        ro, rd, gd, gc = (torch.cat(x) for x in (ro, rd, gd, gc))
        rq = torch.cat(rq) if cfg.use_dynamic_radius else None
        return self._select_batch(ro, rd, gd, gc, rq)

    batched_sampling = True   # the window's rays in one launch (CUDA)
    # captured iterations: batch filter, sample points and query radii in one
    # more launch (engine/point.batch) instead of ~35 torch launches
    fused_batch = True

    def _fused_batch(self, frames, n, Hedge, Wedge, gen, is_mapping):
        from ...engine import point as _pt
        cfg, cam, mc = self.config, self.camera, self.model.config
        ro, rd, td, tc, idx, wcrop = self._sample_window(
            frames, n, Hedge, Wedge, gen, is_mapping, want_radius=False)
        stack = self._radius_stack(frames) if cfg.use_dynamic_radius else None
        geom = (n, wcrop, Hedge, Wedge, cam.width, cam.height * cam.width)
        out = _pt.batch(ro, rd, td, stack, idx.reshape(-1), geom,
                        mc.rendering_n_surface, mc.rendering_near_end_surface,
                        mc.rendering_far_end_surface)
        out.update({'rays_o': ro, 'rays_d': rd, 'target_s': tc,
                    'target_d': td, 'stage': self.stage,
                    'static_shapes': True})
        out.setdefault('batch_dynamic_r', None)
        return out

    def _sample_window(self, frames, n, Hedge, Wedge, gen, is_mapping,
                       want_radius=True):
        """get_samples of every frame of the window (pixels drawn with
        replacement inside the crop, OpenGL rays through the frame's pose,
        sensor depth / colour / query radius of the pixel) as one index draw
        and ONE launch for all frames (xrd_sample_rays_multi: it also builds
        the camera matrices from the quaternion poses), instead of ~15 small
        kernels per frame"""
        from ...engine import slam_ops
        from .nice_slam import NiceSLAM
        cfg, cam, dev = self.config, self.camera, self._dev
        wcrop = cam.width - 2 * Wedge
        cnt = (cam.height - 2 * Hedge) * wcrop
        F = len(frames)
        idx = torch.randint(cnt, (F, n), device=dev, generator=gen)
        imgs = [f.device_images(dev) for f in frames]
        big = 1e30
        bound6 = (-big, big, -big, big, -big, big)
        detach = is_mapping and not getattr(self, 'bundle_adjust', False)
        quat = NiceSLAM._quat_pose_params(frames, dev, detach)
        if quat is not None:
            ro, rd, td, tc, _, _ = slam_ops.SampleRaysPosesFn.apply(
                idx, [i[0] for i in imgs], [i[1] for i in imgs], cam,
                (Hedge, Wedge, wcrop), bound6, quat[0], *quat[1])
        else:
            poses = [f.get_pose().detach() if detach else f.get_pose()
                     for f in frames]
            c2ws = torch.stack([p.to(dev) for p in poses])
            ro, rd, td, tc, _, _ = slam_ops.SampleRaysFn.apply(
                c2ws, idx, [i[0] for i in imgs], [i[1] for i in imgs], cam,
                (Hedge, Wedge, wcrop), bound6, False)
        if not want_radius:
            return ro, rd, td.reshape(-1), tc, idx, wcrop
        rq = None
        if cfg.use_dynamic_radius:
            rows = Hedge + torch.div(idx, wcrop, rounding_mode='floor')
            cols = Wedge + idx % wcrop
            rq = self._radius_stack(frames).gather(
                1, rows * cam.width + cols).reshape(-1)
        return ro, rd, td.reshape(-1), tc, rq

    def _radius_stack(self, frames):
        """[F, H*W] query radii of the window's frames (built once per window:
        the optimisation loop asks for the same frames every iteration)"""
        keys = tuple(np.array2string(np.asarray(f.fid)) for f in frames)
        hit = getattr(self, '_rq_stack', None)
        if hit is None or hit[0] != keys:
            maps = [self.dynamic_r_query_allkeyframe[k].reshape(-1)
                    for k in keys]
            hit = (keys, torch.stack(maps))
            self._rq_stack = hit
        return hit[1]

    def _select_batch(self, ro, rd, gd, gc, rq):
        with torch.no_grad():
            valid = gd > 0
            med = masked_lower_median(gd, valid)
            top = torch.where(valid, gd,
                              torch.full_like(gd, float('-inf'))).max()
            inside = valid & (gd <= torch.minimum(10 * med, 1.2 * top))
        if getattr(self, 'fixed_shape_batches', False):
            # captured iterations (hipGraph): every sampled ray stays in the
            # batch, the selection travels as a mask that the losses apply
            # (same sums; the deselected rays cost a few % of extra work)
            return {'rays_o': ro, 'rays_d': rd, 'target_s': gc, 'target_d': gd,
                    'batch_dynamic_r': rq, 'stage': self.stage,
                    'ray_valid': inside, 'static_shapes': True}
        with torch.no_grad():
            keep = torch.nonzero(inside).reshape(-1)   # the one read-back
        if keep.numel() != inside.numel():
            ro, rd, gc, gd = (t.index_select(0, keep) for t in (ro, rd, gc, gd))
            rq = rq.index_select(0, keep) if rq is not None else None
        # every ray kept carries a positive sensor depth
        return {'rays_o': ro, 'rays_d': rd, 'target_s': gc, 'target_d': gd,
                'batch_dynamic_r': rq, 'stage': self.stage,
                'depth_positive': True}

    # the cloud (positions, features, search grid) is re-allocated whenever
    # points are added: graphs live for ONE tracking / mapping call
    persistent_track_graph = False
    persistent_map_graph = False
    # mapping: compositing + loss + backward as one launch (engine/point.py)
    fused_map_loss = True

    def graph_segment_key(self, is_mapping, step, n_iters, coarse=False):
        """the stage decides the device work AND the learning rates
        (PointSLAMScheduler switches at the same iteration)"""
        saved = getattr(self, 'stage', None)
        self.set_stage(is_mapping, step, n_iters)
        key, self.stage = self.stage, saved
        return key

    def _graphs_ok(self, optimizers, is_mapping):
        # sharded mapping: the fixed-shape batches have the same size on every
        # rank, so the iteration splits into a gradient graph and a step graph
        # around the eager all-reduce (base_algorithm.optimize_update);
        # XRD_POINT_SHARDED_GRAPHS=0 keeps that case eager
        if is_mapping and _dist.state.enabled and \
                os.environ.get('XRD_POINT_SHARDED_GRAPHS', '1') == '0':
            return False
        return super()._graphs_ok(optimizers, is_mapping)

    def set_stage(self, is_mapping, step, n_iters):
        if not is_mapping:
            self.stage = 'color'
        elif step <= int(n_iters * self.config.mapping_geo_iter_ratio):
            self.stage = 'geometry'
        else:
            self.stage = 'color'

    def get_loss(self, optimize_frames, is_mapping, step=None, n_iters=None,
                 coarse=False):
        self.set_stage(is_mapping, step, n_iters)
        inp = self.get_model_input(optimize_frames, is_mapping)
        # tracking steps the pose only: the fused decoders skip the gradients
        # of the map features and the decoder weights (the reference computes
        # and then discards them)
        dec = self.model.decoder
        dec.geo_decoder.map_gradients = dec.color_decoder.map_gradients = \
            bool(is_mapping)
        if is_mapping and self.fused_map_loss and \
                torch.device(self._dev).type == 'cuda' and \
                (inp.get('depth_positive') or inp.get('static_shapes')) and \
                not self.model.config.rendering_sample_near_pcl:
            return self.model.fused_map_loss(inp)
        out = self.model(inp)
        losses = self.model.get_loss_dict(out, inp, is_mapping, self.stage)
        return functools.reduce(torch.add, losses.values())

    def render_img(self, c2w, gt_depth=None, idx=None):
        with torch.no_grad():
            dev, cfg = self._dev, self.config
            rays_o, rays_d = get_rays(self.camera, c2w, device=dev)
            rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
            rq = None
            if cfg.use_dynamic_radius:
                rq = self.dynamic_r_query_allkeyframe[np.array2string(
                    np.asarray(idx))].reshape(-1, 1)
            if gt_depth is not None:
                gt_depth = torch.as_tensor(gt_depth).to(dev).reshape(-1)
            depths, colors = [], []
            bs = cfg.ray_batch_size
            for s in range(0, rays_d.shape[0], bs):
                out = self.model({
                    'rays_o': rays_o[s:s + bs], 'rays_d': rays_d[s:s + bs],
                    'target_s': None, 'stage': 'color',
                    'target_d': None if gt_depth is None
                    else gt_depth[s:s + bs],
                    'batch_dynamic_r': None if rq is None else rq[s:s + bs]})
                depths.append(out['depth'].double())
                colors.append(out['rgb'])
            H, W = self.camera.height, self.camera.width
            return torch.cat(colors).reshape(H, W, 3).cpu().numpy(), \
                torch.cat(depths).reshape(H, W).cpu().numpy()

    # -- helpers ------------------------------------------------------------------
    def cal_dynamic_radius(self, gt_color_np, frame=None):
        """per-pixel add / query radius from the colour-gradient magnitude:
        flat regions get the largest radius, textured ones the smallest
        (piece-wise linear in the clipped gradient, :326-354).  With the
        frame's device-resident f32 image: one kernel
        (xrd_point_dynamic_radius, same f64 arithmetic), computed once per
        frame; otherwise numpy on the host like the reference."""
        cfg = self.config
        if frame is not None and torch.device(self._dev).type == 'cuda' and \
                isinstance(frame.rgb, np.ndarray) and \
                frame.rgb.dtype == np.float32:
            cached = getattr(frame, '_dynamic_radius', None)
            if cached is None:
                from ...engine.map_ops import point_dynamic_radius
                _, rgb = frame.device_images(self._dev)
                cached = frame._dynamic_radius = point_dynamic_radius(
                    rgb.reshape(frame.h, frame.w, 3),
                    cfg.pointcloud_color_grad_threshold,
                    cfg.pointcloud_radius_add_max,
                    cfg.pointcloud_radius_add_min,
                    cfg.pointcloud_radius_query_ratio)
            return cached
        return self.cal_dynamic_radius_host(gt_color_np)

    def cal_dynamic_radius_host(self, gt_color_np):
        cfg = self.config
        color = gt_color_np.cpu().numpy() if torch.is_tensor(gt_color_np) \
            else np.asarray(gt_color_np)
        mag = np.clip(color_gradient_magnitude(color), 0.0,
                      cfg.pointcloud_color_grad_threshold)
        xs = [0, 0.01, cfg.pointcloud_color_grad_threshold]
        ys = [cfg.pointcloud_radius_add_max, cfg.pointcloud_radius_add_max,
              cfg.pointcloud_radius_add_min]
        r_add = np.interp(mag, xs, ys)
        r_query = np.interp(mag, xs, [cfg.pointcloud_radius_query_ratio * y
                                      for y in ys])
        return torch.from_numpy(r_add).to(self._dev), \
            torch.from_numpy(r_query).to(self._dev)

    def get_mask_from_c2w(self, c2w, depth):
        """bool per neural point: projects inside the image (+4 px) and not
        behind the measured depth + 0.5 m (:356-420; bilinear depth lookup with
        zero border in place of cv2.remap).  On the GPU two launches
        (xrd_point_frustum_mask), else ``get_mask_from_c2w_torch``."""
        cam, dev = self.camera, self._dev
        if torch.device(dev).type == 'cuda':
            from ...engine.map_ops import point_frustum_mask
            w2c = torch.linalg.inv(c2w.detach().to(dev).double())
            return point_frustum_mask(
                self.model.neural_point_cloud.cloud_tensor(dev), w2c,
                torch.as_tensor(depth).to(dev), cam.height, cam.width, cam.fx,
                cam.fy, cam.cx, cam.cy, self.config.mapping_frustum_edge)
        return self.get_mask_from_c2w_torch(c2w, depth)

    def get_mask_from_c2w_torch(self, c2w, depth):
        cam, dev = self.camera, self._dev
        H, W, edge = cam.height, cam.width, self.config.mapping_frustum_edge
        pts = self.model.neural_point_cloud.cloud_tensor(dev).double()
        w2c = torch.linalg.inv(c2w.detach().to(dev).double())
        pc = pts @ w2c[:3, :3].T + w2c[:3, 3]
        u = cam.fx * (-pc[:, 0]) + cam.cx * pc[:, 2]
        v = cam.fy * pc[:, 1] + cam.cy * pc[:, 2]
        z = pc[:, 2] + 1e-5
        u, v = (u / z).float(), (v / z).float()
        img = depth.reshape(H, W).float()
        u0, v0 = torch.floor(u), torch.floor(v)
        fu, fv = u - u0, v - v0

        def tap(uu, vv):
            ok = (uu >= 0) & (uu <= W - 1) & (vv >= 0) & (vv <= H - 1)
            val = img[vv.clamp(0, H - 1).long(), uu.clamp(0, W - 1).long()]
            return torch.where(ok, val, torch.zeros_like(val))

        d = tap(u0, v0) * (1 - fu) * (1 - fv) + tap(u0 + 1, v0) * fu * \
            (1 - fv) + tap(u0, v0 + 1) * (1 - fu) * fv + \
            tap(u0 + 1, v0 + 1) * fu * fv
        mask = (u < W - edge) & (u > edge) & (v < H - edge) & (v > edge)
        d = torch.where(d == 0, d.max(), d)
        return mask & (-z >= 0) & (-z.float() <= d + 0.5)

    def update_mesh(self):
        pass

    def get_cloud(self, c2w_np, gt_depth_np):
        npc = self.model.neural_point_cloud
        return np.array(npc.input_pos()), np.array(npc.input_rgb()) / 255.0

    def get_mesh(self):
        raise NotImplementedError('mesh output is out of the hot-path scope')
