"""``CoSLAM`` algorithm plugin (reference: slam/algorithms/coslam.py): a global
bank of keyframe rays (5 % of the pixels of every keyframe as [dir_cam(3),
rgb(3), depth(1)] rows), mapping batches = bank rays + current-frame rays
rotated by their (optimisable) keyframe poses, ONE persistent model optimiser
across all mapping calls, bundle adjustment of every keyframe pose but the
first with 5-step gradient accumulation.

MI355X-side: the ray bank and the per-frame ray table live in HBM and are
sampled there (the reference keeps the bank on the host, draws indices with
python's ``random.sample`` and uploads the batch every iteration,
coslam.py:139-150,188-191); sampling stays WITHOUT replacement like
``random.sample``."""
from __future__ import annotations

import functools
from dataclasses import dataclass, field
from typing import List, Type

import numpy as np
import torch

from ...engine import dist as _dist
from ..common.common import get_rays, get_samples
from ..engine.optimizers import Optimizers
from ..models.joint_encoding import JointEncodingConfig
from .base_algorithm import Algorithm, AlgorithmConfig, _capture


@dataclass
class CoSLAMConfig(AlgorithmConfig):
    _target: Type = field(default_factory=lambda: CoSLAM)
    model: JointEncodingConfig = field(default_factory=JointEncodingConfig)
    rays_to_save_ratio: float = 0.05
    tracking_Wedge: int = 20
    tracking_Hedge: int = 20
    mapping_sample: int = 2048
    min_sample_pixels: int = 100
    tracking_sample: int = 1024
    ray_batch_size: int = 3000
    marching_cubes_bound: List[List[float]] = field(
        default_factory=lambda: [[-3.5, 3], [-3, 3], [-3, 3]])
    mapping_bound: List[List[float]] = field(
        default_factory=lambda: [[-3.5, 3], [-3, 3], [-3, 3]])


def camera_ray_table(camera, device):
    """[H*W, 3] camera-frame ray directions, row-major pixels, OpenGL
    (slam/utils/utils.py:28-65 get_camera_rays)"""
    j, i = torch.meshgrid(
        torch.arange(camera.height, dtype=torch.float32, device=device),
        torch.arange(camera.width, dtype=torch.float32, device=device),
        indexing='ij')
    dirs = torch.stack([(i - camera.cx) / camera.fx,
                        -(j - camera.cy) / camera.fy, -torch.ones_like(i)],
                       -1)
    return dirs.reshape(-1, 3)


class CoSLAM(Algorithm):
    config: CoSLAMConfig

    def __init__(self, config: CoSLAMConfig, camera, device: str) -> None:
        super().__init__(config, camera, device)
        self.marching_cube_bound = torch.from_numpy(
            np.array(config.marching_cubes_bound))
        self.bounding_box = torch.from_numpy(np.array(config.mapping_bound))
        self.model = config.model.setup(camera=camera,
                                        bounding_box=self.bounding_box)
        self.model.to(device)
        self.cur_mesh = None
        self.bundle_adjust = True
        self.num_rays_to_save = int(camera.width * camera.height *
                                    config.rays_to_save_ratio)
        self.rays = None           # [n_kf * num_rays_to_save, 7] on the device
        self.model_optimizers = None
        self._dirs = None

    # -- hooks that are no-ops for Co-SLAM --------------------------------------
    def pre_precessing(self, cur_frame, is_mapping):
        if not is_mapping and getattr(self.model, 'use_fused', True) and \
                torch.device(self.model.device).type == 'cuda' and \
                self.model._fused_tables(self.model.device) is not None:
            # tracking reads the decoder through a static packed buffer:
            # bring it up to date BEFORE the (possibly replayed) iterations
            from ...engine import coslam as ec
            ec.track_pack(self.model, self.model.device, refresh=True)

    def after_mapping_update(self):
        # the decoder was stepped (possibly by replayed graphs, which torch's
        # version counters do not see): tracking re-packs it before its next
        # iterations
        self.model._track_pack_key = None

    def post_processing(self, step, is_mapping, optimizer=None, coarse=False):
        pass

    def optimizer_config_update(self, max_iters, coarse=False):
        pass

    # -- optimisers (coslam.py:66-112) -----------------------------------------
    def setup_optimizers(self, n_iters, optimize_frames, is_mapping=True,
                         coarse=False) -> Optimizers:
        cfg = dict(self.config.optimizers)
        if not is_mapping:
            return Optimizers(cfg, self._pose_groups(optimize_frames[:1],
                                                     'tracking_pose'))
        if self.model_optimizers is None:
            self.model_optimizers = Optimizers(
                cfg, {**self.model.get_param_groups()})
        self._ba = None
        if not self.bundle_adjust or len(optimize_frames) == 1:
            return self.model_optimizers
        # the first keyframe's pose stays fixed
        if self._stack_ba(optimize_frames):
            pose_opt = Optimizers(cfg, {'mapping_pose_r': [self._ba['r']],
                                        'mapping_pose_t': [self._ba['t']]})
        else:
            pose_opt = Optimizers(cfg, self._pose_groups(optimize_frames[1:],
                                                         'mapping_pose'))
        merged = pose_opt + self.model_optimizers
        merged.parameters = {**pose_opt.parameters,
                             **self.model_optimizers.parameters}
        return merged

    # -- bundle adjustment on stacked pose parameters ----------------------------
    def _stack_ba(self, frames):
        """MI355X: the poses of one mapping call are optimised as TWO stacked
        parameters ([n,3] rotations, [n,3] translations) instead of 2n small
        ones — Adam is element-wise, so the arithmetic is that of the
        reference's per-frame parameters (coslam.py:66-112), but pose
        matrices, their gradients and the optimiser step cost a constant
        number of launches however many keyframes there are.  Row 0 (the first
        keyframe) receives no gradient.  ``_unstack_ba`` writes the result
        back into the frames."""
        dev = torch.device(self.model.device)
        if not (self.fused_iteration and dev.type == 'cuda' and
                self.config.separate_LR and
                self.config.rot_rep == 'axis_angle' and
                len(frames) == len(self.keyframe_graph) + 1 and
                self.model._fused_tables(dev) is not None):
            return False
        with torch.no_grad():
            r = torch.stack([f.pose.data_r.detach().to(dev) for f in frames])
            t = torch.stack([f.pose.data_t.detach().to(dev) for f in frames])
        fixed = torch.zeros(len(frames), 1, 1, dtype=torch.bool, device=dev)
        fixed[0] = True
        self._ba = {'frames': list(frames), 'r': torch.nn.Parameter(r),
                    't': torch.nn.Parameter(t), 'fixed': fixed}
        return True

    def _unstack_ba(self):
        ba, self._ba = getattr(self, '_ba', None), None
        if ba is None:
            return
        with torch.no_grad():
            dst = [f.pose.data_r for f in ba['frames']] + \
                  [f.pose.data_t for f in ba['frames']]
            src = [x.to(d.device) for x, d in zip(
                list(ba['r'].detach().unbind(0)) +
                list(ba['t'].detach().unbind(0)), dst)]
            torch._foreach_copy_(dst, src)

    def do_mapping(self, cur_frame):
        try:
            super().do_mapping(cur_frame)
        finally:
            self._unstack_ba()

    # -- ray bank ---------------------------------------------------------------
    def _ray_dirs(self):
        dev = self.model.device
        if self._dirs is None or self._dirs.device != torch.device(dev):
            self._dirs = camera_ray_table(self.camera, dev)
        return self._dirs

    def sample_single_keyframe_rays(self, keyframe, bs):
        """bs rows [dir_cam, rgb, depth] of one frame, without replacement"""
        dev = self.model.device
        depth, rgb = keyframe.device_images(dev)
        n = self.camera.height * self.camera.width
        idx = self._distinct(n, bs, dev)
        return torch.cat([self._ray_dirs()[idx], rgb[idx], depth[idx]], -1)

    def _distinct(self, total, bs, dev):
        """bs distinct indices in [0,total) — random.sample of the reference
        (coslam.py:122,147); O(bs) on the GPU, a randperm elsewhere"""
        if torch.device(dev).type == 'cuda':
            from ...engine import slam_ops
            return slam_ops.sample_distinct(total, bs, dev)
        return torch.randperm(total, device=dev)[:bs]

    def add_keyframe(self, keyframe):
        with self.lock:
            rays = self.sample_single_keyframe_rays(keyframe,
                                                    self.num_rays_to_save)
            # capacity buffer (doubling): the bank keeps its address while
            # keyframes are appended, so a captured mapping iteration can
            # sample it on every replay
            n_have = 0 if self.rays is None else self.rays.shape[0]
            bank = getattr(self, '_bank', None)
            if bank is None or n_have + rays.shape[0] > bank.shape[0]:
                cap = max(32 * self.num_rays_to_save,
                          2 * (n_have + rays.shape[0]))
                new = torch.empty(cap, rays.shape[1], dtype=rays.dtype,
                                  device=rays.device)
                if n_have:
                    new[:n_have] = self.rays
                self._bank = bank = new
                self._bank_version = getattr(self, '_bank_version', 0) + 1
            bank[n_have:n_have + rays.shape[0]] = rays
            self.rays = bank[:n_have + rays.shape[0]]
            # only pose and rays are kept (coslam.py:135-137)
            keyframe.rgb = None
            keyframe.depth = None
            keyframe._dev_cache = None
            self.keyframe_graph.append(keyframe)

    def sample_global_rays(self, bs):
        total = len(self.keyframe_graph) * self.num_rays_to_save
        idx = self._distinct(total, bs, self.rays.device)
        return self.rays[idx], torch.div(idx, self.num_rays_to_save,
                                         rounding_mode='floor')

    def get_model_input(self, optimize_frames, is_mapping):
        cfg, dev = self.config, self.model.device
        cur = optimize_frames[-1]
        if not is_mapping:
            o, d, dep, col = get_samples(self.camera, cfg.tracking_sample,
                                         cur.get_pose(), cur.depth, cur.rgb,
                                         device=dev,
                                         Hedge=cfg.tracking_Hedge,
                                         Wedge=cfg.tracking_Wedge, frame=cur)
            return {'rays_o': o.float(), 'rays_d': d.float(),
                    'target_s': col.float(), 'target_d': dep.float(),
                    'first': False}
        ids, rays, poses = [], [], []
        n_cur = cfg.mapping_sample
        have_kf = len(self.keyframe_graph) > 0
        if have_kf:
            bank, fid = self.sample_global_rays(cfg.mapping_sample)
            ids.append(fid)
            rays.append(bank)
            for f in optimize_frames[:-1]:
                pose = f.get_pose().unsqueeze(0).to(dev)
                poses.append(pose.detach() if f.fid == 0 else pose)
            n_cur = max(cfg.mapping_sample // len(self.keyframe_graph),
                        cfg.min_sample_pixels)
        cur_rays = self.sample_single_keyframe_rays(cur, n_cur)
        poses.append(cur.get_pose().unsqueeze(0).to(dev))
        # index -1 = the current frame = last pose
        ids.append(torch.full((cur_rays.shape[0], ), len(poses) - 1,
                              dtype=torch.int64, device=dev))
        rays.append(cur_rays)
        poses = torch.cat(poses, 0)
        ids = torch.cat(ids, 0)
        rays = torch.cat(rays, 0)
        R = poses[ids, :3, :3]
        rays_d = (rays[:, None, :3] * R).sum(-1)
        rays_o = poses[ids, :3, 3]
        return {'rays_o': rays_o.float(), 'rays_d': rays_d.float(),
                'target_s': rays[:, 3:6].float(),
                'target_d': rays[:, 6:7].float(), 'first': not have_kf}

    fused_iteration = True  # fused sampling / loss launches on the GPU

    def _fused_track_input(self, cur):
        """tracking batch in ONE launch (pixel sampling, colour/depth gather,
        rays from the pose; backward to the pose in one more): the same
        arithmetic as get_samples (slam/common/common.py:188-227)"""
        from ...engine import slam_ops
        cfg, cam, dev = self.config, self.camera, self.model.device
        wcrop = cam.width - 2 * cfg.tracking_Wedge
        cnt = (cam.height - 2 * cfg.tracking_Hedge) * wcrop
        idx = torch.randint(cnt, (1, cfg.tracking_sample), device=dev)
        d_img, c_img = cur.device_images(dev)
        c2w = cur.get_pose().to(dev)
        # (the iteration's best-pose bookkeeping reads this pose: one
        # axis-angle -> matrix launch an iteration instead of two)
        self._iter_c2w = c2w.detach()
        ro, rd, td, tc, _keep, _dmax = slam_ops.SampleRaysFn.apply(
            c2w.unsqueeze(0), idx, [d_img], [c_img], cam,
            (cfg.tracking_Hedge, cfg.tracking_Wedge, wcrop),
            self.bounding_box.reshape(-1).tolist(), False)
        return {'rays_o': ro, 'rays_d': rd, 'target_s': tc, 'target_d': td,
                'first': False}

    def _fused_map_input(self, optimize_frames):
        """mapping batch with the pose gather / rotation (and its backward)
        in one launch each and all poses evaluated as one batch"""
        from ...engine import slam_ops
        from ..utils.opt_pose import axis_angle_translation_to_matrix
        cfg, dev = self.config, self.model.device
        slot = getattr(self, '_pslot', None)
        if slot is not None:
            return self._slot_map_input(slot)
        cur = optimize_frames[-1]
        K = len(self.keyframe_graph)
        ba = getattr(self, '_ba', None)
        if ba is not None:
            assert len(ba['frames']) == len(optimize_frames) == K + 1
            c2w = axis_angle_translation_to_matrix(ba['r'], ba['t'])
            c2w = torch.where(ba['fixed'], c2w.detach(), c2w)
        else:  # no pose is stepped in this call
            with torch.no_grad():
                c2w = torch.stack([f.get_pose().to(dev)
                                   for f in optimize_frames])
        rows, ids = [], []
        n_cur = cfg.mapping_sample
        if K > 0:
            bank, fid = self.sample_global_rays(cfg.mapping_sample)
            rows.append(bank)
            ids.append(fid)
            n_cur = max(cfg.mapping_sample // K, cfg.min_sample_pixels)
        rows.append(self.sample_single_keyframe_rays(cur, n_cur))
        ids.append(torch.full((n_cur, ), c2w.shape[0] - 1, dtype=torch.int64,
                              device=dev))
        rows, ids = torch.cat(rows, 0), torch.cat(ids, 0)
        sharded = _dist.state.enabled
        if sharded:
            # every rank drew the SAME batch (shared RNG stream); it renders a
            # contiguous 1/world slice of it.  Loss normalisers are made
            # global in the loss (engine/coslam.py), gradients are summed in
            # Optimizers.optimizer_step_all
            n, W, r = rows.shape[0], _dist.state.world, _dist.state.rank
            lo, hi = (n * r) // W, (n * (r + 1)) // W
            rows, ids = rows[lo:hi].contiguous(), ids[lo:hi].contiguous()
        rays_o, rays_d = slam_ops.PoseRaysFn.apply(c2w, rows, ids)
        return {'rays_o': rays_o, 'rays_d': rays_d, 'target_s': rows[:, 3:6],
                'target_d': rows[:, 6:7], 'first': K == 0,
                'sharded': sharded}

    # -- persistent mapping graphs (MI355X) -----------------------------------
    # A mapping call is 10 iterations; built per call, its optimisers, eager
    # first iterations and captures cost as much as the iterations.  A slot
    # keeps everything a captured iteration touches at fixed addresses with
    # CAPACITIES instead of sizes — pose stacks of K_cap rows, the bank's
    # capacity buffer, a current-frame part of `bucket` rays — and reads the
    # sizes on the device: bank population (xrd_sample_distinct_dev), id of
    # the current frame's pose row, live ray count (xrd_coslam_loss_live: the
    # rows behind it take part in nothing).  One slot serves every call whose
    # current-frame ray count falls in its bucket; from the 21st keyframe on
    # that count is constant (min_sample_pixels).
    persistent_map = True
    # capture the graphs of every bucket when the first slot is needed
    prewarm_slots = True
    _BUCKETS = (128, 256, 512, 1024, 2048)

    def graph_segment_key(self, is_mapping, step, n_iters, coarse=False):
        if not is_mapping:
            return 0
        acc = self.config.optimizers['mapping_pose_r']['optimizer'].accum_step
        return int(acc is not None and (step + 1) % acc == 0)

    def _persistent_map_ok(self, n_iters, frames):
        dev = torch.device(self.model.device)
        K = len(self.keyframe_graph)
        return (self.persistent_map and self.use_graphs and
                self.fused_iteration and dev.type == 'cuda' and
                not _dist.state.enabled and self.is_initialized() and
                self.bundle_adjust and K >= 1 and len(frames) == K + 1 and
                self.config.separate_LR and
                self.config.rot_rep == 'axis_angle' and
                self.model._fused_tables(dev) is not None and
                all(f is g for f, g in zip(frames, self.keyframe_graph)))

    def _slot_map_input(self, slot):
        from ...engine import slam_ops
        from ..utils.opt_pose import axis_angle_translation_to_matrix
        cfg, dev = self.config, self.model.device
        c2w = axis_angle_translation_to_matrix(slot['r'], slot['t'])
        c2w = torch.where(slot['fixed'], c2w.detach(), c2w)
        idx = slam_ops.sample_distinct_dev(slot['n_bank'],
                                           cfg.mapping_sample, dev)
        n = self.camera.height * self.camera.width
        pix = self._distinct(n, slot['bucket'], dev)
        # bank rows + the current frame's pixels + their pose ids: one launch
        rows, ids = slam_ops.coslam_map_rows(
            self._bank, idx, self.num_rays_to_save, pix, self._ray_dirs(),
            slot['rgb'], slot['depth'], slot['cur_id'])
        rays_o, rays_d = slam_ops.PoseRaysFn.apply(c2w, rows, ids)
        return {'rays_o': rays_o, 'rays_d': rays_d, 'target_s': rows[:, 3:6],
                'target_d': rows[:, 6:7], 'first': False, 'sharded': False,
                'n_live': slot['n_live']}

    def _new_slot(self, bucket, K, d_img, c_img):
        cfg = self.config
        dev = torch.device(self.model.device)
        kcap = max(64, 2 * (K + 1))
        r = torch.nn.Parameter(torch.zeros(kcap, 3, device=dev))
        t = torch.nn.Parameter(torch.zeros(kcap, 3, device=dev))
        r.grad, t.grad = torch.zeros_like(r), torch.zeros_like(t)
        fixed = torch.zeros(kcap, 1, 1, dtype=torch.bool, device=dev)
        fixed[0] = True
        pose_opt = Optimizers(dict(cfg.optimizers),
                              {'mapping_pose_r': [r],
                               'mapping_pose_t': [t]})
        opt = pose_opt + self.model_optimizers
        opt.parameters = {**pose_opt.parameters,
                          **self.model_optimizers.parameters}
        opt.static_grads = True
        return {
            'bucket': bucket, 'r': r, 't': t, 'fixed': fixed,
            'pose_opt': pose_opt, 'opt': opt, 'graphs': {},
            'bank_version': self._bank_version,
            'depth': torch.empty_like(d_img),
            'rgb': torch.empty_like(c_img),
            'n_bank': torch.zeros(1, dtype=torch.int64, device=dev),
            'cur_id': torch.zeros(1, dtype=torch.int64, device=dev),
            'n_live': torch.zeros(1, dtype=torch.int32, device=dev)}

    def _load_slot(self, slot, frames, K, d_img, c_img, n_cur):
        from ..engine.optimizers import reset_optimizer_state
        cfg = self.config
        dev = torch.device(self.model.device)
        with torch.no_grad():
            slot['r'].zero_()
            slot['t'].zero_()
            slot['r'][:K + 1] = torch.stack(
                [f.pose.data_r.detach().to(dev) for f in frames])
            slot['t'][:K + 1] = torch.stack(
                [f.pose.data_t.detach().to(dev) for f in frames])
            slot['r'].grad.zero_()
            slot['t'].grad.zero_()
            slot['depth'].copy_(d_img)
            slot['rgb'].copy_(c_img)
            slot['n_bank'].fill_(K * self.num_rays_to_save)
            slot['cur_id'].fill_(K)
            slot['n_live'].fill_(cfg.mapping_sample + n_cur)
        # the reference builds the pose optimisers per call: fresh Adam state
        for o in slot['pose_opt'].optimizers.values():
            reset_optimizer_state(o)

    def _prewarm_slots(self, frames, K, d_img, c_img, bucket=None):
        """Build and capture the graphs of EVERY bucket the first time a
        capacity slot is needed (and again after the bank was re-allocated):
        the current-frame ray count runs through 2048, 1024, 512, 256, 128
        while the first 20 keyframes arrive, and each new bucket used to pay
        two eager iterations + two captures inside whatever frame needed it
        first (a 20-frame timed region right after start-up held four of
        them: 150 frames/s where the steady state is 214).  The warm-up
        iterations run on the call's real data; the model, its optimiser
        state and the random stream are restored afterwards, so the
        trajectory of the run is unchanged."""
        cfg = self.config
        dev = torch.device(self.model.device)
        slots = self.__dict__.setdefault('_pslots', {})
        todo = [b for b in self._BUCKETS
                if b not in slots or len(slots[b]['graphs']) < 2 or
                slots[b]['bank_version'] != self._bank_version or
                K + 1 > slots[b]['r'].shape[0]]
        if slots and bucket is not None:
            # REGROWTH (a slot was outgrown, the bank re-allocated): the ray
            # count only shrinks with the keyframe count, so the buckets above
            # the one in use are never needed again — re-capturing all five
            # cost ~20 iterations + 10 captures inside one frame, a latency
            # spike that grew with the run length.  Only the first call warms
            # every bucket.
            todo = [b for b in todo if b <= bucket]
        if not todo:
            return
        tensors = [p for ps in self.model_optimizers.parameters.values()
                   for p in ps]

        def opt_state():
            return [v for o in self.model_optimizers.optimizers.values()
                    for stt in o.state.values() for v in stt.values()
                    if torch.is_tensor(v)]
        rng = torch.cuda.get_rng_state(dev)
        cpu_rng = torch.get_rng_state()
        saved_p = [p.detach().clone() for p in tensors]
        saved_s = [(v, v.detach().clone()) for v in opt_state()]
        n_iters = cfg.mapping_n_iters
        acc = cfg.optimizers['mapping_pose_r']['optimizer'].accum_step
        steps = (0, (acc or 1) - 1)      # an iteration of each kind
        for b in todo:
            slot = slots[b] = self._new_slot(b, K, d_img, c_img)
            self._load_slot(slot, frames, K, d_img, c_img,
                            min(b, self.camera.height * self.camera.width))
            self._pslot = slot
            self.fixed_shape_batches = True
            try:
                for st in steps:
                    k = self.graph_segment_key(True, st, n_iters)
                    if k in slot['graphs']:
                        continue
                    self._iteration(slot['opt'], frames, True, st, n_iters,
                                    False, None)
                    g = torch.cuda.CUDAGraph()
                    with _capture(g):
                        self._iteration(slot['opt'], frames, True, st,
                                        n_iters, False, None)
                    slot['graphs'][k] = g
                slot['seen'] = set(slot['graphs'])
            finally:
                self._pslot = None
                self.fixed_shape_batches = False
        with torch.no_grad():
            for p, v in zip(tensors, saved_p):
                p.copy_(v)
            known = {id(v) for v, _ in saved_s}
            for v, c in saved_s:
                v.copy_(c)
            for v in opt_state():
                if id(v) not in known:
                    v.zero_()       # created by the warm-up: back to fresh
        torch.cuda.set_rng_state(rng, dev)
        torch.set_rng_state(cpu_rng)

    def _persistent_map(self, n_iters, frames):
        """one mapping call through a capacity slot; False = not usable
        (the caller takes the per-call path)"""
        cfg = self.config
        dev = torch.device(self.model.device)
        K = len(self.keyframe_graph)
        cur = frames[-1]
        n_cur = max(cfg.mapping_sample // K, cfg.min_sample_pixels)
        bucket = next((b for b in self._BUCKETS if b >= n_cur), None)
        if bucket is None:
            return False
        slots = self.__dict__.setdefault('_pslots', {})
        d_img, c_img = cur.device_images(dev)
        if self.model_optimizers is None:
            self.model_optimizers = Optimizers(
                dict(cfg.optimizers), {**self.model.get_param_groups()})
        slot = slots.get(bucket)
        if slot is not None and (
                K + 1 > slot['r'].shape[0] or
                slot['bank_version'] != self._bank_version or
                slot['depth'].shape != d_img.shape):
            slot = None                      # capacities outgrown
        if slot is None and self.prewarm_slots:
            self._prewarm_slots(frames, K, d_img, c_img, bucket)
            slot = slots.get(bucket)
        if slot is None:
            slot = slots[bucket] = self._new_slot(bucket, K, d_img, c_img)
        self._load_slot(slot, frames, K, d_img, c_img, n_cur)
        opt, graphs = slot['opt'], slot['graphs']
        self._pslot = slot
        self.fixed_shape_batches = True
        try:
            seen = slot.setdefault('seen', set())
            for step in range(n_iters):
                # two kinds of iteration: with / without the pose step
                # (5-step gradient accumulation)
                k = self.graph_segment_key(True, step, n_iters)
                if k in graphs:
                    graphs[k].replay()
                elif k not in seen:
                    seen.add(k)        # lazy state (Adam moments) is created
                    self._iteration(opt, frames, True, step, n_iters, False,
                                    None)
                else:
                    g = torch.cuda.CUDAGraph()
                    with _capture(g):
                        self._iteration(opt, frames, True, step, n_iters,
                                        False, None)
                    graphs[k] = g
                    g.replay()
        finally:
            self._pslot = None
            self.fixed_shape_batches = False
        with torch.no_grad():
            rs, ts = slot['r'].detach(), slot['t'].detach()
            dst = [f.pose.data_r for f in frames[1:]] + \
                  [f.pose.data_t for f in frames[1:]]
            src = [x.to(d.device) for x, d in zip(
                list(rs[1:K + 1].unbind(0)) + list(ts[1:K + 1].unbind(0)),
                dst)]
            torch._foreach_copy_(dst, src)
        self.after_mapping_update()
        return True

    def optimize_update(self, n_iters, optimize_frames, is_mapping,
                        coarse=False):
        if is_mapping and self._persistent_map_ok(n_iters, optimize_frames):
            with self.lock:
                if self._persistent_map(n_iters, optimize_frames):
                    return None
        return super().optimize_update(n_iters, optimize_frames, is_mapping,
                                       coarse=coarse)

    def get_loss(self, optimize_frames, is_mapping, step=None, n_iters=None,
                 coarse=False):
        self.model.fixed_shape_losses = getattr(self, 'fixed_shape_batches',
                                                False)
        # tracking steps the pose only: no map gradients are computed
        self.model.map_trainable = bool(is_mapping)
        on_gpu = torch.device(self.model.device).type == 'cuda'
        fused = self.fused_iteration and on_gpu and \
            self.model._fused_tables(self.model.device) is not None
        self.model.fused_losses = fused
        # multi-GPU mapping: per-rank jitter comes from the rank's own RNG
        # stream so that the shared default stream (batch indices, tracking)
        # stays in lock-step although shard sizes may differ by one ray
        if is_mapping and _dist.state.enabled:
            gen = _dist.state.shard_generator
            self.model._rand = lambda shape, like: torch.rand(
                shape, device=like.device, dtype=like.dtype, generator=gen)
        elif '_rand' in self.model.__dict__:
            del self.model._rand
        if fused and not is_mapping:
            inp = self._fused_track_input(optimize_frames[-1])
        elif fused and len(optimize_frames) == len(self.keyframe_graph) + 1:
            inp = self._fused_map_input(optimize_frames)
        else:
            inp = self.get_model_input(optimize_frames, is_mapping)
        out = self.model(inp)
        losses = self.model.get_loss_dict(out, inp, is_mapping, step)
        return functools.reduce(torch.add, losses.values())

    def render_img(self, c2w, gt_depth=None, idx=None):
        with torch.no_grad():
            dev = self.model.device
            rays_o, rays_d = get_rays(self.camera, c2w, device=dev)
            rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
            if gt_depth is not None:
                gt_depth = torch.as_tensor(gt_depth).to(dev).reshape(-1, 1)
            depths, colors = [], []
            bs = self.config.ray_batch_size
            for i in range(0, rays_d.shape[0], bs):
                out = self.model({
                    'rays_o': rays_o[i:i + bs], 'rays_d': rays_d[i:i + bs],
                    'target_s': None,
                    'target_d': None if gt_depth is None
                    else gt_depth[i:i + bs]})
                depths.append(out['depth'].double())
                colors.append(out['rgb'])
            H, W = self.camera.height, self.camera.width
            return torch.cat(colors).reshape(H, W, 3).cpu().numpy(), \
                torch.cat(depths).reshape(H, W).cpu().numpy()

    def get_mesh(self):
        """the SDF zero level set over marching_cubes_bound, coloured by the
        colour head (coslam.py:291-298)"""
        with self.lock:
            if getattr(self, 'mesher', None) is None:
                from ..common.mesher import MesherConfig
                cfg = getattr(self.config, 'mesher', None) or MesherConfig(
                    resolution=256, points_batch_size=30000)
                self.mesher = cfg.setup(
                    camera=self.camera, bounding_box=self.bounding_box,
                    marching_cubes_bound=self.marching_cube_bound)
            self.cur_mesh = self.mesher.get_mesh(
                keyframe_graph=self.keyframe_graph,
                query_fn=self.model.query_fn,
                color_func=self.model.color_func, device=self.device)
            return self.cur_mesh
