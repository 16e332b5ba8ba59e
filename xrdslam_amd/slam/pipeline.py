"""In-process tracking/mapping driver with the per-frame sequence of the
reference's Tracker.spin / Mapper.spin (slam/pipeline/tracker.py:52-199,
slam/pipeline/mapper.py:20-46).  The reference runs tracker, mapper and the
Algorithm in separate processes that alternate strictly through two events
(tracking and mapping never overlap); here the same sequence runs in one
process, which is what the benchmark and the parity tests drive:

    init pose by constant velocity -> Frame -> do_tracking -> set pose ->
    add_framepose -> [map frame?] do_mapping -> update_framepose ->
    [keyframe?] add_keyframe

The multi-process plumbing itself (BaseManager, queues, viewer) is outside the
hot-path scope.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import numpy as np
import torch

from .common.frame import Frame


def predict_current_pose(frame_id, gt_c2w_np, estimate_c2w_list):
    """constant-velocity motion model (tracker.py:185-199)"""
    if frame_id < 1:
        return gt_c2w_np
    prev = estimate_c2w_list[frame_id - 1].detach().cpu().numpy()
    if frame_id == 1:
        return prev
    prev2 = estimate_c2w_list[frame_id - 2].detach().cpu().numpy()
    return (prev @ np.linalg.inv(prev2)) @ prev


def predict_current_pose_device(frame_id, estimate_c2w_list):
    """the same on the device (frame_id >= 1, estimates on the GPU): one
    launch, no host copy of the previous poses"""
    from ..engine import slam_ops
    prev = estimate_c2w_list[frame_id - 1].detach()
    if frame_id == 1:
        return prev.clone()
    return slam_ops.pose_predict(prev,
                                 estimate_c2w_list[frame_id - 2].detach())


class SequentialSLAM:
    def __init__(self, algorithm, dataset, map_every=5, keyframe_every=50,
                 lazy_start=-1, pose_device='cpu', use_relative_pose=False,
                 init_pose_offset=0, device_poses=None):
        self.algorithm, self.dataset = algorithm, dataset
        # MI355X: keep the pose chain of consecutive frames on the device —
        # best pose of the tracking graph -> pose parameters -> constant-
        # velocity start of the next frame — so that the host never waits for
        # a frame's result and the queue does not drain between frames
        # (tracking-only frames were ~35 % GPU idle: 1.7 ms of graph replays,
        # then ~1 ms of host pose bookkeeping behind a device->host read).
        # Needs the poses on the GPU and the persistent tracking graph;
        # None = on exactly then.  Multi-GPU runs broadcast rank 0's result as
        # a device tensor (_sync_pose).
        if device_poses is None:
            device_poses = torch.device(pose_device).type == 'cuda'
        self.device_poses = bool(device_poses)
        self.map_every, self.keyframe_every = map_every, keyframe_every
        self.lazy_start = lazy_start
        self.pose_device = pose_device
        # tracker.py:76-89: poses relative to the first frame, which is placed
        # at identity + init_pose_offset (Vox-Fusion: keeps octree coordinates
        # positive)
        self.use_relative_pose = use_relative_pose
        self.init_pose_offset = init_pose_offset
        self._first_old = self._first_new = None
        self.t_track = 0.0
        self.t_map = 0.0
        Frame.reset_pose_check()

    def is_mapframe(self, fid):
        every = 1 if fid <= self.lazy_start else self.map_every
        return every != -1 and (fid % every == 0 or
                                fid == len(self.dataset) - 1)

    def step(self, idx: int, sync: Optional[Callable[[], None]] = None):
        """process frame ``idx``: track, then (if it is a map frame) map"""
        alg = self.algorithm
        data = self.dataset[idx]
        gt_c2w = data['c2w'].astype(np.float64)
        if self.use_relative_pose:
            if idx == 0 or self._first_old is None:
                self._first_old = gt_c2w
                self._first_new = np.eye(4)
                self._first_new[:3, 3] += self.init_pose_offset
                gt_c2w = self._first_new
            else:
                gt_c2w = self._first_new @ (np.linalg.inv(self._first_old) @
                                            gt_c2w)
        gt_c2w = gt_c2w.astype(np.float32)
        est = alg.get_estimate_c2w_list()
        on_device = self._device_chain()
        alg.device_track_result = on_device
        if on_device and idx >= 1 and est[idx - 1].is_cuda and \
                (idx < 2 or est[idx - 2].is_cuda):
            init = predict_current_pose_device(idx, est)
        else:
            init = predict_current_pose(idx, gt_c2w, est)
        frame = Frame(fid=idx, rgb=data['rgb'], depth=data['depth'],
                      gt_pose=gt_c2w, init_pose=init,
                      separate_LR=alg.is_separate_LR(),
                      rot_rep=alg.get_rot_rep(), device=self.pose_device)
        if 'depth_dev' in data:
            # images already resident in HBM (device-resident frame store)
            frame._dev_cache = (data['depth_dev'], data['rgb_dev'])
        t0 = time.perf_counter()
        cand = alg.do_tracking(frame)
        cand = self._sync_pose(cand)
        if alg.is_initialized() and cand is not None:
            frame.set_pose(cand, separate_LR=alg.is_separate_LR(),
                           rot_rep=alg.get_rot_rep())
        if sync:
            sync()
        t1 = time.perf_counter()
        g = torch.from_numpy(gt_c2w)
        alg.add_framepose(frame.get_pose().detach(), g, g.clone())
        if self.is_mapframe(idx):
            frame.is_final_frame = idx == len(self.dataset) - 1
            alg.do_mapping(frame)
            # the deferred initial-pose check, without a host wait
            Frame.poll_inconsistent()
            alg.update_framepose(idx, frame.get_pose().detach())
            if idx % self.keyframe_every == 0:
                alg.add_keyframe(frame)
            if sync:
                sync()
        t2 = time.perf_counter()
        self.t_track += t1 - t0
        self.t_map += t2 - t1
        return frame

    def _device_chain(self):
        alg = self.algorithm
        # (multi-GPU: the chain stays on the device too — rank 0's result is
        # broadcast as a device tensor, _sync_pose)
        return self.device_poses and \
            bool(getattr(alg, 'use_graphs', False)) and \
            bool(getattr(alg, 'persistent_track_graph', False)) and \
            torch.device(alg.device).type == 'cuda' and \
            torch.device(self.pose_device).type == 'cuda'

    def _sync_pose(self, cand):
        """multi-GPU: tracking is replicated; rank 0's result is broadcast so
        that every rank continues from bit-identical poses (frustum masks and
        therefore all-reduce bucket sizes must agree across ranks)"""
        from ..engine import dist as _dist
        if not _dist.state.enabled or cand is None:
            return cand
        import torch.distributed as dist
        if torch.is_tensor(cand):
            # the device pose chain: the 4x4 is broadcast where it lives (one
            # RCCL broadcast of 64 bytes over xGMI), no host hop
            # (a copy: ``cand`` is the tracking graph's static best-pose
            # buffer — the broadcast must not write into graph-owned memory
            # and the result must survive the next replay)
            t = cand.detach().float().clone()
            dist.broadcast(t, src=0)
            return t
        dev = self.algorithm.device
        t = torch.as_tensor(cand, dtype=torch.float32).to(dev).contiguous()
        dist.broadcast(t, src=0)
        return t.cpu().numpy()

    def trajectory_stats(self, align=True, correct_scale=False):
        """ATE statistics of the frames processed so far, like ds-eval prints
        them (utils/eval_traj.py); ``align=False`` = ate_rmse()"""
        from .utils.eval_traj import evaluate_trajectory
        Frame.raise_if_inconsistent()
        alg = self.algorithm
        n = len(alg.get_estimate_c2w_list())
        return evaluate_trajectory(alg.get_gt_c2w_list(),
                                   alg.get_estimate_c2w_list(), n,
                                   correct_scale=correct_scale, align=align)

    def save_eval_tar(self, path):
        """the trajectory file the reference's tracker leaves in its output
        directory (tracker.py:411-420)"""
        from .utils.eval_traj import save_eval_tar
        Frame.raise_if_inconsistent()
        save_eval_tar(self.algorithm,
                      len(self.algorithm.get_estimate_c2w_list()), path)

    def ate_rmse(self):
        """translation RMSE between estimated and GT poses (no alignment: the
        synthetic runs start from the GT pose of frame 0)"""
        Frame.raise_if_inconsistent()
        est = torch.stack([p[:3, 3].cpu() for p in
                           self.algorithm.get_estimate_c2w_list()])
        gt = torch.stack([p[:3, 3] for p in self.algorithm.get_gt_c2w_list()])
        return float(((est - gt)**2).sum(1).mean().sqrt())
