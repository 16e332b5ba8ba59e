"""``Frame``: one RGB-D observation + its optimisable pose (reference:
slam/common/frame.py:10-74).  ``rgb``/``depth`` stay numpy like the reference
hands them over; the engine keeps a device-resident copy per frame
(``device_images``) so that the per-iteration whole-image H2D copy of
slam/common/common.py:67-68 happens once per frame."""
from typing import List

import torch
import torch.nn as nn
from torch.nn import Parameter

from ..utils.opt_pose import OptimizablePose


class Frame(nn.Module):
    # device -> largest |initial pose - its parameterisation| since the last
    # raise_if_inconsistent() (device tensors: see set_pose)
    _pose_check = {}

    # device -> (pinned host copy, event) of an asynchronous read-out in flight
    _pose_poll = {}

    @staticmethod
    def reset_pose_check():
        """a new run starts with a clean slate (a pipeline calls this when it
        is constructed: an inconsistency another run or test left unraised
        must not surface here)"""
        for err in Frame._pose_check.values():
            err.zero_()
        Frame._pose_poll.clear()

    @staticmethod
    def poll_inconsistent(atol=1e-3):
        """the deferred check without a host wait: reads the deviation an
        EARLIER call copied out asynchronously (raises if it was too large),
        then starts the next copy.  A pipeline calls it at mapping frames, so
        a bad initial pose is reported at most one mapping interval late even
        when nobody reads the trajectory."""
        for dev, err in Frame._pose_check.items():
            if dev.type != 'cuda':
                continue
            hit = Frame._pose_poll.get(dev)      # [pinned host float, event]
            if hit is None:
                hit = Frame._pose_poll[dev] = [
                    torch.empty(1, dtype=torch.float32).pin_memory(), None]
            if hit[1] is not None:
                if not hit[1].query():
                    continue                     # the copy is still in flight
                e, hit[1] = float(hit[0]), None
                if not e <= atol:
                    err.zero_()
                    raise ValueError('Transformation inconsistency detected! '
                                     f'(largest deviation {e:g} on {dev})')
            hit[0].copy_(err, non_blocking=True)
            hit[1] = torch.cuda.Event()
            hit[1].record(torch.cuda.current_stream(dev))

    @staticmethod
    def raise_if_inconsistent(atol=1e-3):
        """the deferred half of the initial-pose check of device-resident
        poses (one host read for all frames since the last call)"""
        for dev, err in Frame._pose_check.items():
            e = float(err)
            err.zero_()
            if not e <= atol:   # also catches NaN
                raise ValueError('Transformation inconsistency detected! '
                                 f'(largest deviation {e:g} on {dev})')

    def __init__(self, fid, rgb, depth, init_pose=None, gt_pose=None,
                 separate_LR=False, rot_rep='axis_angle',
                 device='cpu') -> None:
        super().__init__()
        self.fid = fid
        self.h, self.w = (depth.shape if depth is not None else rgb.shape[:2])
        self.rgb, self.depth, self.gt_pose = rgb, depth, gt_pose
        self.separate_LR, self.rot_rep = separate_LR, rot_rep
        self.is_final_frame = False
        self.pose_device = device
        self._dev_cache = None
        self.pose = None
        if init_pose is not None:
            self.set_pose(init_pose, separate_LR, rot_rep, check=True)

    def set_pose(self, pose_np, separate_LR=False, rot_rep='axis_angle',
                 check=False):
        """pose parameters from a 4x4 matrix.  Built (and, for the initial
        pose, checked: frame.py:24-29) on the host, then moved to the pose
        device with one upload per parameter — on the device the conversion
        costs two host syncs and ~15 tiny launches per frame."""
        if torch.is_tensor(pose_np) and pose_np.is_cuda and \
                torch.device(self.pose_device).type == 'cuda':
            # a pose that is already on the device (the tracking graph's best
            # pose, the device-side constant-velocity start) stays there: one
            # conversion launch instead of a host round trip that would drain
            # the queue between two frames
            from ...engine import slam_ops
            worst = None
            if check:
                # the initial pose's consistency check of the reference
                # (frame.py:24-29) WITHOUT its host read: the conversion
                # launch folds the deviation into a device float that is
                # raised at the next point that reads poses back anyway
                # (Frame.raise_if_inconsistent: the trajectory readers of the
                # pipeline) — a read here would drain the queue once a frame
                # (measured: Co-SLAM 333 -> 306 frames/s)
                key = torch.device(self.pose_device)
                worst = Frame._pose_check.get(key)
                if worst is None:
                    worst = Frame._pose_check[key] = torch.zeros(
                        1, dtype=torch.float32, device=key)
            vec = slam_ops.pose_from_matrix(pose_np.to(self.pose_device),
                                            rot_rep, dev_max=worst)
            self.pose = OptimizablePose(vec, separate_LR=separate_LR,
                                        rot_rep=rot_rep)
            return
        Rt = torch.as_tensor(pose_np, dtype=torch.float32).cpu()
        pose = OptimizablePose.from_matrix(Rt, separate_LR=separate_LR,
                                           rot_rep=rot_rep)
        if check and not torch.allclose(Rt, pose.matrix().detach(),
                                        atol=1e-3):
            raise ValueError('Transformation inconsistency detected!', Rt,
                             pose.matrix())
        self.pose = pose.to(self.pose_device)

    def get_pose(self):
        return self.pose.matrix()

    def get_translation(self):
        return self.pose.translation()

    def get_rotation(self):
        return self.pose.rotation()

    def get_params(self) -> List[Parameter]:
        if self.pose is None:
            return []
        if self.separate_LR:
            r = self.pose.data_q if self.rot_rep == 'quat' else \
                self.pose.data_r
            return [r, self.pose.data_t]
        return list(self.pose.parameters())

    def device_images(self, device):
        """(depth [H*W,1] f32, rgb [H*W,3] f32) on ``device``, uploaded once"""
        if self._dev_cache is None or self._dev_cache[0].device != \
                torch.device(device):
            d = torch.as_tensor(self.depth, dtype=torch.float32).to(device)
            c = torch.as_tensor(self.rgb, dtype=torch.float32).to(device)
            self._dev_cache = (d.reshape(-1, 1), c.reshape(-1, 3))
            self._dev_chw = None
        return self._dev_cache

    def device_rgb_chw(self, device):
        """rgb as [3,H,W] f32 on ``device`` (SplaTAM's image-space losses),
        permuted once per frame"""
        _, c = self.device_images(device)
        if getattr(self, '_dev_chw', None) is None:
            self._dev_chw = c.reshape(self.h, self.w, 3).permute(2, 0, 1) \
                .contiguous()
        return self._dev_chw
