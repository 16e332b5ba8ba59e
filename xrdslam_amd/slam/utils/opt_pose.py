"""``OptimizablePose``: SE(3) pose as a learnable [t, axis-angle] (6) or
[t, quaternion(r,i,j,k)] (7) vector, optionally split into separate R / t
parameters — interface of the reference (slam/utils/opt_pose.py:13-110).

The reference imports pytorch3d for three conversions; they are restated here
from the published formulas so the class has no
third-party dependency.  Parameters live on ``device`` (the reference: CPU).
"""
from __future__ import annotations

from copy import deepcopy

import torch
import torch.nn as nn


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """q = (r, i, j, k), not necessarily unit: s = 2/|q|^2 (C.5)."""
    r, i, j, k = torch.unbind(q, -1)
    s = 2.0 / (q * q).sum(-1)
    rows = torch.stack([
        1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
        s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
        s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)
    ], -1)
    return rows.reshape(q.shape[:-1] + (3, 3))


def matrix_to_quaternion(R: torch.Tensor) -> torch.Tensor:
    """rotation matrix -> unit quaternion (r,i,j,k), numerically stable branch
    on the largest of the four squared components."""
    m = R
    t = [1 + m[0, 0] + m[1, 1] + m[2, 2], 1 + m[0, 0] - m[1, 1] - m[2, 2],
         1 - m[0, 0] + m[1, 1] - m[2, 2], 1 - m[0, 0] - m[1, 1] + m[2, 2]]
    best = int(torch.argmax(torch.stack([x.detach() for x in t])))
    d = 2.0 * torch.sqrt(torch.clamp(t[best], min=1e-12))
    if best == 0:
        q = [d / 4, (m[2, 1] - m[1, 2]) / d, (m[0, 2] - m[2, 0]) / d,
             (m[1, 0] - m[0, 1]) / d]
    elif best == 1:
        q = [(m[2, 1] - m[1, 2]) / d, d / 4, (m[0, 1] + m[1, 0]) / d,
             (m[0, 2] + m[2, 0]) / d]
    elif best == 2:
        q = [(m[0, 2] - m[2, 0]) / d, (m[0, 1] + m[1, 0]) / d, d / 4,
             (m[1, 2] + m[2, 1]) / d]
    else:
        q = [(m[1, 0] - m[0, 1]) / d, (m[0, 2] + m[2, 0]) / d,
             (m[1, 2] + m[2, 1]) / d, d / 4]
    q = torch.stack(q)
    if q[0] < 0:
        q = -q
    return q


def quaternion_to_axis_angle(q: torch.Tensor) -> torch.Tensor:
    """theta = 2 atan2(|v|, r); axis_angle = v * theta/|v| (limit 2 v)."""
    v = q[..., 1:]
    n = torch.linalg.norm(v, dim=-1, keepdim=True)
    theta = 2.0 * torch.atan2(n, q[..., :1])
    small = n < 1e-8
    scale = torch.where(small, torch.full_like(n, 2.0),
                        theta / torch.where(small, torch.ones_like(n), n))
    return v * scale


def axis_angle_translation_to_matrix(rot: torch.Tensor,
                                     trans: torch.Tensor) -> torch.Tensor:
    """batched OptimizablePose.matrix() for rot_rep='axis_angle': rot [n,3],
    trans [n,3] -> c2w [n,4,4]; same formula (Rodrigues, exact identity below
    1e-8 rad) evaluated for all poses at once"""
    if rot.is_cuda and rot.dtype == torch.float32:
        from ...engine import slam_ops
        return slam_ops.PoseAxisAngleFn.apply(rot, trans)
    n = rot.shape[0]
    small = torch.norm(rot.detach(), dim=-1, keepdim=True) <= 1e-8
    safe = torch.where(small, torch.ones_like(rot), rot)
    angle = torch.norm(safe, dim=-1, keepdim=True)
    w = safe / angle
    z = torch.zeros_like(w[:, 0])
    K = torch.stack([z, -w[:, 2], w[:, 1], w[:, 2], z, -w[:, 0], -w[:, 1],
                     w[:, 0], z], -1).reshape(n, 3, 3)
    eye = torch.eye(3, device=rot.device, dtype=rot.dtype).expand(n, 3, 3)
    s, c = torch.sin(angle)[..., None], torch.cos(angle)[..., None]
    R = eye + K * s + (1. - c) * (K @ K)
    R = torch.where(small[..., None], eye, R)
    top = torch.cat([R, trans[..., None]], -1)
    bottom = torch.tensor([0., 0., 0., 1.], device=rot.device,
                          dtype=rot.dtype).expand(n, 1, 4)
    return torch.cat([top, bottom], 1)


class OptimizablePose(nn.Module):
    def __init__(self, init_pose, separate_LR=True, rot_rep='axis_angle'):
        super().__init__()
        self.separate_LR = separate_LR
        self.rot_rep = rot_rep
        if rot_rep not in ('axis_angle', 'quat'):
            print('Not support rotation represion: ', rot_rep)
        if separate_LR:
            rname = 'data_r' if rot_rep == 'axis_angle' else 'data_q'
            # (clones: two parameters must not be views of one 7-float
            # storage — torch.save would write it whole, and the rotation
            # parameter would sit at a 12-byte offset)
            self.register_parameter(rname,
                                    nn.Parameter(init_pose[3:].clone()))
            self.register_parameter('data_t',
                                    nn.Parameter(init_pose[:3].clone()))
        else:
            self.register_parameter('data', nn.Parameter(init_pose))

    def _rot_param(self):
        if self.separate_LR:
            return self.data_r if self.rot_rep == 'axis_angle' else self.data_q
        return self.data[3:]

    def copy_from(self, pose):
        for name, _ in list(self.named_parameters()):
            setattr(self, name, deepcopy(getattr(pose, name)))

    def matrix(self):
        if self.rot_rep == 'quat':
            # MI355X: one fused launch (fwd) + one (bwd) instead of ~25 + ~40
            first = self.data_q if self.separate_LR else self.data
            if first.is_cuda and first.dtype == torch.float32:
                from ...engine import slam_ops
                if self.separate_LR:
                    return slam_ops.PoseQuatSplitFn.apply(self.data_t,
                                                          self.data_q)
                return slam_ops.PoseQuat7Fn.apply(self.data)
        if self.rot_rep == 'axis_angle' and self.separate_LR and \
                self.data_r.is_cuda and self.data_r.dtype == torch.float32:
            from ...engine import slam_ops
            return slam_ops.PoseAxisAngleFn.apply(
                self.data_r.unsqueeze(0), self.data_t.unsqueeze(0))[0]
        rot, t = self.rotation(), self.translation()
        Rt = torch.eye(4, device=t.device, dtype=t.dtype)
        Rt[:3, :3] = rot
        Rt[:3, 3] = t
        return Rt

    def rotation(self):
        if self.rot_rep == 'axis_angle':
            return self.axis_angle_to_rotation_matrix(self._rot_param())
        return quaternion_to_matrix(self._rot_param())

    def translation(self):
        return self.data_t if self.separate_LR else self.data[:3]

    @staticmethod
    def axis_angle_to_rotation_matrix(angle_axis):
        """Rodrigues; exactly I when the angle is (all)close to 0
        (opt_pose.py:78-95).  Written without the reference's host-side
        ``torch.allclose`` branch so that it can be captured in a hipGraph:
        the small-angle case is selected with ``torch.where`` (same values,
        zero gradient there, like the constant I of the reference)."""
        eye = torch.eye(3, device=angle_axis.device, dtype=angle_axis.dtype)
        small = torch.norm(angle_axis.detach(), dim=-1, keepdim=True) <= 1e-8
        safe = torch.where(small, torch.ones_like(angle_axis), angle_axis)
        angle = torch.norm(safe, dim=-1, keepdim=True)
        w = safe / angle
        z = torch.zeros_like(w[0])
        K = torch.stack([torch.stack([z, -w[2], w[1]]),
                         torch.stack([w[2], z, -w[0]]),
                         torch.stack([-w[1], w[0], z])])
        R = eye + K * torch.sin(angle) + (1. - torch.cos(angle)) * (K @ K)
        return torch.where(small, eye, R)

    @classmethod
    def from_matrix(cls, Rt, separate_LR=True, rot_rep='axis_angle'):
        # the conversion branches on values (largest quaternion component,
        # sign): on a device tensor that is two host syncs and ~15 tiny
        # launches per frame — evaluated on the host, ONE upload of the
        # parameter vector (same float32 arithmetic)
        dev = Rt.device
        Rc = Rt.detach().to('cpu') if dev.type != 'cpu' else Rt
        R, u = Rc[:3, :3], Rc[:3, 3]
        quat = matrix_to_quaternion(R)
        rot = quaternion_to_axis_angle(quat) if rot_rep == 'axis_angle' \
            else quat
        vec = torch.cat([u, rot], dim=-1).detach().clone()
        return cls(vec.to(dev), separate_LR=separate_LR, rot_rep=rot_rep)
