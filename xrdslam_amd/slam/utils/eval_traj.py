"""Trajectory evaluation of a finished run: the ``eval.tar``
file the reference's tracker writes (slam/pipeline/tracker.py:269-278,411-420)
and ``ds-eval`` reads (scripts/eval.py:33-54), and the absolute trajectory
error after a closed-form rigid (optionally similarity) alignment
(scripts/utils/eval_ate.py:63-120,299-318).  numpy/torch on the host; not part
of the per-frame path."""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch

EVAL_KEYS = ('gt_c2w_list_ori', 'gt_c2w_list', 'estimate_c2w_list', 'idx')


def save_eval_tar(algorithm, idx, path: str) -> None:
    """the four entries ds-eval expects, in the legacy (non-zip) torch format
    the reference writes"""
    torch.save({'gt_c2w_list_ori': algorithm.get_gt_c2w_list_ori(),
                'gt_c2w_list': algorithm.get_gt_c2w_list(),
                'estimate_c2w_list': algorithm.get_estimate_c2w_list(),
                'idx': torch.as_tensor(idx)},
               path, _use_new_zipfile_serialization=False)


def load_eval_tar(path: str) -> Dict:
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    missing = [k for k in EVAL_KEYS if k not in ckpt]
    if missing:
        raise KeyError(f'{path}: not an eval.tar (missing {missing})')
    return ckpt


def _positions(c2w_list: Sequence, n: int) -> np.ndarray:
    return np.stack([np.asarray(torch.as_tensor(c2w_list[i]).detach().cpu()
                                .double())[:3, 3] for i in range(n)])


def valid_pose_mask(gt_c2w_list: Sequence, n: int) -> np.ndarray:
    """ground-truth poses with inf/nan entries (ScanNet) are left out, like
    convert_poses (eval_ate.py:321-339)"""
    return np.array([bool(torch.isfinite(torch.as_tensor(gt_c2w_list[i]))
                          .all()) for i in range(n)])


def align_trajectories(est: np.ndarray, gt: np.ndarray, correct_scale=False):
    """least-squares rotation / translation (/ scale) taking ``est`` [n,3] onto
    ``gt`` [n,3] (Horn's closed form through the SVD of the cross-covariance,
    eval_ate.py:63-120).  Returns rot [3,3], trans [3], scale, and the
    per-pose distance after alignment [n]."""
    est = np.asarray(est, dtype=np.float64)
    gt = np.asarray(gt, dtype=np.float64)
    mu_e, mu_g = est.mean(0), gt.mean(0)
    e0, g0 = est - mu_e, gt - mu_g
    # sum_i e_i g_i^T, decomposed transposed like the reference does
    U, _, Vh = np.linalg.svd((e0.T @ g0).T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1.0
    rot = U @ S @ Vh
    scale = 1.0
    if correct_scale:
        scale = float(((e0 @ rot.T) * g0).sum() / (e0 * e0).sum())
    trans = mu_g - scale * rot @ mu_e
    err = np.linalg.norm(scale * est @ rot.T + trans - gt, axis=1)
    return rot, trans, scale, err


def evaluate_trajectory(gt_c2w_list: Sequence, estimate_c2w_list: Sequence,
                        n: int, correct_scale: bool = False,
                        align: bool = True) -> Dict:
    """the statistics ds-eval prints (eval_ate.py:283-306); ``align=False``
    compares in the common world frame (the synthetic runs start from the
    ground-truth pose)"""
    n = int(n)
    mask = valid_pose_mask(gt_c2w_list, n)
    gt = _positions(gt_c2w_list, n)[mask]
    est = _positions(estimate_c2w_list, n)[mask]
    if len(gt) < 2:
        raise ValueError('need at least two valid pose pairs')
    if align:
        rot, trans, scale, err = align_trajectories(est, gt, correct_scale)
    else:
        rot, trans, scale = np.eye(3), np.zeros(3), 1.0
        err = np.linalg.norm(est - gt, axis=1)
    return {'compared_pose_pairs': len(err),
            'absolute_translational_error.rmse':
                float(np.sqrt(np.dot(err, err) / len(err))),
            'absolute_translational_error.mean': float(err.mean()),
            'absolute_translational_error.median': float(np.median(err)),
            'absolute_translational_error.std': float(err.std()),
            'absolute_translational_error.min': float(err.min()),
            'absolute_translational_error.max': float(err.max()),
            'rot': rot, 'trans': trans, 'scale': scale}


def evaluate_eval_tar(path: str, correct_scale: bool = False) -> Dict:
    """what ``ds-eval --eval-traj`` computes from an output directory's
    eval.tar (scripts/eval.py:41-54: estimates against ``gt_c2w_list_ori``)"""
    ckpt = load_eval_tar(path)
    return evaluate_trajectory(ckpt['gt_c2w_list_ori'],
                               ckpt['estimate_c2w_list'], int(ckpt['idx']),
                               correct_scale)
