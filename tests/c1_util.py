"""Trajectory-parity fixtures (tests/golden/c1_<algo>.npz, made by
oracle/make_golden_c1.py from the REFERENCE's own Algorithm classes run over a
short synthetic sequence on the CPU): rebuild the sequence, run the engine on
it the way a user would (graphs, fused iterations, device pose chain), and
compare ATE statistics.  ``c1_coslam`` is BASELINE.json configs[0]."""
import os
import random

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden')

ALGO = {'coslam': 'co-slam', 'voxfusion': 'vox-fusion',
        'pointslam': 'point-slam', 'nice': 'nice-slam', 'splatam': 'splaTAM'}


def fixture(name):
    return np.load(os.path.join(GOLDEN, f'c1_{name}.npz'), allow_pickle=False)


def ref_stats(g):
    """(per-seed ATE [m], per-seed per-frame translation error [seeds, n])"""
    seeds = sorted(int(k[4:]) for k in g.files if k.startswith('est/'))
    gt = g['gt'][:, :3, 3]
    err = np.stack([np.linalg.norm(g[f'est/{s}'][:, :3, 3] - gt, axis=1)
                    for s in seeds])
    return np.sqrt((err**2).mean(1)), err


def frozen_ate(gt):
    """ATE of a "tracker" that never moves: every estimate = the pose of frame
    0.  The yardstick a fixture has to beat to certify TRACKING: a loop whose
    error is not well below it (round 5's NICE-SLAM / Point-SLAM / SplaTAM
    fixtures were above it) says nothing about the tracker."""
    t = np.asarray(gt)[:, :3, 3]
    return float(np.sqrt(((t - t[0])**2).sum(1).mean()))


def room(g, dev):
    from xrdslam_amd.data.synthetic import SyntheticRoom
    fx, fy, cx, cy, W, H = (float(v) for v in g['seq/intrinsics'])
    kw = {}
    if 'seq/shrink' in g.files:
        kw['shrink'] = float(g['seq/shrink'])
    return SyntheticRoom(g['seq/bound'].tolist(), H=int(H), W=int(W), fx=fx,
                         fy=fy, cx=cx, cy=cy,
                         n_frames=int(g['seq/n_frames']), device=dev, **kw)


class _HostImages:
    """numpy images (Point-SLAM's colour-gradient pixel choice and SplaTAM's
    seeding read them on the host) and, for SplaTAM, OpenCV-convention poses
    (camera looks down +z)"""

    def __init__(self, data, cv):
        self.data, self.cv = data, cv

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        d = dict(self.data[i])
        if self.cv:
            c2w = np.array(d['c2w'], dtype=np.float64)
            c2w[:3, 1] *= -1
            c2w[:3, 2] *= -1
            d['c2w'] = c2w
        for k in ('rgb', 'depth'):
            if torch.is_tensor(d[k]):
                d[k] = d[k].cpu().numpy()
        return d


def camera(g):
    from xrdslam_amd.slam.common.camera import Camera
    fx, fy, cx, cy, W, H = (float(v) for v in g['seq/intrinsics'])
    return Camera(fx, fy, cx, cy, int(W), int(H))


def overrides(g, cfg):
    """reduced iteration / ray counts a fixture was generated with
    (``cfg/<field>`` scalars; absent = the reference's input_config values)"""
    for k in g.files:
        if k.startswith('cfg/'):
            v = g[k].item()
            setattr(cfg, k[4:], type(getattr(cfg, k[4:]))(v))
    return cfg


def run_engine(name, seed, dev='cuda', n_frames=None, configure=None,
               **slam_kw):
    """-> (est [n,4,4], gt [n,4,4], seconds, slam)"""
    import time

    from xrdslam_amd.slam.configs import input_config as ic
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    g = fixture(name)
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    bound = g['seq/bound'].tolist()
    key = ALGO[name]
    make = ic.algorithm_configs[key]
    cfg = make(bound) if name in ('coslam', 'nice') else make()
    overrides(g, cfg)
    if configure:
        configure(cfg)
    algo = cfg.setup(camera=camera(g), device=dev)
    data = room(g, dev)
    if name in ('pointslam', 'splatam'):
        data = _HostImages(data, cv='seq/cv_poses' in g.files)
    cad = ic.cadence[key]
    kw = dict(map_every=cad.map_every, keyframe_every=cad.keyframe_every,
              lazy_start=cad.lazy_start, pose_device=dev,
              use_relative_pose=cad.use_relative_pose,
              init_pose_offset=cad.init_pose_offset)
    if 'cad/lazy_start' in g.files:
        kw['lazy_start'] = int(g['cad/lazy_start'])
    kw.update(slam_kw)
    slam = SequentialSLAM(algo, data, **kw)
    n = n_frames or int(g['seq/run_frames'] if 'seq/run_frames' in g.files
                        else g['seq/n_frames'])
    t0 = time.perf_counter()
    for k in range(n):
        slam.step(k)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    from xrdslam_amd.slam.common.frame import Frame
    Frame.raise_if_inconsistent()   # the deferred initial-pose check
    est = torch.stack([p.detach().cpu().float() for p in
                       algo.get_estimate_c2w_list()[:n]]).numpy()
    gt = torch.stack([p.cpu().float() for p in
                      algo.get_gt_c2w_list()[:n]]).numpy()
    return est, gt, sec, slam


def ate(est, gt):
    return float(np.sqrt(((est[:, :3, 3] - gt[:, :3, 3])**2).sum(1).mean()))
