"""GPU: Point-SLAM on the HIP grid kNN.  (1) every stage of the golden made
from the reference's own model with xrd_knn_* as the neighbour search;
(2) a short PointSLAM run on the synthetic room (random-initialised decoders:
the pretrained checkpoint is a git-LFS pointer in the reference tree, so this
checks the loop — growth, frustum masks, dynamic radii, stages — not map
quality)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import pointslam_golden_util as pg  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_point_slam_model_vs_reference():
    g = np.load(pg.GOLDEN)
    errs = pg.run(g, 'cuda:0')

    def tol(k):
        if k.startswith('track/') and ('g_dec' in k or 'loss' in k):
            # the tracking loss divides by sqrt(rendered variance), a sum of
            # w (z - depth)^2 over 5 samples within 2 % of the depth: its
            # rounding (torch on the CPU for the golden, torch on the GPU
            # here) is amplified into the loss (1.2e-4 measured) and the
            # geometry-decoder gradients (up to 4.8e-4).  Everything else, the ray
            # gradients through the 1/d^2 weights included, holds 1e-4.
            return 5e-4
        return TOL
    bad = {k: v for k, v in errs.items() if not v < tol(k)}
    assert not bad, bad


def test_pointslam_loop_runs_on_synthetic_room():
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import (cadence,
                                                       pointslam_config)
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    bound = [[-3, 3], [-4, 2.5], [-2, 2.5]]
    cam = Camera(fx=150., fy=150., cx=79.5, cy=59.5, width=160, height=120)
    cfg = pointslam_config()
    cfg.mapping_first_n_iters, cfg.mapping_n_iters = 60, 20
    cfg.tracking_n_iters = 10
    cfg.tracking_Wedge = cfg.tracking_Hedge = 10
    cfg.pixels_adding, cfg.mapping_pixels_based_on_color_grad = 1500, 200
    cfg.tracking_sample, cfg.mapping_sample = 400, 1000
    algo = cfg.setup(camera=cam, device='cuda:0')

    class Np:  # the algorithm reads numpy images (sobel on the host)
        def __init__(self, d):
            self.d = d

        def __len__(self):
            return len(self.d)

        def __getitem__(self, i):
            x = dict(self.d[i])
            for k in ('rgb', 'depth'):
                if torch.is_tensor(x[k]):
                    x[k] = x[k].cpu().numpy()
            return x

    data = Np(SyntheticRoom(bound, H=120, W=160, fx=150., fy=150., cx=79.5,
                            cy=59.5, n_frames=200, device='cuda:0'))
    cad = cadence['point-slam']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every, lazy_start=2,
                          pose_device='cuda:0')
    n_pts = []
    for k in range(7):
        slam.step(k)
        n_pts.append(algo.model.neural_point_cloud.pts_num())
    assert n_pts[0] > 3000 and n_pts[-1] > n_pts[0]      # the cloud grows
    npc = algo.model.neural_point_cloud
    assert npc.geo_feats.shape == (n_pts[-1], 32)
    assert npc.frustum_mask.shape == (n_pts[-1], 1)
    assert np.isfinite(slam.ate_rmse()) and slam.ate_rmse() < 0.2
    rgb, depth = algo.render_img(algo.get_estimate_c2w_list()[5].to('cuda:0'),
                                 gt_depth=data[5]['depth'], idx=5)
    gt = data[5]['depth']
    assert np.isfinite(depth).all() and np.abs(depth - gt)[gt > 0].mean() < 0.2
