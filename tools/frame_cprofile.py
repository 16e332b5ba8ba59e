"""cProfile of single map frames of the default NICE-SLAM bench loop (host
side), to explain slow outliers.  usage: frame_cprofile.py 10 15 20"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from xrdslam_amd.data.synthetic import SyntheticRoom  # noqa: E402
from xrdslam_amd.engine import dist as xdist  # noqa: E402
from xrdslam_amd.slam.common.camera import Camera  # noqa: E402
from xrdslam_amd.slam.configs.input_config import (cadence,  # noqa: E402
                                                   nice_slam_config)
from xrdslam_amd.slam.pipeline import SequentialSLAM  # noqa: E402

which = [int(a) for a in sys.argv[1:]] or [10, 15, 20]
dev = torch.device('cuda:0')
torch.manual_seed(0)
np.random.seed(0)
cfg = nice_slam_config(bench.BOUND)
cam = Camera(**bench.CAM)
algo = cfg.setup(camera=cam, device=str(dev))
algo.use_graphs = True
xdist.state.setup(dev, seed=0)
data = SyntheticRoom(bench.BOUND, H=cam.height, W=cam.width, fx=cam.fx,
                     fy=cam.fy, cx=cam.cx, cy=cam.cy, n_frames=200, device=dev)
cad = cadence['nice-slam']
slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                      keyframe_every=cad.keyframe_every, pose_device=str(dev))
for k in range(max(which) + 1):
    if k in which:
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        slam.step(k)
        pr.disable()
        dt = (time.perf_counter() - t0) * 1e3
        s = io.StringIO()
        key = os.environ.get('XRD_CP_SORT', 'cumulative')
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
        lines = [ln for ln in s.getvalue().splitlines() if '/' in ln or
                 '{' in ln]
        print(f'--- frame {k}: {dt:.1f} ms')
        print('\n'.join(ln[:150] for ln in lines[:28]))
    else:
        slam.step(k)
