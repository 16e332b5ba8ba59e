"""per-frame wall times of the default NICE-SLAM bench loop (to find cold-start
effects: run twice on a fresh box and compare the map frames)."""
import os
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
out = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 26):
    if k == 6 and os.environ.get('XRD_FT_PROFILE'):
        from xrdslam_amd.engine import nice as en
        en.PROFILE = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    slam.step(k)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) * 1e3)
print(' '.join(f'{k}:{t:.1f}' for k, t in enumerate(out)))
