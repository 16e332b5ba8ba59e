"""host-side cProfile of a few frames of one algorithm (where do the
milliseconds go?):  python tools/profile_host_algo.py vox-fusion"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from xrdslam_amd.data.synthetic import SyntheticRoom
from xrdslam_amd.slam.common.camera import Camera
from xrdslam_amd.slam.configs.input_config import algorithm_configs, cadence
from xrdslam_amd.slam.pipeline import SequentialSLAM

name = sys.argv[1] if len(sys.argv) > 1 else 'vox-fusion'
dev = 'cuda:0'
torch.manual_seed(0)
np.random.seed(0)
cam = Camera(**bench.CAM)
cfg = algorithm_configs[name]()
algo = cfg.setup(camera=cam, device=dev)
algo.use_graphs = name in ('co-slam', 'nice-slam', 'vox-fusion')
data = SyntheticRoom(bench.CO_BOUND, H=cam.height, W=cam.width, fx=cam.fx,
                     fy=cam.fy, cx=cam.cx, cy=cam.cy, n_frames=200, device=dev)
if name == 'splaTAM':
    data = bench._CvPoses(data)
elif name == 'point-slam':
    data = bench._NumpyImages(data)
cad = cadence[name]
slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                      keyframe_every=cad.keyframe_every, pose_device=dev,
                      use_relative_pose=cad.use_relative_pose,
                      init_pose_offset=cad.init_pose_offset)
for k in range(21):
    slam.step(k)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for k in range(21, 31):
    slam.step(k)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40)
print(s.getvalue()[:6000])
