"""host-side profile of the NICE-SLAM loop (where do the milliseconds go?)"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xrdslam_amd.data.synthetic import SyntheticRoom
from xrdslam_amd.slam.common.camera import Camera
from xrdslam_amd.slam.configs.input_config import nice_slam_config
from xrdslam_amd.slam.pipeline import SequentialSLAM
BOUND = [[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]]
dev = 'cuda:0'
torch.manual_seed(0)
cfg = nice_slam_config(BOUND); cfg.mapping_first_n_iters = 60
cam = Camera(320., 320., 319.5, 239.5, 640, 480)
algo = cfg.setup(camera=cam, device=dev)
data = SyntheticRoom(BOUND, n_frames=200, device=dev)
slam = SequentialSLAM(algo, data, pose_device=dev)
for k in range(3): slam.step(k)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for k in range(3, 8): slam.step(k)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:7000])
