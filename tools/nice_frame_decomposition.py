"""NICE-SLAM frame-time decomposition on the bench workload: the default
frame loop, the same without the coarse mapper, and tracking-only frames (no
map frame inside the timed region) - what the coarse mapper costs on its side
stream and what a tracking frame costs.
Run on the GPU box:  python tools/nice_frame_decomposition.py"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from xrdslam_amd.data.synthetic import SyntheticRoom
from xrdslam_amd.slam.common.camera import Camera
from xrdslam_amd.slam.configs.input_config import cadence, nice_slam_config
from xrdslam_amd.slam.pipeline import SequentialSLAM
dev = torch.device('cuda:0')
def run(tag, coarse=True, map_every=None, steps=100, warm=10):
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    cfg = nice_slam_config(bench.BOUND)
    cfg.coarse = coarse
    pre = os.path.join(ROOT, 'xrdslam_amd', 'data', 'pretrained', 'nice_decoders_synth.pt')
    cfg.model.pretrained_decoders_xrd = pre
    cam = Camera(**bench.CAM)
    algo = cfg.setup(camera=cam, device=str(dev)); algo.use_graphs = True
    n = warm + steps + 1
    data = SyntheticRoom(bench.BOUND, H=cam.height, W=cam.width, fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy, n_frames=max(n, bench.NICE_TRAJ_FRAMES), device=dev)
    data.preload(range(n))
    cad = cadence['nice-slam']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every, keyframe_every=cad.keyframe_every, pose_device=str(dev))
    for k in range(1 + warm): slam.step(k)
    if map_every is not None: slam.map_every = map_every
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(1 + warm, n - 1): slam.step(k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{tag}: {(steps - 1) / dt:.1f} frames/s, {dt / (steps - 1) * 1e3:.3f} ms a frame', flush=True)
run('default')
run('no coarse mapper', coarse=False)
run('tracking only (no map frames in the timed region)', map_every=10**9)
run('default again')
