"""Co-SLAM frames/s on the synthetic office0-shaped sequence (development
measurement; the contract line is bench.py)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--first-iters', type=int, default=200)
    ap.add_argument('--graphs', action='store_true')
    ap.add_argument('--algo', default='co-slam')
    args = ap.parse_args()
    from bench import BOUND, CAM
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import (algorithm_configs,
                                                       cadence)
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = algorithm_configs[args.algo]() if args.algo == 'vox-fusion' \
        else algorithm_configs[args.algo](BOUND)
    if args.algo != 'vox-fusion':
        cfg.mapping_first_n_iters = args.first_iters
    cam = Camera(**CAM)
    algo = cfg.setup(camera=cam, device=str(dev))
    algo.use_graphs = args.graphs
    data = SyntheticRoom(BOUND, H=cam.height, W=cam.width, fx=cam.fx,
                         fy=cam.fy, cx=cam.cx, cy=cam.cy, n_frames=200,
                         device=dev)
    cad = cadence[args.algo]
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device=str(dev),
                          use_relative_pose=cad.use_relative_pose,
                          init_pose_offset=cad.init_pose_offset)
    for k in range(1 + args.warmup):
        slam.step(k)
    slam.t_track = slam.t_map = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1 + args.warmup, 1 + args.warmup + args.steps):
        slam.step(k, sync=torch.cuda.synchronize)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({'algo': args.algo, 'fps': args.steps / el,
                      'ms_per_frame': el / args.steps * 1e3,
                      'track_ms': slam.t_track / args.steps * 1e3,
                      'map_ms': slam.t_map / args.steps * 1e3,
                      'ate_rmse_m': slam.ate_rmse()}))


if __name__ == '__main__':
    main()
