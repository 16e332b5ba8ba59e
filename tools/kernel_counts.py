"""kernel launches of ONE eager NICE-SLAM iteration per stage and of one
tracking iteration (torch profiler, grouped by kernel name)"""
import os
import sys
import collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

from xrdslam_amd.data.synthetic import SyntheticRoom
from xrdslam_amd.slam.common.camera import Camera
from xrdslam_amd.slam.common.frame import Frame
from xrdslam_amd.slam.configs.input_config import nice_slam_config
from xrdslam_amd.slam.pipeline import SequentialSLAM

BOUND = [[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]]
dev = 'cuda:0'
torch.manual_seed(0)
cfg = nice_slam_config(BOUND)
cfg.mapping_first_n_iters = 60
cam = Camera(320., 320., 319.5, 239.5, 640, 480)
algo = cfg.setup(camera=cam, device=dev)
algo.use_graphs = False
data = SyntheticRoom(BOUND, n_frames=200, device=dev)
slam = SequentialSLAM(algo, data, pose_device=dev)
for k in range(6):
    slam.step(k)
frame = slam.step(6)
algo.fixed_shape_batches = True


def count(fn, tag):
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as p:
        fn()
        torch.cuda.synchronize()
    c = collections.Counter()
    t = collections.Counter()
    for e in p.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            c[e.name[:70]] += 1
            t[e.name[:70]] += e.device_time if hasattr(e, 'device_time') \
                else e.cuda_time
    print(f'== {tag}: {sum(c.values())} device activities, '
          f'{sum(t.values()):.0f} us')
    for n, k in c.most_common(30):
        print(f'   {k:4d} x {n}  ({t[n]:.0f} us)')


frames = algo.select_optimize_frames(frame, 'overlap')
for stage_step, tag in ((0, 'map middle'), (30, 'map fine'), (50, 'map color')):
    opt = algo.setup_optimizers(60, frames, is_mapping=True)
    algo.pre_precessing(frame, True)
    count(lambda: algo._iteration(opt, frames, True, stage_step, 60, False,
                                  None), tag)
opt = algo.setup_optimizers(60, frames, is_mapping=True, coarse=True)
count(lambda: algo._iteration(opt, frames, True, 0, 60, True, None),
      'map coarse')
f = Frame(fid=7, rgb=data[7]['rgb'], depth=data[7]['depth'],
          gt_pose=data[7]['c2w'].astype(np.float32),
          init_pose=data[6]['c2w'].astype(np.float32),
          separate_LR=algo.is_separate_LR(), rot_rep=algo.get_rot_rep(),
          device=dev)
algo.pre_precessing(f, False)
opt = algo.setup_optimizers(10, [f], is_mapping=False)
track = {'loss': torch.full((), 1e10, dtype=torch.float64, device=dev),
         'c2w': torch.zeros(4, 4, device=dev),
         'valid': torch.zeros((), dtype=torch.bool, device=dev)}
count(lambda: algo._iteration(opt, [f], False, 0, 10, False, track), 'track')
