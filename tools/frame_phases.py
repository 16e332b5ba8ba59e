"""Wall time of the phases of a frame of one algorithm, each bracketed by
device syncs (so a phase's number is its own host + device time, not the
drain of what was queued before it).  The syncs remove the overlap between
host and device that the real loop has: the sum is an upper bound of a frame.

    python tools/frame_phases.py splaTAM|point-slam|vox-fusion [warm] [frames]
"""
import collections
import gc
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from xrdslam_amd.data.synthetic import SyntheticRoom  # noqa: E402
from xrdslam_amd.slam.algorithms import base_algorithm as ba  # noqa: E402
from xrdslam_amd.slam.common.camera import Camera  # noqa: E402
from xrdslam_amd.slam.configs import input_config as ic  # noqa: E402
from xrdslam_amd.slam.pipeline import SequentialSLAM  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'splaTAM'
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device('cuda:0')
torch.manual_seed(0)
np.random.seed(0)
random.seed(0)
cam = Camera(**bench.CAM)
make = {'point-slam': ic.pointslam_config, 'splaTAM': ic.splatam_config,
        'vox-fusion': ic.voxfusion_config}[name]
wrap = {'point-slam': bench._NumpyImages, 'splaTAM': bench._CvPoses,
        'vox-fusion': bench._CvPoses}[name]
algo = make().setup(camera=cam, device=str(dev))
algo.use_graphs = True
data = wrap(SyntheticRoom(bench.CO_BOUND, H=cam.height, W=cam.width,
                          fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy,
                          n_frames=200, device=dev))
cad = ic.cadence[name]
slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                      keyframe_every=cad.keyframe_every,
                      lazy_start=cad.lazy_start, pose_device=str(dev),
                      use_relative_pose=cad.use_relative_pose,
                      init_pose_offset=cad.init_pose_offset)
T, N = collections.defaultdict(float), collections.Counter()


def timed(obj, attr, label=None):
    f = getattr(obj, attr)
    label = label or attr

    def w(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            return f(*a, **k)
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize()
        T[label] += time.perf_counter() - t
        N[label] += 1
        return r
    setattr(obj, attr, w)


for k in range(warm):
    slam.step(k)
for attr in ('do_tracking', 'do_mapping', 'select_optimize_frames',
             'pre_precessing', 'post_processing', '_iteration'):
    timed(algo, attr)
if hasattr(algo.model, 'model_update'):
    timed(algo.model, 'model_update', 'model.model_update')
replay0 = torch.cuda.CUDAGraph.replay


def replay(self):
    torch.cuda.synchronize()
    t = time.perf_counter()
    replay0(self)
    torch.cuda.synchronize()
    T['graph replay'] += time.perf_counter() - t
    N['graph replay'] += 1


torch.cuda.CUDAGraph.replay = replay
capture0 = ba._capture


class Capture(capture0):
    def __enter__(self):
        torch.cuda.synchronize()
        self._t = time.perf_counter()
        return super().__enter__()

    def __exit__(self, *a):
        r = super().__exit__(*a)
        torch.cuda.synchronize()
        T['capture (with its iteration)'] += time.perf_counter() - self._t
        N['capture (with its iteration)'] += 1
        return r


ba._capture = Capture
gc.collect()
gc.disable()       # like bench.py's timed region
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(warm, warm + n):
    slam.step(k)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f'{name}: {n} frames, {tot / n * 1e3:.1f} ms a frame with the syncs of '
      'this tool (phases nest: do_mapping contains the rows below it)')
for k, v in sorted(T.items(), key=lambda x: -x[1]):
    print(f'{k:32s} {v / n * 1e3:8.2f} ms/frame {N[k] / n:7.1f} calls/frame '
          f'{v / N[k] * 1e3:9.3f} ms/call')
