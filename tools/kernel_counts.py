"""device activities of ONE optimisation iteration per algorithm and stage
(torch profiler, grouped by kernel name), with the share of launches / device
time in this repo's own kernels (xrd::) vs torch's.  The iterations are taken
from a normal run with the captured iterations' fixed shapes, eagerly.

    python tools/kernel_counts.py [algo ...]     (default: all five)"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from xrdslam_amd.data.synthetic import SyntheticRoom  # noqa: E402
from xrdslam_amd.slam.common.camera import Camera  # noqa: E402
from xrdslam_amd.slam.configs import input_config as ic  # noqa: E402
from xrdslam_amd.slam.pipeline import SequentialSLAM  # noqa: E402

dev = 'cuda:0'
# (config factory, bound, frames to run, wrap dataset, {tag: (is_mapping,
# step)} taken from the LAST frame that reaches them)
SPEC = {
    'nice-slam': (lambda: ic.nice_slam_config(bench.BOUND), bench.BOUND, 11,
                  None, {'track': (False, 3), 'map middle': (True, 3),
                         'map fine': (True, 33), 'map color': (True, 53),
                         'map coarse': (True, 3, True)}),
    'co-slam': (lambda: ic.coslam_config(bench.CO_BOUND), bench.CO_BOUND, 11,
                None, {'track': (False, 3), 'map': (True, 3)}),
    'vox-fusion': (ic.voxfusion_config, bench.CO_BOUND, 6, bench._CvPoses,
                   {'track': (False, 3), 'map': (True, 3)}),
    'splaTAM': (ic.splatam_config, bench.CO_BOUND, 4, bench._CvPoses,
                {'track': (False, 3), 'map': (True, 3)}),
    'point-slam': (ic.pointslam_config, bench.CO_BOUND, 3, bench._NumpyImages,
                   {'track': (False, 3), 'map geometry': (True, 3),
                    'map color': (True, 200)}),
}


def report(tag, p):
    c, t = collections.Counter(), collections.Counter()
    for e in p.events():
        if e.device_type == torch.autograd.DeviceType.CUDA and \
                '#' not in e.name:          # Optimizer.step#... ranges
            c[e.name[:64]] += 1
            t[e.name[:64]] += e.device_time if hasattr(e, 'device_time') \
                else e.cuda_time
    own_n = sum(k for n, k in c.items() if 'xrd::' in n)
    own_t = sum(v for n, v in t.items() if 'xrd::' in n)
    tot_n, tot_t = sum(c.values()), sum(t.values())
    print(f'== {tag}: {tot_n} device activities, {tot_t:.0f} us; own kernels '
          f'{own_n} launches ({100.0 * own_n / max(tot_n, 1):.0f} %), '
          f'{100.0 * own_t / max(tot_t, 1e-9):.1f} % of device time')
    for n, k in sorted(c.items(), key=lambda x: -t[x[0]])[:40]:
        print(f'   {k:4d} x {n}  ({t[n]:.0f} us)')
    if os.environ.get('XRD_KC_STACK'):
        # where the small torch launches come from (python frames)
        src = collections.Counter()
        for e in p.events():
            if e.device_type != torch.autograd.DeviceType.CUDA and \
                    e.name in ('aten::fill_', 'aten::zero_', 'aten::copy_',
                               'aten::cat', 'aten::add', 'aten::mul',
                               'aten::where', 'aten::sum', 'aten::index',
                               'aten::sort', 'aten::clone', 'aten::sub',
                               'aten::div', 'aten::abs', 'aten::lt',
                               'aten::gt', 'aten::bitwise_and',
                               'aten::randint', 'aten::rand',
                               'aten::normal_', 'aten::_to_copy',
                               'aten::masked_fill_', 'aten::index_put_'):
                fr = [f for f in (e.stack or []) if 'xrdslam_amd' in f or
                      'bench.py' in f]
                where = fr[0].split('xrdslam_amd/')[-1] if fr else \
                    str(getattr(e, 'input_shapes', ''))[:70]
                src[(e.name, where)] += 1
        for (op, where), k in src.most_common(60):
            print(f'      {k:3d} {op:22s} {where}')


def run(name):
    make, bound, n_frames, wrap, targets = SPEC[name]
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = make()
    if hasattr(cfg, 'mapping_first_n_iters'):
        cfg.mapping_first_n_iters = min(cfg.mapping_first_n_iters, 300)
    cam = Camera(**bench.CAM)
    algo = cfg.setup(camera=cam, device=dev)
    algo.use_graphs = False
    algo.eager_fixed_shapes = True
    data = SyntheticRoom(bound, H=cam.height, W=cam.width, fx=cam.fx,
                         fy=cam.fy, cx=cam.cx, cy=cam.cy, n_frames=200,
                         device=dev)
    if wrap is not None:
        data = wrap(data)
    cad = ic.cadence[name]
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          lazy_start=cad.lazy_start, pose_device=dev,
                          use_relative_pose=cad.use_relative_pose,
                          init_pose_offset=cad.init_pose_offset)
    state = {'on': False, 'seen': {}}
    orig = algo._iteration

    def wrapped(optimizers, frames, is_mapping, step, *a, **kw):
        for tag, key in targets.items():
            coarse = bool(kw['coarse']) if 'coarse' in kw else \
                bool(a[1]) if len(a) > 1 else False
            if state['on'] and key[:2] == (is_mapping, step) and \
                    coarse == (len(key) > 2 and key[2]) and \
                    tag not in state['seen']:
                torch.cuda.synchronize()
                with profile(activities=[ProfilerActivity.CUDA,
                                         ProfilerActivity.CPU],
                             record_shapes=bool(os.environ.get(
                                 'XRD_KC_STACK')),
                             with_stack=bool(os.environ.get(
                                 'XRD_KC_STACK'))) as p:
                    out = orig(optimizers, frames, is_mapping, step, *a, **kw)
                    torch.cuda.synchronize()
                state['seen'][tag] = p
                return out
        return orig(optimizers, frames, is_mapping, step, *a, **kw)
    algo._iteration = wrapped
    for k in range(n_frames):
        state['on'] = k == n_frames - 1
        slam.step(k)
    print(f'#### {name} (frame {n_frames - 1})')
    for tag in targets:
        if tag in state['seen']:
            report(f'{name} {tag}', state['seen'][tag])
        else:
            print(f'== {name} {tag}: not reached')


if __name__ == '__main__':
    for name in (sys.argv[1:] or list(SPEC)):
        run(name)
