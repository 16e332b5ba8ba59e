"""Which line of the host code launches what: one frame of an algorithm run
EAGERLY (no graph replay) under torch.profiler with stacks, every device
kernel listed with the innermost frame of this repo that caused it.  Used to
find the small torch launches (fills, concatenations, index gathers) that are
left in an iteration.

    python tools/launch_trace.py --algo coslam [--frames 12] [--frame-kind map]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--algo', default='coslam',
                    choices=['coslam', 'nice', 'voxfusion', 'pointslam',
                             'splatam'])
    ap.add_argument('--frames', type=int, default=11,
                    help='frames run (with graphs) before the traced one')
    a = ap.parse_args()
    import c1_util
    from torch.profiler import ProfilerActivity, profile
    est, gt, sec, slam = c1_util.run_engine(a.algo, 0, n_frames=a.frames)
    algo = slam.algorithm
    algo.use_graphs = False
    k = a.frames
    while not slam.is_mapframe(k):
        slam.step(k)
        k += 1
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA],
                 with_stack=True) as prof:
        slam.step(k)
        torch.cuda.synchronize()
    # kernel events <- launching CPU op through the correlation id
    evs = prof.events()
    by_line = collections.defaultdict(lambda: [0, 0.0, set()])
    for e in evs:
        if e.device_type != torch.autograd.DeviceType.CPU:
            continue
        ks = [x for x in e.kernels]
        if not ks:
            continue
        # only leaf ops (an op that launched the kernel itself)
        where = '?'
        for fr in e.stack:
            if 'xrdslam_amd' in fr and 'site-packages' not in fr:
                where = fr.split('xrdslam_amd/')[-1]
                break
        for kk in ks:
            key = (where, kk.name[:60])
            by_line[key][0] += 1
            by_line[key][1] += kk.duration
            by_line[key][2].add(e.name)
    rows = sorted(by_line.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print(f'# {a.algo}: frame {k} eager, {sum(v[0] for _, v in rows)} '
          f'launches, {tot:.0f} us of kernels')
    for (where, name), (n, us, ops) in rows:
        print(f'{n:5d} {us:9.1f} us  {name:60s} {where}  '
              f'[{",".join(sorted(ops))[:60]}]')


if __name__ == '__main__':
    main()
