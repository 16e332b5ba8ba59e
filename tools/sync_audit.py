"""Which calls of the frame loop make the host wait for the device?  Runs the
timed region of bench.py for one algorithm with
torch.cuda.set_sync_debug_mode('warn') and lists the synchronising calls by
the innermost frame inside this repository (count over the timed frames).
A frame loop that keeps its pose chain on the device should list only what
the algorithm's own host logic needs (keyframe selection, map maintenance).
Run on the GPU box:  python tools/sync_audit.py [--algo NAME] [--steps K]"""
import collections
import os
import sys
import traceback
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

hits = collections.Counter()
from xrdslam_amd.slam.pipeline import SequentialSLAM

_step = SequentialSLAM.step
state = {'first': None, 'last': None, 'on': False}


def show(message, category, filename, lineno, file=None, line=None):
    if 'synchroniz' not in str(message):
        return
    stack = [f for f in traceback.extract_stack()
             if f.filename.startswith(ROOT) and
             'sync_audit' not in f.filename]
    where = ' <- '.join(
        f'{os.path.relpath(f.filename, ROOT)}:{f.lineno} {f.name}'
        for f in reversed(stack[-3:])) or f'{filename}:{lineno}'
    hits[where] += 1


def step(self, idx, sync=None):
    """the frames of bench.py's timed region run under sync debug mode"""
    if idx == state['first'] and not state['on']:
        state['on'] = True
        warnings.showwarning = show
        warnings.simplefilter('always')
        torch.cuda.set_sync_debug_mode('warn')
    try:
        return _step(self, idx, sync)
    finally:
        if idx == state['last'] and state['on']:
            torch.cuda.set_sync_debug_mode('default')
            state['first'] = None      # later runs of the bench: not audited


SequentialSLAM.step = step
if '--steps' not in sys.argv:
    sys.argv += ['--steps', '20']
sys.argv += ['--no-cpu-baseline', '--no-side-runs', '--no-others',
             '--no-steady-state']
steps = int(sys.argv[sys.argv.index('--steps') + 1])
warm = int(sys.argv[sys.argv.index('--warmup') + 1]) \
    if '--warmup' in sys.argv else 10
state['first'], state['last'] = 1 + warm, warm + steps
try:
    bench.main()
except SystemExit:
    pass
print(f'\nsynchronising calls inside {steps} timed frames '
      f'({sum(hits.values())} in all):', file=sys.stderr)
for where, n in hits.most_common(40):
    print(f'{n:6d} ({n / steps:6.2f} a frame)  {where}', file=sys.stderr)
