"""Point-SLAM colour path: per-kernel timing of the fused forward / backward
(+ weight products) at the mapping and tracking batch sizes, HIP events around
every call.  `python tools/pc_timing.py [lib.so] [n ...]` (an alternative
library: a variant built by tools/_exp/build_var.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
args = sys.argv[1:]
if args and args[0].endswith('.so'):
    from xrdslam_amd import _lib
    _lib.LIB_PATH = os.path.abspath(args.pop(0))
import torch  # noqa: E402

import test_pointslam_hip as T  # noqa: E402
from xrdslam_amd.engine import point as ep  # noqa: E402

dev = 'cuda:0'
sizes = [int(a) for a in args] or [24508, 7500]
for n in sizes:
    dec, npc, q, radius, w_out = T._color_case(dev, N=19389, n=n)
    for mapping in (True, 'weights only', 'features only', False):
        dec.map_gradients = bool(mapping)
        p = q.clone().to(dev).requires_grad_(True)
        nb = npc.find_neighbors_faiss(p.detach(), dynamic_radius=radius)
        real_feats, real_params = npc.col_feats, ep.color_params

        def fwd():
            if mapping == 'weights only':
                npc.col_feats = real_feats.detach()
            if mapping == 'features only':
                ep.color_params = lambda d: [t.detach() for t in real_params(d)]
            try:
                return ep.color(dec, p, nb, npc, radius)
            finally:
                npc.col_feats, ep.color_params = real_feats, real_params
        rgb = fwd()
        g = torch.ones_like(rgb)
        rgb.backward(g)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        reps = 10
        for _ in range(reps):
            e[0].record()
            rgb = fwd()
            e[1].record()
            rgb.backward(g)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1])
            tb += e[1].elapsed_time(e[2])
        label = {True: 'mapping (features + weights)', False:
                 'tracking (pose only)'}.get(mapping, mapping)
        print(f'n={n} {label}'
              f': fwd {tf / reps * 1e3:.0f} us  bwd group {tb / reps * 1e3:.0f}'
              f' us (events around the autograd calls, glue included)',
              flush=True)
