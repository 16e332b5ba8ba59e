"""Helpers shared by the NICE-SLAM oracle and HIP parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_nice_golden():
    d = np.load(os.path.join(GOLDEN, 'nice_render.npz'))
    g = {k: d[k] for k in d.files}
    bound = torch.from_numpy(g['bound'])
    grids = {k: torch.from_numpy(g[k]) for k in
             ('grid_coarse', 'grid_middle', 'grid_fine', 'grid_color')}
    decs = {}
    for name in ('coarse', 'middle', 'fine', 'color'):
        pre = f'dec_{name}/'
        decs[name] = {k[len(pre):]: torch.from_numpy(v)
                      for k, v in g.items() if k.startswith(pre)}
    fx, fy, cx, cy, W, H = g['cam']
    return g, bound, grids, decs, (fx, fy, cx, cy, int(W), int(H))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
