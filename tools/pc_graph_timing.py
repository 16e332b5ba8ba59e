"""Point-SLAM colour path: HIP events directly around the C-ABI calls
(engine.point.PROFILE: no autograd glue, no profiler) at the mapping batch
size, the launches queued BEHIND a long matmul so that they run back to back
(an idle queue adds the host's launch latency to every event span):
`python tools/pc_graph_timing.py [lib.so] [n]`."""
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
n = int(args[0]) if args else 24508
dec, npc, q, radius, w_out = T._color_case(dev, N=19389, n=n)
dec.map_gradients = True
p = q.clone().to(dev).requires_grad_(True)
nb = npc.find_neighbors_faiss(p.detach(), dynamic_radius=radius)
big = torch.randn(8192, 8192, device=dev)
for it in range(13):
    if it == 3:
        ep.PROFILE = {}
    rgb = ep.color(dec, p, nb, npc, radius)
    g = torch.ones_like(rgb)
    torch.cuda.synchronize()
    big @ big                       # ~10 ms: the queue fills behind it
    rgb.backward(g)
    torch.cuda.synchronize()
for k, evs in ep.PROFILE.items():
    us = sorted(a.elapsed_time(b) * 1e3 for a, b, _ in evs)
    print(f'n={n} {k}: median {us[len(us) // 2]:.1f} us, min {us[0]:.1f} us '
          f'({len(us)} calls; HIP events around the C-ABI call)')
