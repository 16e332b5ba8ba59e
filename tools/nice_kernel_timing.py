"""Kernel-level timing of the NICE-SLAM fused render at the office0 config
(grids [1,32,63,75,71] etc.).  Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xrdslam_amd.engine import nice as en

dev = torch.device('cuda:0')
torch.manual_seed(0)
bound = torch.tensor([[-5.5, 6.0199995], [-6.7, 5.4599998], [-4.7, 5.5399998]], dtype=torch.float64)
shapes = {'grid_coarse': (10, 12, 11), 'grid_middle': (31, 37, 35), 'grid_fine': (63, 75, 71), 'grid_color': (63, 75, 71)}
scene = en.NiceScene(bound, device=dev)
for k, s in shapes.items():
    g = torch.randn(1, 32, *s, device=dev) * 0.01
    scene.set_grid(k, en.to_channels_last_grid(g).requires_grad_(True))
for kind in ('coarse', 'middle', 'fine', 'color'):
    flat = torch.cat([torch.randn(int(np.prod(s))) * (25. if n == 'embedder._B' else 0.2) for n, s in en.param_shapes(kind)]).to(dev)
    if kind == 'color':
        flat.requires_grad_(True)
    scene.set_decoder(kind, flat)

def rays(n):
    o = (torch.rand(n, 3, device=dev) - 0.5) * 2.0
    d = torch.randn(n, 3, device=dev); d = d / d.norm(dim=1, keepdim=True)
    depth = 1.0 + 2.0 * torch.rand(n, 1, device=dev)
    return o, d, depth

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us

NS = [int(a) for a in sys.argv[1:]] or [200, 1000, 5000, 100000]
for n in NS:
    o, d, depth = rays(n)
    for stage in ('coarse', 'middle', 'fine', 'color'):
        with torch.no_grad():
            t_f = timeit(lambda: en.nice_render(scene, stage, o, d, depth))
        msg = f'n={n:6d} {stage:7s} fwd {t_f:9.1f} us'
        if n <= 5000:
            for need_pose in (False, True):
                oo = o.clone().requires_grad_(need_pose)
                dd = d.clone().requires_grad_(need_pose)
                def step():
                    dep, var, rgb = en.nice_render(scene, stage, oo, dd, depth)
                    (dep.sum() + rgb.sum()).backward()
                t = timeit(step, iters=10)
                msg += f' | fwd+bwd(pose={int(need_pose)}) {t:9.1f} us'
        print(msg, flush=True)
