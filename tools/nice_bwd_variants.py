"""time every backward variant of the NICE-SLAM render at the mapping size
(run under rocprofv3 --kernel-trace --stats; see tools/prof_summary.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xrdslam_amd.engine import nice as en
dev = torch.device('cuda:0'); torch.manual_seed(0)
bound = torch.tensor([[-5.5, 6.0199995], [-6.7, 5.4599998], [-4.7, 5.5399998]], dtype=torch.float64)
shapes = {'grid_coarse': (10, 12, 11), 'grid_middle': (31, 37, 35), 'grid_fine': (63, 75, 71), 'grid_color': (63, 75, 71)}
scene = en.NiceScene(bound, device=dev)
for k, s in shapes.items():
    scene.set_grid(k, en.to_channels_last_grid(torch.randn(1, 32, *s, device=dev) * 0.01).requires_grad_(True))
flats = {}
for kind in ('coarse', 'middle', 'fine', 'color'):
    flats[kind] = torch.cat([torch.randn(int(np.prod(s))) * (25. if n == 'embedder._B' else 0.2) for n, s in en.param_shapes(kind)]).to(dev)
    scene.set_decoder(kind, flats[kind])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
o = (torch.rand(n, 3, device=dev) - 0.5) * 2.0
d = torch.randn(n, 3, device=dev); d = d / d.norm(dim=1, keepdim=True)
depth = 1.0 + 2.0 * torch.rand(n, 1, device=dev)
for stage in ('middle', 'fine', 'color'):
    for pose in (False, True):
        for dec in ((False, True) if stage == 'color' else (False,)):
            for g in scene.grids.values(): g.requires_grad_(True)
            flat = flats['color'].clone().requires_grad_(dec)
            scene.set_decoder('color', flat)
            oo, dd = o.clone().requires_grad_(pose), d.clone().requires_grad_(pose)
            for _ in range(8):
                dep, var, rgb = en.nice_render(scene, stage, oo, dd, depth)
                (dep.sum() + rgb.sum()).backward()
torch.cuda.synchronize()
