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
for kind in ('coarse', 'middle', 'fine', 'color'):
    f = torch.cat([torch.randn(int(np.prod(s))) * (25. if n == 'embedder._B' else 0.2) for n, s in en.param_shapes(kind)]).to(dev)
    scene.set_decoder(kind, f.requires_grad_(kind == 'color'))
n = 1000
o = (torch.rand(n, 3, device=dev) - 0.5) * 2.0
d = torch.randn(n, 3, device=dev); d = d / d.norm(dim=1, keepdim=True)
depth = 1.0 + 2.0 * torch.rand(n, 1, device=dev)
en.PROFILE = {}
for pose in (False, True):
    oo, dd = o.clone().requires_grad_(pose), d.clone().requires_grad_(pose)
    for _ in range(12):
        dep, var, rgb = en.nice_render(scene, 'color', oo, dd, depth)
        (dep.sum() + rgb.sum()).backward()
torch.cuda.synchronize()
for k, ev in en.PROFILE.items():
    ms = [a.elapsed_time(b) for a, b in ev][2:]
    print(os.environ.get('TAG', ''), k, f'{1e3*sum(ms)/len(ms):.1f} us')
