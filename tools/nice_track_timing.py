"""HIP-event timing of the NICE-SLAM tracking pair at the tracking batch (200
rays x 48 samples, colour stage, office0 grids): xrd_nice_render_fwd (the
three-pass kernel), xrd_nice_render_fwd_masks (decoder-per-block launches + the
finishing launch) and xrd_nice_render_bwd_masks, each as 20 calls captured in
one hipGraph, replayed behind 10 ms of matmul (sustained clocks).  Also checks
that the two forwards agree bit for bit.
Run on the GPU box:  python tools/nice_track_timing.py [n_rays [lib.so]]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from xrdslam_amd import _lib

if len(sys.argv) > 2:      # A/B against another build of the library
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from xrdslam_amd.engine import nice as en

dev = torch.device('cuda:0')
torch.manual_seed(0)
bound = torch.tensor([[-5.5, 6.0199995], [-6.7, 5.4599998],
                      [-4.7, 5.5399998]], dtype=torch.float64)
shapes = {'grid_coarse': (10, 12, 11), 'grid_middle': (31, 37, 35),
          'grid_fine': (63, 75, 71), 'grid_color': (63, 75, 71)}
scene = en.NiceScene(bound, device=dev)
for k, s in shapes.items():
    scene.set_grid(k, en.to_channels_last_grid(
        torch.randn(1, 32, *s, device=dev) * 0.01))
for kind in ('coarse', 'middle', 'fine', 'color'):
    flat = torch.cat([torch.randn(int(np.prod(s))) *
                      (25. if nm == 'embedder._B' else 0.2)
                      for nm, s in en.param_shapes(kind)]).to(dev)
    scene.set_decoder(kind, flat)
lib, P = _lib.lib(), _lib.ptr


def st():
    return _lib.stream_ptr(dev)   # the capture's stream inside a capture


n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
o = ((torch.rand(n, 3, device=dev) - 0.5) * 2.0).contiguous()
d = torch.randn(n, 3, device=dev)
d = (d / d.norm(dim=1, keepdim=True)).contiguous()
depth = (1.0 + 2.0 * torch.rand(n, device=dev)).contiguous()
dmax = depth.max().reshape(1)
cs = scene.c_struct()
dep = torch.empty(n, dtype=torch.float64, device=dev)
var = torch.empty_like(dep)
rgb = torch.empty(n, 3, device=dev)
raw = torch.empty(n, 48, 4, device=dev)
masks = torch.empty(lib.xrd_nice_fwd_masks_words(n), dtype=torch.int64,
                    device=dev)
gdep = torch.randn(n, dtype=torch.float64, device=dev)
grgb = torch.randn(n, 3, device=dev)
g_o, g_d = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
ws = torch.empty(lib.xrd_nice_bwd_ws_floats(n), device=dev)


def f_plain():
    _lib.check(lib.xrd_nice_render_fwd(
        C.byref(cs), 3, n, P(o), P(d), P(depth), P(dmax), P(dep), P(var),
        P(rgb), P(raw), st()))


def f_masks():
    _lib.check(lib.xrd_nice_render_fwd_masks(
        C.byref(cs), 3, n, P(o), P(d), P(depth), P(dmax), P(dep), P(var),
        P(rgb), P(raw), P(masks), st()))


def b_masks():
    _lib.check(lib.xrd_nice_render_bwd_masks(
        C.byref(cs), 3, n, P(o), P(d), P(depth), P(dmax), P(raw), P(gdep),
        None, P(grgb), P(masks), P(g_o), P(g_d), P(ws), st()))


f_plain()
torch.cuda.synchronize()
ref = [t.clone() for t in (dep, var, rgb, raw)]
for t in (dep, var, rgb, raw):
    t.zero_()
f_masks()
torch.cuda.synchronize()
print(f'{n} rays: decoder-per-block forward == three-pass forward (depth, '
      'var, rgb, raw):',
      [bool(torch.equal(a, b)) for a, b in zip(ref, (dep, var, rgb, raw))],
      'largest |difference|:',
      [float((a.double() - b.double()).abs().max())
       for a, b in zip(ref, (dep, var, rgb, raw))])
A = torch.randn(8192, 8192, device=dev)
for name, f in (('xrd_nice_render_fwd (three passes a wave)', f_plain),
                ('xrd_nice_render_fwd_masks (roles + finish)', f_masks),
                ('xrd_nice_render_bwd_masks (roles + finish)', b_masks)):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    ts = []
    for _ in range(5):
        for _ in range(3):
            A @ A
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f'{name}: {min(ts):.1f} / {sorted(ts)[2]:.1f} us a call '
          '(min / median of 5 replays of 20 captured calls)')
