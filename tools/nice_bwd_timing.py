"""HIP-event timing of xrd_nice_render_fwd / xrd_nice_render_bwd (the launch
group of one call) per stage and gradient set at the office0 config.
Run on the GPU box:  python tools/nice_bwd_timing.py [--cameras K] [n_rays ...]
(--cameras K: the rays leave K camera centres through random pixels, like a
mapping window of K frames — their first samples share grid cells, which is
what the gradient scatters see in a frame loop; default: rays scattered over
the room)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xrdslam_amd import _lib
from xrdslam_amd.engine import nice as en

dev = torch.device('cuda:0')
torch.manual_seed(0)
bound = torch.tensor([[-5.5, 6.0199995], [-6.7, 5.4599998],
                      [-4.7, 5.5399998]], dtype=torch.float64)
shapes = {'grid_coarse': (10, 12, 11), 'grid_middle': (31, 37, 35),
          'grid_fine': (63, 75, 71), 'grid_color': (63, 75, 71)}
scene = en.NiceScene(bound, device=dev)
grads = {}
for k, s in shapes.items():
    g = en.to_channels_last_grid(torch.randn(1, 32, *s, device=dev) * 0.01)
    scene.set_grid(k, g)
    grads[k] = torch.zeros_like(g, memory_format=torch.preserve_format)
for kind in ('coarse', 'middle', 'fine', 'color'):
    flat = torch.cat([torch.randn(int(np.prod(s))) *
                      (25. if n == 'embedder._B' else 0.2)
                      for n, s in en.param_shapes(kind)]).to(dev)
    scene.set_decoder(kind, flat)
lib = _lib.lib()
st = _lib.stream_ptr(dev)
P = _lib.ptr


CAMERAS = 0
if '--cameras' in sys.argv:
    k = sys.argv.index('--cameras')
    CAMERAS = int(sys.argv[k + 1])
    del sys.argv[k:k + 2]


def make_rays(n):
    if not CAMERAS:
        o = ((torch.rand(n, 3, device=dev) - 0.5) * 2.0).contiguous()
        d = torch.randn(n, 3, device=dev)
        return o, (d / d.norm(dim=1, keepdim=True)).contiguous()
    cam = torch.randint(0, CAMERAS, (n, ), device=dev)
    centres = (torch.rand(CAMERAS, 3, device=dev) - 0.5) * 2.0
    # 640x480, 90 degrees across: pixel directions around a per-camera axis
    u = (torch.rand(n, device=dev) - 0.5) * 2.0
    v = (torch.rand(n, device=dev) - 0.5) * 1.5
    axis = torch.randn(CAMERAS, 3, device=dev)
    axis = axis / axis.norm(dim=1, keepdim=True)
    up = torch.tensor([0., 0., 1.], device=dev).expand(CAMERAS, 3)
    right = torch.linalg.cross(axis, up)
    right = right / right.norm(dim=1, keepdim=True)
    up2 = torch.linalg.cross(right, axis)
    d = axis[cam] + u[:, None] * right[cam] + v[:, None] * up2[cam]
    return centres[cam].contiguous(), \
        (d / d.norm(dim=1, keepdim=True)).contiguous()


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for n in [int(a) for a in sys.argv[1:]] or [200, 1000]:
    if True:
        # coarse stage: old chain (fwd + bwd incl. replica reduce) vs one launch
        cs = scene.c_struct()
        o, d = make_rays(n)
        depth = (1.0 + 2.0 * torch.rand(n, device=dev)).contiguous()
        dep = torch.empty(n, dtype=torch.float64, device=dev)
        var = torch.empty_like(dep)
        rgb = torch.empty(n, 3, device=dev)
        raw = torch.empty(n, 32, 4, device=dev)
        gdv = torch.ones(n, dtype=torch.float64, device=dev)
        wsc = torch.zeros(lib.xrd_nice_coarse_ws_floats(C.byref(cs)),
                          device=dev)
        gg = (C.c_void_p * 4)()
        gg[0] = grads['grid_coarse'].data_ptr()
        gdec = (C.c_void_p * 4)()

        def cf():
            _lib.check(lib.xrd_nice_render_fwd(
                C.byref(cs), 0, n, P(o), P(d), None, None, P(dep), P(var),
                P(rgb), P(raw), st))

        def cb():
            _lib.check(lib.xrd_nice_render_bwd(
                C.byref(cs), 0, n, P(o), P(d), None, None, P(raw), P(gdv),
                None, None, None, None, C.byref(gg), C.byref(gdec), P(wsc),
                st))
        wsm = torch.zeros(lib.xrd_nice_map_ws_floats(C.byref(cs), 0, n),
                          device=dev)
        loss = torch.empty((), dtype=torch.float64, device=dev)
        keep = torch.ones(n, dtype=torch.uint8, device=dev)

        def cm():
            _lib.check(lib.xrd_nice_map_iter(
                C.byref(cs), 0, n, P(o), P(d), P(depth), None, None, P(keep),
                0.2, None, None, C.byref(gg), None, P(wsm), P(loss), st))
        print(f'n={n:5d} coarse fwd {timeit(cf):7.1f} us | bwd+reduce '
              f'{timeit(cb):7.1f} | map_iter (2 launches) {timeit(cm):7.1f}',
              flush=True)
    o, d = make_rays(n)
    depth = (1.0 + 2.0 * torch.rand(n, device=dev)).contiguous()
    dmax = depth.max().reshape(1)
    for stage, si in (('middle', 1), ('fine', 2), ('color', 3)):
        S = 48
        dep = torch.empty(n, dtype=torch.float64, device=dev)
        var = torch.empty_like(dep)
        rgb = torch.empty(n, 3, device=dev)
        raw = torch.empty(n, S, 4, device=dev)
        cs = scene.c_struct()

        def fwd():
            _lib.check(lib.xrd_nice_render_fwd(
                C.byref(cs), si, n, P(o), P(d), P(depth), P(dmax), P(dep),
                P(var), P(rgb), P(raw), st))
        t_f = timeit(fwd)
        gd = torch.ones(n, dtype=torch.float64, device=dev)
        gr = torch.full((n, 3), 0.2, device=dev)
        ws = torch.empty(lib.xrd_nice_bwd_ws_floats(n), device=dev)
        g_o = torch.empty(n, 3, device=dev)
        g_d = torch.empty(n, 3, device=dev)
        g_flat = torch.empty(lib.xrd_nice_flat_len(3), device=dev)
        msg = f'n={n:5d} {stage:6s} fwd {t_f:7.1f} us |'
        for grid in (1, 0):
            for dp in (0, 1):
                for dw in ((0, 1) if stage == 'color' else (0, )):
                    gg = (C.c_void_p * 4)()
                    if grid:
                        for gi, k in enumerate(('grid_coarse', 'grid_middle',
                                                'grid_fine', 'grid_color')):
                            if 1 <= gi <= si:
                                gg[gi] = grads[k].data_ptr()
                    gdec = (C.c_void_p * 4)()
                    if dw:
                        gdec[3] = g_flat.data_ptr()

                    def bwd():
                        _lib.check(lib.xrd_nice_render_bwd(
                            C.byref(cs), si, n, P(o), P(d), P(depth), P(dmax),
                            P(raw), P(gd), None, P(gr),
                            P(g_o) if dp else None, P(g_d) if dp else None,
                            C.byref(gg), C.byref(gdec), P(ws), st))
                    msg += f' grid{grid}dp{dp}dw{dw} {timeit(bwd):7.1f}'
        print(msg, flush=True)
        # the one-launch mapping iteration (forward + loss + backward)
        tcol = torch.rand(n, 3, device=dev)
        keep = torch.ones(n, dtype=torch.uint8, device=dev)
        wsm = torch.zeros(lib.xrd_nice_map_ws_floats(C.byref(cs), si, n),
                          device=dev)
        loss = torch.empty((), dtype=torch.float64, device=dev)
        msg = f'n={n:5d} {stage:6s} map_iter (fwd+loss+bwd, 2 launches) |'
        for dp in (0, 1):
            for dw in ((0, 1) if stage == 'color' else (0, )):
                gg = (C.c_void_p * 4)()
                for gi, k in enumerate(('grid_coarse', 'grid_middle',
                                        'grid_fine', 'grid_color')):
                    if 1 <= gi <= si:
                        gg[gi] = grads[k].data_ptr()

                def mapit():
                    _lib.check(lib.xrd_nice_map_iter(
                        C.byref(cs), si, n, P(o), P(d), P(depth), P(dmax),
                        P(tcol), P(keep), 0.2, P(g_o) if dp else None,
                        P(g_d) if dp else None, C.byref(gg),
                        P(g_flat) if dw else None, P(wsm), P(loss), st))
                msg += f' grid1dp{dp}dw{dw} {timeit(mapit):7.1f}'
        print(msg, flush=True)
