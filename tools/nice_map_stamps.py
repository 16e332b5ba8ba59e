"""Where a block of the NICE-SLAM one-launch mapping iteration spends its time.

Builds a STAMPED COPY of csrc/nice_map.hip (wall_clock64() at the phase borders
of nice_map_fused_kernel, written by every wave of the first blocks into a
__device__ array; the copy is compiled into tools/scratch/, never into the
product library), runs the colour-stage launch with decoder gradients at the
office0 configuration and prints the phases of some waves in microseconds.

    python tools/nice_map_stamps.py build        # build container (hipcc)
    python tools/nice_map_stamps.py run [n_rays [cameras]]   # GPU box
"""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tools', 'scratch')
LIB = os.path.join(OUT, 'libxrdslam_hip_stamps.so')
NS = 40          # stamp slots a wave
NB = 8           # stamped blocks

# (label of the phase that ENDS at the stamp, anchor text, occurrence, where)
ANCHORS = [
    ('start', '    const bool active = ray < n;\n', 0, 'after'),
    ('ray+z+gatherM', '    stage_weights(wl, sc.dec[1], PM::WHT);\n', 0, 'before'),
    ('stageM', '    stage_weights(wl, sc.dec[1], PM::WHT);\n', 0, 'after'),
    ('fwdM', '    if (STAGE >= XRD_STAGE_FINE) {\n      f32x4 c_f[1][4];\n      if (active) {\n        f32x4 cf[2];', 0, 'before'),
    ('gatherF', '      stage_weights(wl, sc.dec[2], PF::WHT);\n', 0, 'before'),
    ('stageF', '      stage_weights(wl, sc.dec[2], PF::WHT);\n', 0, 'after'),
    ('fwdF', '    if (STAGE == XRD_STAGE_COLOR) {\n      if (active) {\n        tri_prepare_n<NEED_DP>(xn, sc.bound, sc.gdim + 9, tr);\n        tri_gather', 0, 'before'),
    ('gatherC', '      stage_weights(wl, sc.dec[3], PC::WHT);\n', 0, 'before'),
    ('stageC', '      stage_weights(wl, sc.dec[3], PC::WHT);\n', 0, 'after'),
    ('fwdC', '    if (active) {\n      if (!tg.inb) occ = 100.f;', 0, 'before'),
    ('raw+barrier', '    // ---- compositing, loss, compositing backward', 0, 'before'),
    ('composite', '    // ---- backward: colour -> fine -> middle', 0, 'before'),
    ('stageCb', '      if (NEED_DW) {\n        if (active)\n          color_bwd_emit', 0, 'before'),
    ('bwdC', '      if (active) {\n        tri_prepare_n<NEED_DP>(xn, sc.bound, sc.gdim + 9, tr);\n        if (NEED_DP) tri_backward_dp(sc.grid[3]', 0, 'before'),
    ('scatterC', '    if (STAGE >= XRD_STAGE_FINE) {\n      f32x4 c_f[1][4], gc[1][4];', 0, 'before'),
    ('stageFb', '      asm volatile("" : "+v"(lane));\n      if (active) {\n        mlp_bwd_ra<64, 1, NEED_DP, NEED_DP>', 0, 'before'),
    ('bwdF', '        tri_prepare_n<NEED_DP>(xn, sc.bound, sc.gdim + 6, tr);\n        if (NEED_DP) tri_backward_dp(sc.grid[2]', 0, 'before'),
    ('scatterF', '    {\n      const float go[1][1] = {{gocc}};\n      f32x4 gc[1][2];\n      stage_weights(wl, sc.dec[1] + PM::EMB', 0, 'before'),
    ('stageMb', '      asm volatile("" : "+v"(lane));\n      if (active) {\n        mlp_bwd_ra<32, 1, NEED_DP, NEED_DP>', 0, 'before'),
    ('bwdM', '        tri_prepare_n<NEED_DP>(xn, sc.bound, sc.gdim + 3, tr);\n        if (NEED_DP) tri_backward_dp(sc.grid[1]', 0, 'before'),
    ('scatterM', '    if (NEED_DW) {\n      // every tile of the group has left its operands', 0, 'before'),
    ('dW barrier', '      const int rays_here = n - grp * G::RPBM', 0, 'before'),
    ('dW contract', '      // (a further group\'s set-up writes the LDS the last round', 0, 'before'),
]
# inside dw_contract: stamps 30.. (per round: landed+barrier / tiles done)
RING = [
    ('    ring_landed();\n    __syncthreads();  // round r is in its buffer',
     'before', 'XRD_STAMP(24 + 3 * r);'),
    ('    if (r + 1 < kRingRounds && (r + 1) * kRingTiles < ntiles)\n      ring_issue',
     'before', 'XRD_STAMP(25 + 3 * r);'),
    ('  switch (wave) {\n    case 6: dwl_role_flush<0>', 'before',
     'XRD_STAMP(36);'),
]


def build():
    src = open(os.path.join(ROOT, 'xrdslam_amd', 'csrc', 'nice_map.hip')).read()
    pre = ('\n__device__ unsigned long long g_xrd_stamps[%d];\n'
           '#define XRD_STAMP(k) do { if ((threadIdx.x & 63) == 0 && '
           'blockIdx.x < %d) g_xrd_stamps[(blockIdx.x * 12 + (threadIdx.x >> '
           '6)) * %d + (k)] = wall_clock64(); } while (0)\n'
           % (NB * 12 * NS, NB, NS))
    src = src.replace('namespace xrd {\nnamespace {\n',
                      'namespace xrd {\nnamespace {\n' + pre, 1)
    for k, (label, anchor, occ, where) in enumerate(ANCHORS):
        pos = -1
        for _ in range(occ + 1):
            pos = src.index(anchor, pos + 1)
        at = pos if where == 'before' else pos + len(anchor)
        src = src[:at] + f'    XRD_STAMP({k});\n' + src[at:]
    for anchor, where, code in RING:
        pos = src.index(anchor)
        at = pos if where == 'before' else pos + len(anchor)
        src = src[:at] + '    ' + code + '\n' + src[at:]
    src += ('\nextern "C" int xrd_debug_stamps(unsigned long long* out) {\n'
            '  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL('
            'xrd::g_xrd_stamps), sizeof(unsigned long long) * %d);\n}\n'
            % (NB * 12 * NS))
    src = src.replace('namespace xrd {\nnamespace {\n' + pre,
                      'namespace xrd {\n' + pre + 'namespace {\n', 1)
    os.makedirs(OUT, exist_ok=True)
    tmp = os.path.join(OUT, 'nice_map_stamped.hip')
    open(tmp, 'w').write(src)
    from xrdslam_amd import build as b
    b.build(verbose=False)
    obj = os.path.join(OUT, 'nice_map_stamped.o')
    extra = []
    # what-if experiments on the stamped copy only: XRD_STAMP_EDIT =
    # 'old text=>new text' applied to a scratch copy of nice_device.h
    edit = os.environ.get('XRD_STAMP_EDIT')
    if edit:
        inc = os.path.join(OUT, 'inc')
        os.makedirs(inc, exist_ok=True)
        h = open(os.path.join(b.CSRC, 'nice_device.h')).read()
        for e in edit.split('||'):
            old, new = e.split('=>')
            assert old in h, old
            h = h.replace(old, new)
        open(os.path.join(inc, 'nice_device.h'), 'w').write(h)
        extra = ['-I' + inc]
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + extra + b.FLAGS +
                          ['-x', 'hip', '-c', tmp, '-o', obj])
    objs = [os.path.join(b.OBJ, os.path.basename(s) + '.o')
            for s in b.sources() if not s.endswith('nice_map.hip')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950',
                           '-shared', '-fPIC', '-o', LIB] + objs + [obj,
                                                                   '-ldl'])
    print('built', LIB)


def run():
    import numpy as np
    import torch
    from xrdslam_amd import _lib
    _lib.LIB_PATH = LIB
    from xrdslam_amd.engine import nice as en
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    cams = int(sys.argv[3]) if len(sys.argv) > 3 else 0
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
        scene.set_grid(k, g.requires_grad_())
    for kind in ('coarse', 'middle', 'fine', 'color'):
        flat = torch.cat([torch.randn(int(np.prod(s))) *
                          (25. if nm == 'embedder._B' else 0.2)
                          for nm, s in en.param_shapes(kind)]).to(dev)
        scene.set_decoder(kind, flat)
    lib = _lib.lib()
    lib.xrd_debug_stamps.restype = C.c_int
    lib.xrd_debug_stamps.argtypes = [C.c_void_p]
    if cams:
        cam = torch.randint(0, cams, (n, ), device=dev)
        centres = (torch.rand(cams, 3, device=dev) - 0.5) * 2.0
        d = torch.randn(n, 3, device=dev)
        d = d / d.norm(dim=1, keepdim=True)
        o = centres[cam].contiguous()
    else:
        o = ((torch.rand(n, 3, device=dev) - 0.5) * 2.0).contiguous()
        d = torch.randn(n, 3, device=dev)
        d = (d / d.norm(dim=1, keepdim=True)).contiguous()
    depth = (0.5 + torch.rand(n, 1, device=dev) * 2.0).contiguous()
    rgb = torch.rand(n, 3, device=dev)
    for dw in (True, False):
        for it in range(3):
            en.nice_map_iter(scene, 'color', o, d, depth, None, rgb, None,
                             0.2, False, dw)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (NB * 12 * NS))()
        rc = lib.xrd_debug_stamps(buf)
        assert rc == 0, rc
        st = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 12, NS).astype(
            np.int64)
        print(f'== n={n} cameras={cams} colour stage, decoder gradients={dw} '
              '(us per phase; wall_clock64 = 100 MHz)')
        labels = [a[0] for a in ANCHORS]
        for b, w in ((0, 0), (0, 7), (0, 9), (3, 0), (5, 4)):
            t = st[b, w]
            k_end = len(labels) - (0 if dw else 2)
            ph = ' '.join(f'{labels[k]} {(t[k] - t[k - 1]) / 100.0:.1f}'
                          for k in range(1, k_end))
            print(f' block {b} wave {w}: total '
                  f'{(t[k_end - 1] - t[0]) / 100.0:.1f} | {ph}')
            if dw:
                rr = ' '.join(
                    f'r{r}: wait {(t[25 + 3 * r] - t[24 + 3 * r]) / 100.0:.1f}'
                    f' compute {((t[24 + 3 * (r + 1)] if r < 3 else t[36]) - t[25 + 3 * r]) / 100.0:.1f}'
                    for r in range(4))
                print(f'    contraction rounds: {rr}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    else:
        run()
