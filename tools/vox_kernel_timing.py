"""Event-timed launches of the Vox-Fusion point kernels (xrd_vox_points_fwd /
_bwd / xrd_vox_dw, csrc/vox_render.hip, csrc/vox_dw.hip) at the two point
counts of a frame: ~36 k (a tracking iteration's 1024 rays) and ~73 k (a
mapping iteration's 2048), in the two roles (tracking: pose gradient only;
mapping: embedding + decoder gradients).  Points come in runs that share a
voxel, like consecutive samples of a ray.

    python tools/vox_kernel_timing.py [--reps 30]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))


def case(P, seed, dev, n_vox=6000, n_emb=9000, run=9):
    from xrdslam_amd.slam.model_components.decoder_voxfusion import Decoder
    g = torch.Generator().manual_seed(seed)
    vs = 0.2
    centres = (torch.randint(40, 90, (n_vox, 3), generator=g).float() +
               0.5) * vs
    vertex_idx = torch.randint(0, n_emb, (n_vox, 8), generator=g).int()
    emb = torch.randn(n_emb, 16, generator=g) * 0.3
    vox = torch.randint(0, n_vox, ((P + run - 1) // run, ), generator=g) \
        .repeat_interleave(run)[:P].int()
    xyz = centres[vox.long()] + (torch.rand(P, 3, generator=g) - 0.5) * vs
    torch.manual_seed(seed)
    dec = Decoder(depth=2, width=128, in_dim=16, embedder='none').to(dev)
    ms = {'voxel_vertex_idx': vertex_idx.to(dev),
          'voxel_center_xyz': centres.to(dev),
          'voxel_vertex_emb': emb.to(dev)}
    return dec, xyz.to(dev), vox.to(dev), ms, vs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--points', type=int, nargs='*',
                    default=[18000, 36000, 73000, 110000])
    ap.add_argument('--lib', default=None,
                    help='a what-if library (tools/whatif_build.py)')
    a = ap.parse_args()
    if a.lib:
        from xrdslam_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
        print('# library:', a.lib)
    from xrdslam_amd.engine import vox as ev
    dev = 'cuda:0'
    print(f'{"points":>8} {"role":>9} {"fwd":>8} {"bwd":>8} {"dw+red":>8}  '
          '(us, median of %d)' % a.reps)
    for P in a.points:
        for role in ('tracking', 'mapping'):
            dec, xyz, vox, ms, vs = case(P, 1, dev)
            mapping = role == 'mapping'
            for p in dec.parameters():
                p.requires_grad_(mapping)
            ms['voxel_vertex_emb'].requires_grad_(mapping)
            w_s = torch.randn(P, device=dev)
            w_c = torch.randn(P, 3, device=dev)
            ev.PROFILE = None
            for it in range(a.reps + 3):
                if it == 3:
                    ev.PROFILE = {}
                x = xyz.clone().requires_grad_(not mapping)
                out = ev.points(dec, x, vox, ms, vs)
                ((out['sdf'] * w_s).sum() +
                 (out['color'] * w_c).sum()).backward()
            torch.cuda.synchronize()
            t = {}
            for k, evs in ev.PROFILE.items():
                t[k[0]] = float(np.median([e0.elapsed_time(e1) * 1e3
                                           for e0, e1 in evs]))
            ev.PROFILE = None
            print(f'{P:8d} {role:>9} {t.get("vox_points_fwd", 0):8.1f} '
                  f'{t.get("vox_points_bwd", 0):8.1f} '
                  f'{t.get("vox_dw", 0):8.1f}')


if __name__ == '__main__':
    main()
