"""HIP-event timing of the fused Co-SLAM render kernels at tracking / mapping
batch sizes (synthetic rays in the office0 bound, trained-like random table)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, nargs='+', default=[1024, 2560])
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    from bench import BOUND, CAM
    from xrdslam_amd.engine import coslam as ec
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.models.joint_encoding import (JointEncoding,
                                                        JointEncodingConfig)
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = JointEncoding(JointEncodingConfig(cam_depth_trunc=100.,
                                              tcnn_encoding=True),
                          Camera(**CAM),
                          torch.from_numpy(np.array(BOUND, dtype=np.float64))
                          ).to(dev)
    with torch.no_grad():
        model.embed_fn.params.normal_(0, 0.05)
    tab = model._fused_tables(dev)
    out = {}
    for n in args.rays:
        ro = (torch.rand(n, 3, device=dev) - 0.5) * 2.0
        rd = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
        td = 1.0 + 2.0 * torch.rand(n, 1, device=dev)
        rnd = torch.rand(n, 43, device=dev)
        for mode in ('track', 'map', 'map_ba'):
            ro_ = ro.clone().requires_grad_(mode != 'map')
            rd_ = rd.clone().requires_grad_(mode != 'map')

            def fwd():
                return ec.render(model, tab, ro_, rd_, td, rnd,
                                 train_map=mode != 'track')

            res = fwd()
            # loss-like gradients: depth + colour + sdf channel in front/near
            def loss(r):
                z, raw = r['z_vals'], r['raw']
                near = (z < td + 0.1).float()
                return (r['rgb'].sum() + r['depth'].sum() +
                        (raw[..., 3] * near).sum())
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tf = tb = 0.0
            for it in range(args.reps + 3):
                model.zero_grad(set_to_none=True)
                ev[0].record()
                r = fwd()
                ev[1].record()
                l = loss(r)
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                l.backward()
                e1.record()
                torch.cuda.synchronize()
                if it >= 3:
                    tf += ev[0].elapsed_time(ev[1])
                    tb += e0.elapsed_time(e1)
            out[f'{mode}_{n}'] = {'fwd_us': tf / args.reps * 1e3,
                                  'bwd_incl_loss_us': tb / args.reps * 1e3}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
