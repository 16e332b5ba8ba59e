"""Pre-train the four NICE-SLAM decoders on the synthetic room (run on the GPU
box; seeded, ~2 minutes) -> xrdslam_amd/data/pretrained/nice_decoders_synth.pt

Why: NICE-SLAM tracks against the occupancy prior of PRETRAINED, fixed
middle / fine decoders (slam/models/conv_onet.py:293-322 loads
pretrained/{coarse,middle_fine}.pt; the files in the reference tree are
git-LFS pointers).  With random-init decoders the bench's NICE trajectory
drifts by 0.1-0.2 m, which says nothing about the engine.  This script gives
the decoders such a prior: decoders AND feature grids are optimised jointly
with torch autograd on the CPU oracle's arithmetic (oracle/nice_oracle.py run
on the GPU — a training-time tool, not the product path) over frames 100..199
of the synthetic sequence at their ground-truth poses (the bench runs frames
0..110 of the same room: the prior is scene-specific, as the checkpoint's
name says); only the DECODERS are kept.

    python tools/pretrain_nice_decoders.py [iters [out.pt]]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import nice_oracle as no  # noqa: E402
from xrdslam_amd.data.synthetic import SyntheticRoom  # noqa: E402
from xrdslam_amd.engine import nice as en  # noqa: E402
from xrdslam_amd.slam.model_components.decoder_nice import NICE  # noqa: E402

BOUND = [[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]]
OUT = os.path.join(ROOT, 'xrdslam_amd', 'data', 'pretrained',
                   'nice_decoders_synth.pt')


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
    out_path = sys.argv[2] if len(sys.argv) > 2 else OUT
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    np.random.seed(0)
    H, W, fx, fy, cx, cy = 480, 640, 320.0, 320.0, 319.5, 239.5
    room = SyntheticRoom(BOUND, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy,
                         n_frames=200, device='cpu')
    # the bound the model derives (conv_onet.py:324-337) and its grid shapes
    bound = torch.tensor([[-5.5, 6.0199995], [-6.7, 5.4599998],
                          [-4.7, 5.5399998]], dtype=torch.float64, device=dev)
    shapes = {'grid_coarse': (10, 12, 11), 'grid_middle': (31, 37, 35),
              'grid_fine': (63, 75, 71), 'grid_color': (63, 75, 71)}
    std = {'grid_coarse': 0.01, 'grid_middle': 0.01, 'grid_fine': 0.0001,
           'grid_color': 0.01}
    grids = {k: (torch.randn(1, 32, *s) * std[k]).to(dev).requires_grad_()
             for k, s in shapes.items()}
    net = NICE(coarse=True)
    decs = {}
    for kind, d in net.decoders().items():
        decs[kind] = {n: v.clone().to(dev).requires_grad_()
                      for n, v in d.state_dict().items()}
    frames = []
    for k in range(100, 200, 4):
        it = room[k]
        frames.append((torch.from_numpy(np.asarray(it['c2w'], np.float32))
                       .to(dev),
                       torch.from_numpy(np.asarray(it['depth'], np.float32))
                       .to(dev).reshape(-1),
                       torch.from_numpy(np.asarray(it['rgb'], np.float32))
                       .to(dev).reshape(-1, 3)))
    opt = torch.optim.Adam([
        {'params': [p for kind in ('coarse', 'middle', 'fine')
                    for p in decs[kind].values()], 'lr': 1e-3},
        {'params': list(decs['color'].values()), 'lr': 3e-3},
        {'params': list(grids.values()), 'lr': 2e-2}])
    g = torch.Generator(device=dev).manual_seed(1)
    jj, ii = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32),
                            torch.arange(W, device=dev, dtype=torch.float32),
                            indexing='ij')
    dirs = torch.stack([(ii - cx) / fx, -(jj - cy) / fy, -torch.ones_like(ii)],
                       -1).reshape(-1, 3)
    t0 = time.time()
    for it in range(iters):
        stage = 'coarse' if it % 4 == 3 else 'color'
        sel = torch.randint(len(frames), (4, ), generator=g, device=dev)
        ro, rd, td, tc = [], [], [], []
        for f in sel.tolist():
            c2w, depth, rgb = frames[f]
            pix = torch.randint(H * W, (500, ), generator=g, device=dev)
            ro.append(c2w[:3, 3].expand(500, 3))
            rd.append(dirs[pix] @ c2w[:3, :3].T)
            td.append(depth[pix])
            tc.append(rgb[pix])
        ro, rd = torch.cat(ro), torch.cat(rd)
        td, tc = torch.cat(td).reshape(-1, 1), torch.cat(tc)
        keep = no.inside_mask(ro, rd, td, bound)
        ro, rd, td, tc = ro[keep], rd[keep], td[keep], tc[keep]
        out = no.render_batch_ray(ro, rd, td, grids, decs, bound, stage)
        loss = sum(no.loss_dict(out, td, tc, True, stage).values()) / \
            ro.shape[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if it % 250 == 0 or it == iters - 1:
            print(f'it {it:5d} {stage:6s} loss/ray {float(loss):.4f} '
                  f'({time.time() - t0:.0f} s)', flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    torch.save({kind: {n: v.detach().cpu() for n, v in sd.items()}
                for kind, sd in decs.items()}, out_path)
    print('wrote', out_path, os.path.getsize(out_path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
