"""SplaTAM with tile-band sharded mapping on N ranks: every rank must hold the
SAME cloud after every frame (Gaussian count, parameter sums, last pose are
all-gathered and compared) — growth, pruning and tracking are replicated, the
Gaussian gradients all-reduced.  Found round 6's stale-parameter-list bug
(the ranks' clouds drifted apart from the first pruning step).

    XRD_DIST_SAME_GPU=1 XRD_DIST_BACKEND=gloo python -m torch.distributed.run \\
        --nproc-per-node 2 --master-addr 127.0.0.1 tools/splatam_two_ranks.py
    DBG_GRAPHS=0: eager iterations, also reports inside the mapping calls"""
import os, sys, random
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench
import torch.distributed as dist

def main():
    rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import cadence, splatam_config
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    cam = Camera(**bench.CAM)
    algo = splatam_config().setup(camera=cam, device=str(dev))
    algo.use_graphs = os.environ.get('DBG_GRAPHS', '1') == '1'
    bench._setup_dist(dev, world)
    data = bench._CvPoses(SyntheticRoom(bench.CO_BOUND, H=cam.height, W=cam.width, fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy, n_frames=200, device=dev))
    cad = cadence['splaTAM']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every, keyframe_every=cad.keyframe_every, pose_device=str(dev), use_relative_pose=cad.use_relative_pose)
    def report(tag):
        torch.cuda.synchronize()
        gc = algo.model.gaussian_cloud
        vals = [float(gc.params['means3D'].shape[0])]
        for k in ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales'):
            vals.append(float(gc.params[k].detach().double().sum()))
        est = algo.get_estimate_c2w_list()
        vals.append(float(est[len(est)-1].double().sum()) if len(est) else 0.0)
        t = torch.tensor(vals, dtype=torch.float64)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        if rank == 0:
            same = all(torch.equal(out[0], o) for o in out)
            print(tag, 'SAME' if same else 'DIFF', [o.tolist() for o in out] if not same else out[0].tolist()[:2], flush=True)
    # hook the iteration to report inside mapping
    orig = algo._iteration
    state = {'k': 0}
    def wrapped(optimizers, frames, is_mapping, step, *a, **kw):
        r = orig(optimizers, frames, is_mapping, step, *a, **kw)
        if is_mapping and state['k'] >= 1 and step in (0, 1, 2, 19, 20, 21, 59) and os.environ.get('DBG_GRAPHS', '1') != '1':
            report(f'  frame {state["k"]} map it {step}')
        return r
    algo._iteration = wrapped
    for k in range(4):
        state['k'] = k
        try:
            slam.step(k)
        except Exception as e:
            print(rank, 'EXC', repr(e)[:300], flush=True)
            break
        report(f'frame {k}')
    dist.destroy_process_group()
main()
