"""Multi-GPU invariant of every sharded mapping path: tracking,
map maintenance and the optimiser steps are replicated, only gradients are
exchanged — so after every frame all ranks must hold the SAME model (every
parameter / buffer of the model and the estimated poses, compared through an
all-gather of float64 sums and element counts).  Runs the trajectory fixtures'
sequences (tests/c1_util.py) on N ranks.

    XRD_DIST_SAME_GPU=1 XRD_DIST_BACKEND=gloo python -m torch.distributed.run \
        --nproc-per-node 2 --master-addr 127.0.0.1 tools/ranks_identical.py [algo ...]
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def _tensors(obj, depth, seen, out, path):
    """every float tensor reachable from ``obj`` through attributes, dicts and
    lists (the maps live in plain containers: NICE's grid dict, SplaTAM's
    parameter dict, Point-SLAM's point cloud)"""
    if id(obj) in seen or depth < 0:
        return
    seen[id(obj)] = obj    # (kept alive: a freed temporary's id is reused)
    if torch.is_tensor(obj):
        if obj.is_floating_point() and obj.numel():
            out.append((path, obj))
        return
    if isinstance(obj, dict):
        items = [(str(k), v) for k, v in obj.items()]
    elif isinstance(obj, (list, tuple)):
        items = [(str(i), v) for i, v in enumerate(obj)]
    elif isinstance(obj, torch.nn.Module):
        items = [(k, v) for k, v in obj.state_dict().items()] + \
            [(k, v) for k, v in vars(obj).items() if not k.startswith('_')] + \
            [(k, v) for k, v in obj._modules.items()]
    elif hasattr(obj, '__dict__') and not isinstance(obj, type):
        items = [(k, v) for k, v in vars(obj).items()
                 if not k.startswith('_')]
    else:
        return
    for k, v in items:
        # scratch that legitimately differs (workspaces, caches, graph slots)
        if any(w in k.lower() for w in ('ws', 'workspace', 'cache', 'slot',
                                        'graph', 'scratch', 'buf')):
            continue
        _tensors(v, depth - 1, seen, out, f'{path}.{k}')


def checksum(algo):
    found = []
    _tensors(algo.model, 4, {}, found, 'model')
    found.sort(key=lambda kv: kv[0])
    vals = []
    for _, v in found:
        vals += [float(v.numel()), float(v.detach().double().nan_to_num().sum())]
    names = [k for k, _ in found for _ in (0, 1)]
    for i, p in enumerate(algo.get_estimate_c2w_list()):
        vals.append(float(p.detach().double().sum()))
        names.append(f'pose {i}')
    return torch.tensor(vals, dtype=torch.float64), names


def main():
    import bench
    import c1_util
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dev = torch.device('cuda:0' if os.environ.get('XRD_DIST_SAME_GPU') == '1'
                       else f'cuda:{os.environ.get("LOCAL_RANK", 0)}')
    torch.cuda.set_device(dev)
    dist.init_process_group(os.environ.get('XRD_DIST_BACKEND', 'nccl'),
                            rank=rank, world_size=world)
    bench._setup_dist(dev, world)
    names = sys.argv[1:] or ['coslam', 'voxfusion', 'nice', 'pointslam',
                             'splatam']
    bad = 0
    for name in names:
        n = {'nice': 12, 'coslam': 16, 'voxfusion': 8, 'pointslam': 4,
             'splatam': 5}[name]
        est, gt, sec, slam = c1_util.run_engine(name, 0, dev=str(dev),
                                                n_frames=n)
        t, tnames = checksum(slam.algorithm)
        cnt = torch.tensor([t.numel()], dtype=torch.int64)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        same = all(int(c) == int(cnt) for c in cnts)
        if same:
            out = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            same = all(torch.equal(out[0], o) for o in out)
            worst = max(float((out[0] - o).abs().nan_to_num(1e300).max())
                        for o in out)
            if not same and rank == 0:
                for o in out[1:]:
                    for i in torch.nonzero(out[0] != o).flatten()[:12].tolist():
                        print(f'   differs: {tnames[i]}  {float(out[0][i])!r} '
                              f'vs {float(o[i])!r}', flush=True)
        else:
            worst = float('nan')
            allnames = [None] * world
            dist.all_gather_object(allnames, tnames)
            if rank == 0:
                print('   the ranks hold different NUMBERS of tensors:',
                      [int(c) for c in cnts], flush=True)
                for r in range(1, world):
                    extra = set(allnames[r]) ^ set(allnames[0])
                    if extra:
                        print(f'   rank {r} vs rank 0:', sorted(extra),
                              flush=True)
        if rank == 0:
            print(f'{name}: {n} frames on {world} ranks, ATE '
                  f'{c1_util.ate(est, gt) * 100:.2f} cm, {t.numel()} sums: '
                  f'{"IDENTICAL" if same else "DIFFERENT (largest difference of a sum %g)" % worst}',
                  flush=True)
        bad += 0 if same else 1
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
