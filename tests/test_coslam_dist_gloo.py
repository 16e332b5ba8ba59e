"""world_size-2 gloo test of sharded Co-SLAM mapping on the CPU: each rank
renders half of a fixed ray batch with the host mirror of JointEncoding (oracle
encodings), uses the sharded loss (batch-global normalisers all-reduced) and
exchanges gradients through Optimizers' all-reduce; the summed gradients must
equal those of one process rendering the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'oracle'))
sys.path.insert(0, HERE)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model_and_batch():
    import coslam_golden_util as cg
    import tcnn_standin
    import xrdslam_amd.slam.model_components.encodings_coslam as enc
    enc.tcnn = tcnn_standin.module()
    g = np.load(cg.GOLDEN)
    model = cg.build_model(g, 'cpu')
    draws = torch.from_numpy(g['map/rand0'])
    batch = {k: torch.from_numpy(g[k]) for k in
             ('rays_o', 'rays_d', 'target_s', 'target_d')}
    return model, batch, draws


def _grads(model):
    return {'table': model.embed_fn.params.grad.clone(),
            **{k: p.grad.clone() for k, p in model.decoder.named_parameters()}}


def _run(model, batch, draws, lo, hi, sharded):
    model._rand = lambda shape, like: draws[lo:hi].to(like) \
        if tuple(shape) == (hi - lo, draws.shape[1]) else \
        torch.full(shape, 0.37, dtype=like.dtype)
    for p in model.parameters():
        p.grad = None
    inp = {k: v[lo:hi] for k, v in batch.items()}
    inp.update(first=True, sharded=sharded)   # 'first': no smoothness term
    out = model.get_outputs(inp)
    losses = model.get_loss_dict(out, inp, True, 0)
    sum(losses.values()).backward()
    return {k: float(v.detach()) for k, v in losses.items()}


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from xrdslam_amd.engine import dist as xd
    xd.state.setup('cpu', seed=1)
    model, batch, draws = _model_and_batch()
    n = batch['rays_o'].shape[0]
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    losses = _run(model, batch, draws, lo, hi, sharded=True)
    xd.allreduce_param_grads({'embed_fn': [model.embed_fn.params],
                              'decoder': list(model.decoder.parameters())})
    out[rank] = (_grads(model), losses)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])   # 3: uneven shards
def test_sharded_coslam_mapping_equals_single_process(world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    model, batch, draws = _model_and_batch()
    n = batch['rays_o'].shape[0]
    full_losses = _run(model, batch, draws, 0, n, sharded=False)
    full = _grads(model)
    # per-rank losses add up to the single-process loss terms
    for k, v in full_losses.items():
        s = sum(out[r][1][k] for r in range(world))
        assert abs(s - v) < 1e-5 * max(abs(v), 1e-6), (k, s, v)
    for r in range(world):
        for k, gfull in full.items():
            g = out[r][0][k]
            err = (g - gfull).abs().max() / gfull.abs().max().clamp(min=1e-30)
            assert err < 1e-4, (r, k, float(err))
