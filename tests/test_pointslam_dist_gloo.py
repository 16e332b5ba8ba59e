"""world_size-2/3 gloo test of sharded Point-SLAM mapping on the CPU.  The
mapping losses are plain sums over rays (conv_onet_pointslam.py:190-204) and a
ray's render only depends on the shared cloud, so a rank that renders its
share of a batch and exchanges gradients (engine/dist.allreduce_param_grads:
one flat bucket, SUM) must end up with the gradients of the single-process
batch: map features (geometry, colour) and the colour decoder.  Neighbour
search: the exact brute-force stand-in (oracle/faiss_standin.py)."""
import os
import socket
import sys

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


def _setup():
    import faiss_standin
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.models.conv_onet_pointslam import (ConvOnet2,
                                                             ConvOnet2Config)
    torch.manual_seed(0)
    model = ConvOnet2(ConvOnet2Config(mapping_pixels_based_on_color_grad=40),
                      Camera(40., 40., 31.5, 23.5, 64, 48))
    model.knn_factory = faiss_standin.TorchKNN
    g = torch.Generator().manual_seed(2)
    n = 240
    d = torch.randn(n, 3, generator=g) * 0.2 + torch.tensor([0., 0., -1.])
    o = torch.zeros(n, 3)
    depth = 1.5 + 0.3 * torch.rand(n, generator=g)
    color = torch.rand(n, 3, generator=g)
    r = torch.full((n, ), 0.08)
    model.model_update({
        'batch_rays_o': o, 'batch_rays_d': d, 'batch_gt_depth': depth,
        'batch_gt_color': color, 'batch_dynamic_r': r,
        'batch_rays_o_grad': o[:40], 'batch_rays_d_grad': d[:40],
        'batch_gt_depth_grad': depth[:40], 'batch_gt_color_grad': color[:40],
        'batch_dynamic_r_grad': r[:40]})
    npc = model.neural_point_cloud
    with torch.no_grad():
        npc.geo_feats.normal_(0, 0.3, generator=g)
        npc.col_feats.normal_(0, 0.3, generator=g)
    fixed = torch.randn(32, generator=g) * 0.01
    model.decoder.geo_decoder.empty_feature_fn = lambda c, dv: fixed
    model.decoder.color_decoder.empty_feature_fn = lambda c, dv: fixed
    td = depth * (1 + 0.02 * torch.randn(n, generator=g))
    batch = {'rays_o': o, 'rays_d': d, 'target_s': color, 'target_d': td,
             'batch_dynamic_r': torch.full((n, ), 0.16)}
    return model, batch


def _groups(model):
    npc = model.neural_point_cloud
    return {'geometry': [npc.geo_feats], 'color': [npc.col_feats],
            'decoder': list(model.decoder.color_decoder.parameters())}


def _run(model, batch, idx):
    groups = _groups(model)
    for ps in groups.values():
        for p in ps:
            p.grad = None
    inp = {k: v[idx] for k, v in batch.items()}
    inp.update(stage='color', depth_positive=True)
    out = model(inp)
    losses = model.get_loss_dict(out, inp, True, 'color')
    sum(losses.values()).backward()
    return {k: float(v.detach()) for k, v in losses.items()}


def _grads(model):
    return [p.grad.clone() for ps in _groups(model).values() for p in ps]


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from xrdslam_amd.engine import dist as xd
    xd.state.setup('cpu', seed=1)
    model, batch = _setup()
    n = batch['rays_o'].shape[0]
    losses = _run(model, batch, torch.arange(n)[rank::world])
    xd.allreduce_param_grads(_groups(model))
    out[rank] = (_grads(model), losses)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])   # 3: uneven shards
def test_sharded_pointslam_mapping_equals_single_process(world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    model, batch = _setup()
    n = batch['rays_o'].shape[0]
    full_losses = _run(model, batch, torch.arange(n))
    full = _grads(model)
    for k, v in full_losses.items():
        s = sum(out[r][1][k] for r in range(world))
        assert abs(s - v) < 1e-5 * max(abs(v), 1e-6), (k, s, v)
    for r in range(world):
        for i, gfull in enumerate(full):
            g = out[r][0][i]
            err = (g - gfull).abs().max() / gfull.abs().max().clamp(min=1e-30)
            assert err < 1e-4, (r, i, float(err))
