"""world_size 2 / 3 gloo test of NICE-SLAM's deterministic mapping shards
(engine/dist.py, NiceSLAM.get_model_input): with the shared RNG stream every
rank draws the same batch, keeps a contiguous slice of every frame's rays, and
carries the WHOLE batch's max kept depth (it bounds the sampling range,
conv_onet.py:418,455); the slices tile the single-process batch exactly."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.common.frame import Frame
    cam = Camera(80., 80., 31.5, 23.5, 64, 48)
    g = torch.Generator().manual_seed(41)
    frames = []
    for k in range(3):
        depth = (0.5 + 3.5 * torch.rand(48, 64, generator=g)).numpy() \
            .astype(np.float32)          # some depths beyond the small bound
        color = torch.rand(48, 64, 3, generator=g).numpy().astype(np.float32)
        c2w = np.eye(4, dtype=np.float32)
        c2w[:3, 3] = [0.1 * k, -0.05 * k, 0.02 * k]
        frames.append(Frame(fid=k, rgb=color, depth=depth, init_pose=c2w,
                            gt_pose=c2w, separate_LR=False, rot_rep='quat'))
    bound = torch.tensor([[-2.0, 2.0], [-2.0, 2.0], [-2.0, 2.0]])
    cfg = types.SimpleNamespace(tracking_sample=150, tracking_Hedge=4,
                                tracking_Wedge=6, mapping_sample=400,
                                min_sample_pixels=50)
    from xrdslam_amd.slam.algorithms.nice_slam import NiceSLAM
    me = types.SimpleNamespace(
        config=cfg, camera=cam, device='cpu', bounding_box=bound,
        stage='color', bundle_adjust=True, fixed_shape_batches=True,
        model=types.SimpleNamespace(device='cpu'))
    me._shard_rows = types.MethodType(NiceSLAM._shard_rows, me)
    return NiceSLAM, me, frames


def _batch(NiceSLAM, me, frames):
    torch.manual_seed(2)   # the stream tracking keeps in lock-step
    out = NiceSLAM.get_model_input(me, frames, True)
    return {k: (v.detach().clone() if torch.is_tensor(v) else v)
            for k, v in out.items()}


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from xrdslam_amd.engine import dist as xd
    xd.state.setup('cpu', seed=3)
    assert xd.state.enabled and xd.state.deterministic
    out[rank] = _batch(*_setup())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_shards_tile_the_single_process_batch(world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    full = _batch(*_setup())          # single process, same seed
    n_frames = 3
    n_pix = full['rays_o'].shape[0] // n_frames
    keep = full['ray_mask']
    dmax = torch.where(keep, full['target_d'].squeeze(-1),
                       torch.zeros(())).max()
    for key in ('rays_o', 'rays_d', 'target_s', 'target_d', 'ray_mask'):
        whole = full[key].reshape(n_frames, n_pix, *full[key].shape[1:])
        parts = []
        for r in range(world):
            lo, hi = (n_pix * r) // world, (n_pix * (r + 1)) // world
            parts.append(out[r][key].reshape(n_frames, hi - lo,
                                             *full[key].shape[1:]))
        assert torch.equal(torch.cat(parts, 1), whole), key
    for r in range(world):
        assert torch.equal(out[r]['dmax'], dmax)
