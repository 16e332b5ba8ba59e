"""world_size-2 gloo test of the data-parallel mapping plumbing
(xrdslam_amd/engine/dist.py): shard sizes, the flat-bucket all-reduce, and the
selected-cell gradient exchange for grid parameters."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from xrdslam_amd.engine import dist as xd
    xd.state.setup('cpu', seed=3)
    assert xd.state.enabled and xd.state.world == world
    assert xd.state.shard_count(1000) == 500 and xd.state.shard_count(7) == 4
    # different ranks draw different shard pixels
    draw = torch.randint(10**6, (4, ), generator=xd.state.shard_generator)
    # dense parameter + a "grid" parameter with selected cells
    dense = torch.nn.Parameter(torch.zeros(5))
    dense.grad = torch.full((5, ), float(rank + 1))
    grid = torch.zeros(1, 32, 2, 2, 3).contiguous(
        memory_format=torch.channels_last_3d).requires_grad_(True)
    grid.grad = torch.full_like(grid, float(rank + 1),
                                memory_format=torch.preserve_format)
    grid._xrd_cells = torch.tensor([1, 7, 8], dtype=torch.int32)
    grid._xrd_grad_fresh = True
    stale = torch.zeros(1, 32, 1, 1, 2).contiguous(
        memory_format=torch.channels_last_3d).requires_grad_(True)
    stale.grad = torch.full_like(stale, 5.0,
                                 memory_format=torch.preserve_format)
    stale._xrd_cells = None
    stale._xrd_grad_fresh = False  # no gradient this iteration: not exchanged
    xd.allreduce_param_grads({'a': [dense], 'g': [grid], 's': [stale]})
    cells = grid.grad.permute(0, 2, 3, 4, 1).reshape(-1, 32)
    # the exchange's bookkeeping (bench.py --gpus N: rccl.bucket_bytes /
    # allreduce_ms): one exchange of 5 dense floats + 3 selected cells x 32
    assert xd.state.stats['exchanges'] == 1
    assert xd.state.stats['bucket_bytes_max'] == 4 * (5 + 3 * 32)
    ms = xd.state.measure_allreduce_ms(xd.state.stats['bucket_bytes_max'],
                                       iters=3)
    assert ms is not None and 0.0 < ms < 1e4
    out[rank] = (dense.grad.clone(), cells.clone(), stale.grad.clone(),
                 draw.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_param_grads_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        dense, cells, stale, _ = out[r]
        assert torch.equal(dense, torch.full((5, ), 3.0))  # 1 + 2
        sel = torch.zeros(12, dtype=torch.bool)
        sel[[1, 7, 8]] = True
        assert torch.equal(cells[sel], torch.full((3, 32), 3.0))
        assert torch.equal(cells[~sel], torch.full((9, 32), float(r + 1)))
        assert torch.equal(stale, torch.full_like(stale, 5.0))
    assert not torch.equal(out[0][3], out[1][3])


def _worker_static(rank, world, port, out):
    """a job list kept with a persistent mapping graph: the static selection
    buffer (capacity = all cells, host-known count) changes between calls"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from xrdslam_amd.engine import dist as xd
    xd.state.setup('cpu', seed=3)
    grid = torch.zeros(1, 32, 2, 2, 3).contiguous(
        memory_format=torch.channels_last_3d).requires_grad_(True)
    grid.grad = torch.zeros_like(grid, memory_format=torch.preserve_format)
    cells_buf = torch.zeros(12, dtype=torch.int32)
    grid._xrd_cells = cells_buf
    grid._xrd_cells_count = torch.zeros(1, dtype=torch.int32)
    grid._xrd_grad_fresh = True
    res = []
    jobs = None
    for sel in ([1, 7, 8], [0, 2, 3, 4, 11]):
        grid.grad.fill_(float(rank + 1))
        cells_buf.zero_()
        cells_buf[:len(sel)] = torch.tensor(sel, dtype=torch.int32)
        grid._xrd_cells_n = len(sel)
        jobs = xd.collect_grad_jobs({'g': [grid]}) if jobs is None \
            else xd.refresh_grad_jobs(jobs)
        xd.run_grad_jobs(jobs)
        res.append(grid.grad.permute(0, 2, 3, 4, 1).reshape(-1, 32).clone())
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_refreshed_grad_jobs_follow_the_static_selection():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_static, args=(world, _free_port(), out), nprocs=world,
             join=True)
    for r in range(world):
        for cells, sel in zip(out[r], ([1, 7, 8], [0, 2, 3, 4, 11])):
            m = torch.zeros(12, dtype=torch.bool)
            m[sel] = True
            assert torch.equal(cells[m], torch.full((len(sel), 32), 3.0))
            assert torch.equal(cells[~m],
                               torch.full((12 - len(sel), 32), float(r + 1)))


def test_exchanged_parameters_follow_the_optimisers_after_surgery():
    """SplaTAM's growth / pruning REPLACES the Parameter objects inside its
    optimisers' groups (gaussian_cloud_splatam.py:114-190 like the reference's
    remove_points / cat_params_to_optimizer).  The gradients the mapping
    all-reduce exchanges must be those of the live parameters, not of the list
    the Optimizers object was built with — with the stale list the ranks'
    clouds drifted apart from the first pruning step of a run (found by a
    2-rank run in round 6: the all-reduce of the next frame failed on
    different Gaussian counts)."""
    from xrdslam_amd.slam.engine.optimizers import (AdamOptimizerConfig,
                                                    Optimizers)
    a = torch.nn.Parameter(torch.zeros(5, 3))
    b = torch.nn.Parameter(torch.zeros(5, 1))
    cfg = {'means3D': {'optimizer': AdamOptimizerConfig(lr=1e-3)},
           'logit_opacities': {'optimizer': AdamOptimizerConfig(lr=1e-2)}}
    opts = Optimizers(cfg, {'means3D': [a], 'logit_opacities': [b]})
    live = opts.stepping_parameters(0)
    assert live['means3D'][0] is a and live['logit_opacities'][0] is b
    # pruning: rows 1, 3 go; the optimiser's group gets a NEW Parameter
    keep = torch.tensor([0, 2, 4])
    a2 = torch.nn.Parameter(a.detach()[keep].clone())
    opts.optimizers['means3D'].param_groups[0]['params'][0] = a2
    live = opts.stepping_parameters(7)
    assert live['means3D'][0] is a2
    assert live['logit_opacities'][0] is b


def test_tile_band_partition_covers_the_image_once():
    """SplaTAM's tile-row bands (engine/dist.tile_band): every pixel row is
    owned by exactly one rank, the rendered tile rows cover the owned rows
    plus the 5-row SSIM halo, for image heights that are and are not
    multiples of the 16-pixel tile and for more ranks than tile rows"""
    from xrdslam_amd.engine.dist import tile_band
    for height in (480, 120, 100, 33):
        gy = (height + 15) // 16
        for world in (1, 2, 3, 4, 8):
            owned = np.zeros(height, np.int64)
            for r in range(world):
                b = tile_band(r, world, height)
                r0, r1 = b['own']
                owned[r0:r1] += 1
                t0, t1 = b['render_tiles']
                if r1 > r0:
                    assert t0 * 16 <= max(0, r0 - 5)
                    assert min(t1 * 16, height) >= min(height, r1 + 5)
                    assert 0 <= t0 < t1 <= gy
                else:
                    assert (t0, t1) == (0, 0)
            assert (owned == 1).all(), (height, world)


def _pose_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from xrdslam_amd.engine import dist as xd
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    xd.state.setup('cpu', seed=3)

    class _Alg:
        device = 'cpu'
    slam = SequentialSLAM(_Alg(), [], pose_device='cpu')
    mine = torch.eye(4) * float(rank + 1)
    as_tensor = slam._sync_pose(mine.clone())          # device pose chain
    as_numpy = slam._sync_pose(mine.numpy().copy())    # host hand-over
    out[rank] = (torch.is_tensor(as_tensor), as_tensor.clone(),
                 isinstance(as_numpy, np.ndarray), torch.from_numpy(as_numpy))
    dist.barrier()
    dist.destroy_process_group()


def test_tracking_result_is_broadcast_where_it_lives():
    """multi-GPU frame loop: tracking is replicated, rank 0's result continues
    on every rank.  A tensor result (the device pose chain) is broadcast as a
    tensor and stays one — no host hop —, a numpy result stays numpy."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_pose_worker, args=(world, _free_port(), out), nprocs=world,
             join=True)
    for r in range(world):
        is_t, t, is_n, n = out[r]
        assert is_t and is_n
        assert torch.equal(t, torch.eye(4)) and torch.equal(n, torch.eye(4))
