"""Data-parallel mapping across the GPUs of one node (new
functionality, the reference is single-GPU).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).
Every rank holds a full replica of the map and the decoders.  Per mapping
iteration each rank renders 1/world of the sampled rays; the NICE-SLAM mapping
losses are plain sums (conv_onet.py:178-184), so ONE all-reduce (SUM) of a flat
bucket holding the selected-cell grid gradients, the decoder gradient and the
bundle-adjustment pose gradients makes the following Adam step identical on all
ranks.  Only the cells selected by the frustum mask are exchanged (a few MB),
not the 87 MiB of dense grids.  Tracking does not shard (single frame, batch
global median in the loss): every rank tracks redundantly with the same RNG
stream and stays in lock-step.
"""
from __future__ import annotations

import os
import time
import warnings
from typing import List, Optional

import torch
import torch.distributed as dist


class RcclComm:
    """the C-ABI's own RCCL communicator (csrc/comm.hip: xrd_comm_*,
    xrd_allreduce_grads): collectives are enqueued on torch's CURRENT stream,
    between the kernels that produce and consume the bucket.  The unique id
    travels over the existing torch.distributed group; RCCL itself is the
    instance PyTorch already loaded (its bundled librccl.so)."""

    def __init__(self, device):
        import ctypes as C

        from .. import _lib
        lib = _lib.lib()
        bundled = os.path.join(os.path.dirname(torch.__file__), 'lib',
                               'librccl.so')
        path = bundled if os.path.exists(bundled) else None
        _lib.check(lib.xrd_comm_load(path.encode() if path else None),
                   'xrd_comm_load')
        nbytes = lib.xrd_comm_unique_id_bytes()
        buf = (C.c_char * nbytes)()
        rank, world = (dist.get_rank(), dist.get_world_size()) \
            if dist.is_initialized() else (0, 1)
        if rank == 0:
            _lib.check(lib.xrd_comm_unique_id(buf), 'xrd_comm_unique_id')
        if world > 1:
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0)
            buf.raw = box[0]
        torch.cuda.set_device(device)
        self.handle = lib.xrd_comm_create(buf, rank, world)
        if not self.handle:
            raise _lib.XrdError('xrd_comm_create: ' +
                                (lib.xrd_last_error() or b'').decode())
        self.world = lib.xrd_comm_world(self.handle)
        self.device = torch.device(device)

    def all_reduce_sum(self, flat: torch.Tensor) -> None:
        from .. import _lib
        assert flat.is_cuda and flat.dtype == torch.float32 and \
            flat.is_contiguous()
        _lib.check(_lib.lib().xrd_allreduce_grads(
            self.handle, _lib.ptr(flat), flat.numel(),
            _lib.stream_ptr(flat.device)), 'xrd_allreduce_grads')

    def all_reduce_max_i32(self, values: torch.Tensor) -> None:
        from .. import _lib
        assert values.is_cuda and values.dtype == torch.int32 and \
            values.is_contiguous()
        _lib.check(_lib.lib().xrd_allreduce_max_i32(
            self.handle, _lib.ptr(values), values.numel(),
            _lib.stream_ptr(values.device)), 'xrd_allreduce_max_i32')

    def close(self):
        if self.handle:
            from .. import _lib
            _lib.lib().xrd_comm_destroy(self.handle)
            self.handle = None


def tile_band(rank: int, world: int, height: int, tile: int = 16,
              halo_px: int = 5):
    """Tile-band partition of one image over ranks (SplaTAM mapping, SURVEY
    8e): rank r OWNS the pixel rows [row0, row1) = tile rows [t0, t1) of an
    even split of the ceil(height / tile) tile rows, and RENDERS the tile rows
    [r0, r1) that cover its rows plus ``halo_px`` rows on either side (the
    11 x 11 SSIM window of a pixel it owns reaches 5 rows into its neighbours'
    bands).  -> dict(own=(row0, row1), render_tiles=(r0, r1))"""
    gy = (height + tile - 1) // tile
    t0, t1 = (gy * rank) // world, (gy * (rank + 1)) // world
    row0, row1 = t0 * tile, min(t1 * tile, height)
    if t1 <= t0:
        return {'own': (0, 0), 'render_tiles': (0, 0)}
    r0 = max(0, (row0 - halo_px) // tile)
    r1 = min(gy, (row1 + halo_px + tile - 1) // tile)
    return {'own': (row0, row1), 'render_tiles': (r0, r1)}


class DistState:
    def __init__(self):
        self.enabled = False
        self.rank = 0
        self.world = 1
        self.shard_generator: Optional[torch.Generator] = None
        self.comm: Optional[RcclComm] = None
        # deterministic sharding: every rank draws the SAME mapping batch from
        # the shared RNG stream and renders a contiguous 1/world slice of it,
        # so the summed gradients equal the single-GPU ones (up to summation
        # order); off = every rank draws its own 1/world from its own stream
        self.deterministic = os.environ.get('XRD_DIST_DETERMINISTIC',
                                            '1') != '0'
        self._bucket = None
        # host-side bookkeeping of the gradient exchange (what bench.py
        # --gpus N reports per rank): number of exchanges enqueued from Python
        # (a captured iteration counts once, its replays do not pass here) and
        # the size of the flat bucket
        self.stats = {'exchanges': 0, 'bucket_bytes_last': 0,
                      'bucket_bytes_max': 0}

    def note_exchange(self, n_floats: int) -> None:
        st = self.stats
        st['exchanges'] += 1
        st['bucket_bytes_last'] = 4 * int(n_floats)
        st['bucket_bytes_max'] = max(st['bucket_bytes_max'], 4 * int(n_floats))

    def measure_allreduce_ms(self, nbytes: int, iters: int = 20):
        """mean time of ONE sum all-reduce of ``nbytes`` on the exchange path
        (what a mapping iteration pays for its gradient bucket), in ms"""
        if not self.enabled or nbytes <= 0:
            return None
        dev = self.comm.device if self.comm is not None else (
            torch.device('cuda', torch.cuda.current_device())
            if dist.get_backend() == 'nccl' else torch.device('cpu'))
        flat = torch.zeros(max(1, nbytes // 4), dtype=torch.float32,
                           device=dev)
        for _ in range(3):
            allreduce_flat(flat)
        if dev.type == 'cuda':
            torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            allreduce_flat(flat)
        if dev.type == 'cuda':
            torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / iters * 1e3

    def setup(self, device, seed=0):
        self.enabled = dist.is_available() and dist.is_initialized() and \
            dist.get_world_size() > 1
        dev = torch.device(device)
        if self.enabled and dev.type == 'cuda' and self.comm is None and \
                dist.get_backend() == 'nccl' and \
                os.environ.get('XRD_RCCL_ABI', '1') != '0':
            try:
                self.comm = RcclComm(dev)
            except Exception as e:  # keep the run alive on torch's own group
                warnings.warn(f'C-ABI RCCL communicator unavailable ({e}); '
                              'using torch.distributed collectives')
                self.comm = None
        if self.enabled:
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
            self.shard_generator = torch.Generator(device=device)
            self.shard_generator.manual_seed(seed * 1000 + 17 + self.rank)
            # replicated decisions (keyframe window: random.sample, overlap
            # selection: numpy) must come out the same on every rank
            import random

            import numpy as np
            random.seed(seed * 7919 + 13)
            np.random.seed(seed * 7919 + 13)

    def shard_count(self, n: int) -> int:
        """rays this rank draws out of n in the NON-deterministic mode (ceil
        split, every rank the same: world * ceil(n / world) >= n rays in
        total, so the summed gradient is that of a slightly larger batch; the
        deterministic mode tiles the batch exactly, ``shard_slice``)"""
        return (n + self.world - 1) // self.world if self.enabled else n

    def shard_slice(self, n: int):
        """[lo, hi) of this rank in a batch of n that every rank holds
        (deterministic sharding; the slices tile the batch exactly)"""
        if not self.enabled:
            return 0, n
        return (n * self.rank) // self.world, \
            (n * (self.rank + 1)) // self.world

    def bucket(self, n: int, device) -> torch.Tensor:
        """the persistent flat exchange buffer (grown, never shrunk)"""
        b = self._bucket
        if b is None or b.numel() < n or b.device != torch.device(device):
            b = torch.empty(max(n, 1 << 16), dtype=torch.float32,
                            device=device)
            self._bucket = b
        return b[:n]

    def backend_name(self) -> str:
        if not self.enabled:
            return 'single process'
        if self.comm is not None:
            return 'RCCL through the C-ABI (xrd_allreduce_grads)'
        return f'torch.distributed {dist.get_backend()}'

    def measure_busbw(self, nbytes: int = 8 << 20, iters: int = 20):
        """ring all-reduce bus bandwidth of the exchange path in GB/s:
        2 (N-1)/N x bytes / time, over ``iters`` all-reduces of ``nbytes``"""
        if not self.enabled:
            return None
        dev = self.comm.device if self.comm is not None else (
            torch.device('cuda', torch.cuda.current_device())
            if dist.get_backend() == 'nccl' else torch.device('cpu'))
        flat = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
        for _ in range(3):
            allreduce_flat(flat)
        if dev.type == 'cuda':
            torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            allreduce_flat(flat)
        if dev.type == 'cuda':
            torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / iters
        return 2.0 * (self.world - 1) / self.world * nbytes / dt / 1e9


state = DistState()


def allreduce_flat(flat: torch.Tensor) -> None:
    """SUM all-reduce of one contiguous float32 tensor, in place"""
    if not state.enabled:
        return
    if state.comm is not None and flat.is_cuda:
        state.comm.all_reduce_sum(flat)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)


def allreduce_bucket(tensors: List[torch.Tensor]) -> None:
    """SUM all-reduce of a list of (possibly strided) gradient tensors through
    the persistent flat bucket; results are written back in place."""
    if not state.enabled or not tensors:
        return
    total = sum(t.numel() for t in tensors)
    state.note_exchange(total)
    flat = state.bucket(total, tensors[0].device)
    views, off = [], 0
    for t in tensors:
        views.append(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    with torch.no_grad():
        torch._foreach_copy_(views, list(tensors))
        allreduce_flat(flat)
        torch._foreach_copy_(list(tensors), views)


def collect_grad_jobs(param_groups):
    """what an exchange touches: (dense gradient tensors, [(cell-major view of
    a grid gradient, selected cell ids)]).  For grid parameters only the
    selected cells (``_xrd_cells``) travel.  The returned tensors alias the
    gradients, so a job list collected in the iteration that a hipGraph is
    captured from stays valid for its replays (static addresses)."""
    dense, cell_jobs = [], []
    for params in param_groups.values():
        for p in params:
            if p.grad is None:
                continue
            if hasattr(p, '_xrd_cells'):
                if not getattr(p, '_xrd_grad_fresh', False):
                    continue
                g = p.grad.permute(0, 2, 3, 4, 1).reshape(-1, p.shape[1])
                cell_jobs.append((g, _selected_cells(p), p))
            else:
                dense.append(p.grad)
    return dense, cell_jobs


def _selected_cells(p):
    cells = p._xrd_cells
    if cells is None:
        return None
    if getattr(p, '_xrd_cells_count', None) is not None:
        # static selection buffer (capacity = all cells): the host knows how
        # many entries the current mapping call selected
        cells = cells[:p._xrd_cells_n]
    return cells.long()


def refresh_grad_jobs(jobs):
    """a job list kept with a persistent mapping graph, for a new mapping
    call: same gradient tensors, the call's own cell selection"""
    dense, cell_jobs = jobs
    return dense, [(g, _selected_cells(p), p) for g, _, p in cell_jobs]


def run_grad_jobs(jobs) -> None:
    """SUM all-reduce of one flat bucket holding every job's gradients"""
    if not state.enabled:
        return
    dense, cell_jobs = jobs
    if not dense and not cell_jobs:
        return
    # one persistent bucket: dense gradients are copied in, the selected cells
    # of a grid gradient are gathered straight into their slice
    rows = [g.shape[0] if cells is None else int(cells.numel())
            for g, cells, _ in cell_jobs]
    total = sum(t.numel() for t in dense) + sum(
        r * g.shape[1] for r, (g, _, _) in zip(rows, cell_jobs))
    ref = dense[0] if dense else cell_jobs[0][0]
    state.note_exchange(total)
    flat = state.bucket(total, ref.device)
    off, dviews, cviews = 0, [], []
    for t in dense:
        dviews.append(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    for r, (g, cells, _) in zip(rows, cell_jobs):
        cviews.append(flat[off:off + r * g.shape[1]].view(r, g.shape[1]))
        off += r * g.shape[1]
    with torch.no_grad():
        if dense:
            torch._foreach_copy_(dviews, list(dense))
        for v, (g, cells, _) in zip(cviews, cell_jobs):
            if cells is None:
                v.copy_(g)
            else:
                torch.index_select(g, 0, cells, out=v)
        allreduce_flat(flat)
        if dense:
            torch._foreach_copy_(list(dense), dviews)
        for v, (g, cells, _) in zip(cviews, cell_jobs):
            if cells is None:
                g.copy_(v)
            else:
                g[cells] = v


def allreduce_param_grads(param_groups) -> None:
    """exchange the gradients the optimisers are about to consume"""
    if not state.enabled:
        return
    run_grad_jobs(collect_grad_jobs(param_groups))
