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

from typing import List, Optional

import torch
import torch.distributed as dist


class DistState:
    def __init__(self):
        self.enabled = False
        self.rank = 0
        self.world = 1
        self.shard_generator: Optional[torch.Generator] = None

    def setup(self, device, seed=0):
        self.enabled = dist.is_available() and dist.is_initialized() and \
            dist.get_world_size() > 1
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
        """rays this rank draws out of n (ceil split, every rank the same)"""
        return (n + self.world - 1) // self.world if self.enabled else n


state = DistState()


def allreduce_bucket(tensors: List[torch.Tensor]) -> None:
    """SUM all-reduce of a list of (possibly strided) gradient tensors through
    one flat bucket; results are written back in place."""
    if not state.enabled or not tensors:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


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
    sels = [g if cells is None else g[cells] for g, cells, _ in cell_jobs]
    allreduce_bucket(list(dense) + sels)
    for (g, cells, _), sel in zip(cell_jobs, sels):
        if cells is not None:
            g[cells] = sel


def allreduce_param_grads(param_groups) -> None:
    """exchange the gradients the optimisers are about to consume"""
    if not state.enabled:
        return
    run_grad_jobs(collect_grad_jobs(param_groups))
