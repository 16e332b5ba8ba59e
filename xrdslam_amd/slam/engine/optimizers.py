"""``Optimizers``: one optimizer (+ optional scheduler) per parameter group, with
the reference's interface and semantics (slam/engine/optimizers.py:63-171):
group names must exist in the config (RuntimeError otherwise), ``accum_step``
gradient accumulation, ``max_norm`` clipping, fresh Adam state per instance.

MI355X addition: a parameter that carries ``_xrd_cells`` (an int32 list of
selected 32-float cells of a channel-last feature grid; ``None`` = every cell)
is stepped by the fused ``xrd_adam_cells`` kernel in place — arithmetically the
same as the reference's Adam over the 1-D ``val[mask]`` parameter plus its two
whole-grid ``index_put`` round trips per iteration (conv_onet.py:94-114).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple, Type

import torch
from torch.nn.parameter import Parameter

from ..configs.base_config import PrintableConfig


@dataclass
class OptimizerConfig(PrintableConfig):
    _target: Type = torch.optim.Adam
    lr: float = 0.0005
    eps: float = 1e-08
    betas: Tuple[float, float] = (0.9, 0.999)
    max_norm: Optional[float] = None
    accum_step: Optional[int] = None

    def setup(self, params) -> torch.optim.Optimizer:
        kwargs = {k: v for k, v in vars(self).items()
                  if k not in ('_target', 'max_norm', 'accum_step')}
        if len(params) == 1 and hasattr(params[0], '_xrd_cells') and \
                self._target is torch.optim.Adam and \
                not kwargs.get('weight_decay', 0):
            return FusedCellAdam(params, lr=self.lr, betas=self.betas,
                                 eps=self.eps)
        if self._target is torch.optim.Adam and len(params) > 0 and \
                all(p.is_cuda and p.dtype == torch.float32 for p in params):
            # one launch per parameter, device-side step counter (replayable
            # from a hipGraph); same arithmetic as torch.optim.Adam
            from ...engine.slam_ops import FusedDenseAdam
            return FusedDenseAdam(params, **kwargs)
        return self._target(params, **kwargs)


@dataclass
class AdamOptimizerConfig(OptimizerConfig):
    _target: Type = torch.optim.Adam
    weight_decay: float = 0


@dataclass
class RAdamOptimizerConfig(OptimizerConfig):
    _target: Type = torch.optim.RAdam
    weight_decay: float = 0


class FusedCellAdam(torch.optim.Optimizer):
    """Adam over the selected cells of ONE channel-last grid parameter, in
    place, through ``xrd_adam_cells``.  Like torch.optim.Adam it does nothing
    while the parameter has received no gradient (``p.grad is None`` there,
    ``_xrd_grad_fresh`` unset here), so the per-parameter step count starts
    when the stage that feeds the grid starts."""

    def __init__(self, params, lr, betas, eps):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._m = self._v = self._step_dev = None

    def zero_grad(self, set_to_none: bool = True):
        # the kernel clears the used gradient cells itself
        if torch.cuda.is_current_stream_capturing():
            return
        for g in self.param_groups:
            for p in g['params']:
                p._xrd_grad_fresh = False

    def _launch_args(self):
        """what this step would hand to xrd_adam_cells_tick, or None when
        there is nothing to do (no fresh gradient, empty selection)"""
        grp = self.param_groups[0]
        p = grp['params'][0]
        if not getattr(p, '_xrd_grad_fresh', False) or p.grad is None:
            return None
        cells = p._xrd_cells
        # static selection (persistent mapping graphs): ``cells`` is a buffer
        # with room for every cell, the valid count lives on the device
        count = getattr(p, '_xrd_cells_count', None) if cells is not None \
            else None
        cf = p.shape[1]
        n = int(cells.numel()) if cells is not None else p.numel() // cf
        if n == 0 and count is None:
            # empty frustum selection: torch's Adam over an empty val[mask]
            # does nothing (and does not advance this parameter's step)
            p._xrd_grad_fresh = False
            return None
        if self._m is None:
            self._m = torch.zeros(n * cf, dtype=torch.float32, device=p.device)
            self._v = torch.zeros_like(self._m)
            # {steps taken, ticket}: the kernel advances it (replayable from
            # a hipGraph, no separate increment launch)
            self._step_dev = torch.zeros(2, dtype=torch.int32,
                                         device=p.device)
        b1, b2 = grp['betas']
        return dict(p=p, cells=cells, count=count, n=n, cf=cf,
                    lr=float(grp['lr']), b1=float(b1), b2=float(b2),
                    eps=float(grp['eps']))

    def _stepped(self):
        if not torch.cuda.is_current_stream_capturing():
            self.param_groups[0]['params'][0]._xrd_grad_fresh = False

    @torch.no_grad()
    def step(self, closure=None):
        from ... import _lib
        a = self._launch_args()
        if a is None:
            return
        p = a['p']
        _lib.check(_lib.lib().xrd_adam_cells_tick(
            _lib.ptr(p), _lib.ptr(p.grad), _lib.ptr(self._m),
            _lib.ptr(self._v), _lib.ptr(a['cells']), a['n'], a['cf'],
            a['lr'], a['b1'], a['b2'], a['eps'],
            _lib.ptr(self._step_dev), _lib.ptr(a['count']), 1,
            _lib.stream_ptr(p.device)), 'xrd_adam_cells_tick')
        self._stepped()

    @staticmethod
    @torch.no_grad()
    def step_together(opts):
        """the steps of several grids' optimisers as ONE launch
        (xrd_adam_cells_multi) where their cell sizes / betas / eps agree;
        -> the optimisers this call has stepped (or found nothing to do for)"""
        from ... import _lib
        done, ready = [], []
        for o in opts:
            a = o._launch_args()
            done.append(o)
            if a is not None:
                ready.append((o, a))
        while ready:
            o0, a0 = ready[0]
            key = (a0['cf'], a0['b1'], a0['b2'], a0['eps'], a0['p'].device)
            group = [(o, a) for o, a in ready
                     if (a['cf'], a['b1'], a['b2'], a['eps'],
                         a['p'].device) == key][:_lib.ADAM_MAX_SETS]
            ready = [x for x in ready if all(x[0] is not g[0] for g in group)]
            if len(group) == 1:
                o0.step()
                continue
            sets = (_lib.AdamCellsSet * len(group))()
            for k, (o, a) in enumerate(group):
                p = a['p']
                sets[k].param, sets[k].grad = _lib.ptr(p), _lib.ptr(p.grad)
                sets[k].m, sets[k].v = _lib.ptr(o._m), _lib.ptr(o._v)
                sets[k].cell_idx = _lib.ptr(a['cells'])
                sets[k].n_cells, sets[k].lr = a['n'], a['lr']
                sets[k].step_ticket = _lib.ptr(o._step_dev)
                sets[k].n_cells_dev = _lib.ptr(a['count'])
            _lib.check(_lib.lib().xrd_adam_cells_multi(
                len(group), sets, a0['cf'], a0['b1'], a0['b2'], a0['eps'], 1,
                _lib.stream_ptr(a0['p'].device)), 'xrd_adam_cells_multi')
            for o, _ in group:
                o._stepped()
        return done


def reset_optimizer_state(opt: torch.optim.Optimizer) -> None:
    """back to the state of a freshly built optimiser, keeping every tensor
    (and its address: the launches may live in a captured hipGraph)"""
    if isinstance(opt, FusedCellAdam):
        for t in (opt._m, opt._v, opt._step_dev):
            if t is not None:
                t.zero_()
        return
    for st in opt.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()


class Optimizers:
    def __init__(self, config: Dict[str, Any] = None,
                 param_groups: Dict[str, List[Parameter]] = None,
                 optimizers: Dict[str, Any] = None) -> None:
        self.config = config
        self.schedulers = {}
        if optimizers:
            self.optimizers = optimizers
            return
        self.optimizers = {}
        self.parameters = {}
        for name, params in param_groups.items():
            if name not in config:
                raise RuntimeError(
                    f"Optimizer config for '{name}' not found in config file. "
                    'Make sure you specify an optimizer for each parameter '
                    f'group. Provided configs were: {config.keys()}')
            ocfg = config[name]['optimizer']
            self.optimizers[name] = ocfg.setup(params=params)
            self.parameters[name] = params
            if config[name].get('scheduler'):
                self.schedulers[name] = config[name]['scheduler'].setup(
                ).get_scheduler(optimizer=self.optimizers[name],
                                lr_init=ocfg.lr)

    def __add__(self, other: 'Optimizers') -> 'Optimizers':
        # only used by Co-SLAM (persistent model optimizer + pose optimizers)
        return Optimizers(config={**self.config, **other.config},
                          optimizers={**self.optimizers, **other.optimizers})

    def optimizer_step(self, param_group_name: str) -> None:
        self.optimizers[param_group_name].step()

    def scheduler_step(self, param_group_name: str) -> None:
        if param_group_name in self.schedulers:
            self.schedulers[param_group_name].step()

    def zero_grad_all(self) -> None:
        for name, opt in self.optimizers.items():
            if self.config[name]['optimizer'].accum_step is None:
                opt.zero_grad(set_to_none=True)

    def stepping_parameters(self, step: int):
        """parameter groups whose optimiser steps at ``step``: gradients of
        an accumulating group (accum_step) stay local until its step, then
        their SUM is exchanged once (the all-reduce is linear)"""
        # (the LIVE parameters of each optimiser: SplaTAM's growth / pruning
        # replaces the Parameter objects inside the optimiser's groups, and
        # the list this object was built with goes stale — its gradients were
        # exchanged instead of the live ones and the ranks drifted apart from
        # the first pruning step on)
        return {
            n: [p for g in self.optimizers[n].param_groups
                for p in g['params']]
            for n in getattr(self, 'parameters', {})
            if n in self.optimizers and (
                self.config[n]['optimizer'].accum_step is None or
                (step + 1) % self.config[n]['optimizer'].accum_step == 0)}

    def optimizer_step_all(self, step: int, exchange: bool = True) -> None:
        from ...engine import dist as _dist
        if exchange and _dist.state.enabled and \
                getattr(self, 'parameters', None) and \
                getattr(self, 'allreduce', False):
            _dist.allreduce_param_grads(self.stepping_parameters(step))
        # the feature grids of a stage step in one launch
        together = [opt for name, opt in self.optimizers.items()
                    if isinstance(opt, FusedCellAdam) and
                    self.config[name]['optimizer'].max_norm is None and
                    self.config[name]['optimizer'].accum_step is None]
        stepped = FusedCellAdam.step_together(together) \
            if len(together) > 1 else []
        # ... and the dense tensors that step now (one optimiser a parameter
        # group: five for SplaTAM's cloud, table + decoder + poses for the
        # others) in one launch too
        from ...engine.slam_ops import FusedDenseAdam
        dense = [opt for name, opt in self.optimizers.items()
                 if isinstance(opt, FusedDenseAdam) and
                 self.config[name]['optimizer'].max_norm is None and
                 self.config[name]['optimizer'].accum_step is None]
        if len(dense) > 1:
            FusedDenseAdam.step_together(dense)
            stepped = list(stepped) + dense
        for name, opt in self.optimizers.items():
            if any(opt is o for o in stepped):
                continue
            ocfg = self.config[name]['optimizer']
            if ocfg.max_norm is not None:
                torch.nn.utils.clip_grad_norm_(self.parameters[name],
                                               ocfg.max_norm)
            if ocfg.accum_step is None:
                opt.step()
            elif (step + 1) % ocfg.accum_step == 0:
                opt.step()
                # static_grads (captured graphs): the accumulation buffer
                # keeps its address and is zeroed in place
                opt.zero_grad(
                    set_to_none=not getattr(self, 'static_grads', False))

    def scheduler_step_all(self) -> None:
        for sch in self.schedulers.values():
            sch.step()

    def load_optimizers(self, loaded_state: Dict[str, Any]) -> None:
        for k, v in loaded_state.items():
            self.optimizers[k].load_state_dict(v)

    def load_schedulers(self, loaded_state: Dict[str, Any]) -> None:
        for k, v in loaded_state.items():
            self.schedulers[k].load_state_dict(v)
