"""LR schedulers of the reference (slam/engine/schedulers.py:45-112): LambdaLR
stage switches.  The factor functions are exposed (``factor(step)``) so the
fused Adam path can evaluate them without a torch scheduler object."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Type

from torch.optim import Optimizer, lr_scheduler

from ..configs.base_config import InstantiateConfig


@dataclass
class SchedulerConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: Scheduler)


class Scheduler:
    def __init__(self, config) -> None:
        self.config = config

    def factor(self, step: int) -> float:
        raise NotImplementedError

    def get_scheduler(self, optimizer: Optimizer, lr_init: float):
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=self.factor)


@dataclass
class LRconfig:
    coarse: float = 0.0
    middle: float = 0.0
    fine: float = 0.0
    color: float = 0.005


@dataclass
class NiceSLAMSchedulerConfig(SchedulerConfig):
    _target: Type = field(default_factory=lambda: NiceSLAMScheduler)
    coarse: bool = True
    middle_iter_ratio: float = 0.4
    fine_iter_ratio: float = 0.6
    stage_lr: LRconfig = field(default_factory=LRconfig)
    max_steps: int = 1000


class NiceSLAMScheduler(Scheduler):
    """schedulers.py:67-86: factor = stage_lr of the stage `step` falls in"""

    def factor(self, step: int) -> float:
        c = self.config
        if c.coarse:
            return c.stage_lr.coarse
        if step <= c.max_steps * c.middle_iter_ratio:
            return c.stage_lr.middle
        if step <= c.max_steps * c.fine_iter_ratio:
            return c.stage_lr.fine
        return c.stage_lr.color


@dataclass
class PointSLAMSchedulerConfig(SchedulerConfig):
    _target: Type = field(default_factory=lambda: PointSLAMScheduler)
    geo_iter_ratio: float = 0.4
    start_lr: float = 0.001
    end_lr: float = 0.005
    max_steps: int = 1000


class PointSLAMScheduler(Scheduler):
    """schedulers.py:98-112"""

    def factor(self, step: int) -> float:
        c = self.config
        return c.start_lr if step <= c.max_steps * c.geo_iter_ratio \
            else c.end_lr
