"""``Model`` base class — the plugin contract of slam/models/base_model.py:23-70:
``populate_modules``, ``forward = get_outputs(input: dict) -> dict``,
``get_loss_dict(outputs, inputs, is_mapping, stage)``,
``get_param_groups() -> Dict[str, List[Parameter]]``."""
from __future__ import annotations

from abc import abstractmethod
from dataclasses import dataclass, field
from typing import Dict, List, Type, Union

import torch
from torch import nn
from torch.nn import Parameter

from ..configs.base_config import InstantiateConfig


@dataclass
class ModelConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: Model)


class Model(nn.Module):
    config: ModelConfig

    def __init__(self, config, camera, bounding_box=None, **kwargs) -> None:
        super().__init__()
        self.config, self.camera = config, camera
        self.bounding_box, self.kwargs = bounding_box, kwargs
        self.populate_modules()

    @property
    def device(self):
        return self.device_indicator_param.device

    @abstractmethod
    def populate_modules(self):
        self.device_indicator_param = nn.Parameter(torch.empty(0))

    def forward(self, input) -> Dict[str, Union[torch.Tensor, List]]:
        return self.get_outputs(input)

    @abstractmethod
    def get_loss_dict(self, outputs, inputs, is_mapping,
                      stage=None) -> Dict[str, torch.Tensor]:
        pass

    @abstractmethod
    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        pass

    @abstractmethod
    def get_outputs(self, input) -> Dict[str, Union[torch.Tensor, List]]:
        pass
