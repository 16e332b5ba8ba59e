"""Plugin contract of a SLAM model (what slam/models/base_model.py:23-70 asks
of its subclasses):

* ``populate_modules()``  build sub-modules (called once from ``__init__``)
* ``get_outputs(input: dict) -> dict``  the forward pass (``forward`` calls it)
* ``get_loss_dict(outputs, inputs, is_mapping, stage) -> dict of tensors``
* ``get_param_groups() -> {group name: [Parameter, ...]}`` for ``Optimizers``

``device`` is wherever the (empty) indicator parameter lives, so
``model.to(dev)`` moves it."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Type, Union

import torch
from torch import nn
from torch.nn import Parameter

from ..configs.base_config import InstantiateConfig

TensorDict = Dict[str, Union[torch.Tensor, List]]


@dataclass
class ModelConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: Model)


class Model(nn.Module):
    config: ModelConfig

    def __init__(self, config, camera, bounding_box=None, **kwargs) -> None:
        nn.Module.__init__(self)
        self.kwargs = kwargs
        self.camera, self.bounding_box = camera, bounding_box
        self.config = config
        self.populate_modules()

    def populate_modules(self):
        # subclasses extend this; the indicator parameter tracks the device
        self.device_indicator_param = nn.Parameter(torch.empty(0))

    @property
    def device(self):
        return self.device_indicator_param.device

    def forward(self, input) -> TensorDict:
        return self.get_outputs(input)

    # -- to be provided by the algorithm's model --------------------------------
    def get_outputs(self, input) -> TensorDict:
        raise NotImplementedError

    def get_loss_dict(self, outputs, inputs, is_mapping,
                      stage=None) -> Dict[str, torch.Tensor]:
        raise NotImplementedError

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        raise NotImplementedError
