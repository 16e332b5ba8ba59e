"""Config base classes with the reference's contract
(slam/configs/base_config.py:28-37): ``config.setup(**kw)`` instantiates
``config._target(config, **kw)``."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Type


class PrintableConfig:
    def __str__(self):
        body = ', '.join(f'{k}={v!r}' for k, v in vars(self).items())
        return f'{type(self).__name__}({body})'


@dataclass
class InstantiateConfig(PrintableConfig):
    _target: Type = None

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)
