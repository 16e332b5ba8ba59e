from dataclasses import dataclass


@dataclass
class Camera:
    """pinhole intrinsics (reference: slam/common/camera.py)"""
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int
