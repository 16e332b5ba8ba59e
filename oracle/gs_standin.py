"""TEST INFRASTRUCTURE ONLY.  CPU stand-in for the
``diff_gaussian_rasterization`` extension evaluated by oracle/gs_oracle.py
("parity unpinned", see that file).  Used to execute the reference's SplaTAM
model on the CPU for tests/golden/splatam_render.npz and by the CPU test of
the host mirror.  The dummy ``means2D`` input receives no gradient here (the
densification statistics that read it are off in the reference's defaults).
``TILED = True`` evaluates the same rasteriser tile by tile (oracle/gs_tiled.py,
held to the dense evaluation by tests/test_gs_tiled_oracle.py): what the
trajectory fixture needs to run the reference loop at 160x120."""
import types
from typing import NamedTuple

import torch
import torch.nn as nn

import gs_oracle
import gs_tiled

TILED = False


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, colors_precomp=None,
                scales=None, rotations=None, shs=None, cov3D_precomp=None):
        rs = self.raster_settings
        fn = gs_tiled.rasterize if TILED else gs_oracle.rasterize
        color, radii, depth, _ = fn(
            means3D, colors_precomp, opacities, scales, rotations,
            rs.viewmatrix.reshape(4, 4), rs.projmatrix.reshape(4, 4),
            rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy,
            bg=rs.bg, scale_modifier=rs.scale_modifier)
        return color, radii, depth


def module():
    m = types.ModuleType('diff_gaussian_rasterization')
    m.GaussianRasterizationSettings = GaussianRasterizationSettings
    m.GaussianRasterizer = GaussianRasterizer
    return m
