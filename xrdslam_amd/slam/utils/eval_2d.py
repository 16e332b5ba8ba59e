"""2-D metrics of a rendered frame against the input frame, the two that need
no third-party network or package: PSNR and the rendered-depth L1 error
(slam/common/common.py:429-479, ``save_render_imgs``).  MS-SSIM
(pytorch_msssim) and LPIPS (torchmetrics + AlexNet weights) are not available
offline and are not restated."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def render_metrics(gt_color: np.ndarray, gt_depth: np.ndarray,
                   color: np.ndarray, depth: Optional[np.ndarray]
                   ) -> Tuple[float, float]:
    """(PSNR [dB], depth L1 [cm]).  Colours are clipped to [0,1]; with a
    rendered depth the PSNR runs over all pixels, without one (SplaTAM-style
    colour-only call) over pixels with valid input depth only — pixels without
    are zeroed on both sides, like the reference does; the depth error is the
    mean |d - d^| over pixels with valid input depth, in centimetres."""
    gt_c = np.clip(np.asarray(gt_color, dtype=np.float32), 0, 1)
    c = np.clip(np.asarray(color, dtype=np.float32), 0, 1)
    gt_d = np.asarray(gt_depth, dtype=np.float32)
    valid = gt_d > 0
    if depth is None:
        gt_c = gt_c * valid[..., None]
        c = c * valid[..., None]
        depth_l1 = 0.0
    else:
        d = np.asarray(depth, dtype=np.float32)
        depth_l1 = float(np.abs(gt_d[valid] - d[valid]).mean()) * 100.0
    mse = float(np.mean((gt_c.astype(np.float64) - c.astype(np.float64))**2))
    psnr = float('inf') if mse == 0 else -10.0 * float(np.log10(mse))
    return psnr, depth_l1
