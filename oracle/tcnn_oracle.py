"""TEST INFRASTRUCTURE ONLY — parity oracle for the tiny-cuda-nn encodings that
Co-SLAM uses (slam/model_components/encodings_coslam.py:43-53 HashGrid,
:68-75 OneBlob).

PARITY UNPINNED: tiny-cuda-nn (NVlabs, unpinned git HEAD in the reference's
requirements.txt:5) is not vendored under /root/reference, cannot be imported
here, and the reference has no test touching it.  This file restates the
published algorithm in torch so that autograd
provides the gradient oracle; the HIP kernels are tested against it plus
self-consistency properties (partition of unity, finite differences, the level
resolution table measured in SURVEY §8a-A10)."""
from __future__ import annotations

import math

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)


def hash_levels(n_levels, base_resolution, per_level_scale, log2_hashmap_size,
                n_feat=2, dense=False):
    """per level: (scale f32, resolution, params_in_level, offset); tcnn
    grid.h: scale = exp2f(l*log2f(pls))*base - 1; res = ceilf(scale)+1;
    params = min(align8(res^3), 2^log2_T) (dense: no cap)."""
    log2_pls = np.log2(np.float32(per_level_scale)).astype(np.float32)
    out, off = [], 0
    cap = (1 << log2_hashmap_size)
    for lvl in range(n_levels):
        scale = np.float32(np.exp2(np.float32(lvl) * log2_pls) *
                           np.float32(base_resolution) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        n = res**3
        n = (n + 7) // 8 * 8
        if not dense:
            n = min(n, cap)
        out.append((float(scale), res, n, off))
        off += n
    return out, off


def hashgrid_forward(x, params, levels, n_feat=2):
    """x [N,3] in [0,1] f32; params flat [total*F]; returns [N, L*F]"""
    x = x.float()
    outs = []
    table = params.reshape(-1, n_feat)
    for scale, res, n, off in levels:
        pos = x * scale + 0.5
        cell = torch.floor(pos)
        w = pos - cell
        cell = cell.to(torch.int64)
        acc = 0
        for corner in range(8):
            bits = [(corner >> d) & 1 for d in range(3)]
            cg = [cell[:, d] + bits[d] for d in range(3)]
            wt = 1.0
            for d in range(3):
                wt = wt * (w[:, d] if bits[d] else 1.0 - w[:, d])
            # grid_index: dense stride sum while stride <= n, else hash
            stride, idx = 1, torch.zeros_like(cg[0])
            for d in range(3):
                if stride <= n:
                    idx = idx + cg[d] * stride
                    stride *= res
            if n < stride:
                h = torch.zeros_like(cg[0])
                for d in range(3):
                    h = h ^ ((cg[d] * PRIMES[d]) & 0xFFFFFFFF)
                idx = h
            idx = (idx & 0xFFFFFFFF) % n
            acc = acc + wt[:, None] * table[off + idx]
        outs.append(acc)
    return torch.cat(outs, 1)


def quartic_cdf(x, inv_radius):
    u = x * inv_radius
    u2 = u * u
    u4 = u2 * u2
    return torch.clamp((15.0 / 16.0) * u * (1 - (2.0 / 3.0) * u2 +
                                            (1.0 / 5.0) * u4) + 0.5, 0.0, 1.0)


def oneblob_forward(x, n_bins=16):
    """x [N,D] in [0,1]; returns [N, D*n_bins] (dimension-major)"""
    x = x.float()
    outs = []

    def cdf3(t):
        return quartic_cdf(t, n_bins) + quartic_cdf(t - 1.0, n_bins) + \
            quartic_cdf(t + 1.0, n_bins)

    for d in range(x.shape[1]):
        xd = x[:, d]
        left = cdf3(-xd)
        cols = []
        for k in range(n_bins):
            right = cdf3((k + 1) / n_bins - xd)
            cols.append(right - left)
            left = right
        outs.append(torch.stack(cols, 1))
    return torch.cat(outs, 1)
