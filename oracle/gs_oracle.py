"""TEST INFRASTRUCTURE ONLY — parity oracle for SplaTAM's Gaussian rasteriser.

PARITY UNPINNED: ``diff-gaussian-rasterization-w-depth`` @ cb65e4b (reference
requirements.txt:3) is not vendored under /root/reference and cannot be
imported; no reference test touches it.  This is a dense torch restatement of
the published algorithm: per-Gaussian EWA projection
(+0.3 low-pass, 3-sigma radius, 16x16 tile rectangle), per-pixel front-to-back
blending of depth-sorted Gaussians restricted to the Gaussian's tile rectangle
(alpha = min(.99, o*exp(p)), skip alpha < 1/255, stop before T < 1e-4), colour
and depth outputs.  Gradients come from autograd on this forward; the gradient
the CUDA code reports for the dummy ``means2D`` input is d loss / d ndc.xy.

The clamp ``alpha = min(0.99, o * G)``: the published backward does not
branch on it (``dL_dG = o * dL_dalpha`` and ``dL_do = G * dL_dalpha`` whether
or not alpha saturated), i.e. the gradient PASSES THROUGH the clamp; the
forward below clamps the value and keeps the unclamped derivative (round 2
used torch.clamp, whose autograd zeroes it — the HIP kernel follows the
published code, tests/test_gs_hip.py covers the saturated case).

``window=(x0, y0, w, h)`` evaluates only those pixels (and skips Gaussians
whose tile rectangle misses them): the dense evaluation is O(N H W), a crop of
a 640x480 image with 1e5 Gaussians is seconds."""
from __future__ import annotations

import math

import torch

TILE = 16


def quat_to_rot(q):
    r, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)
    ], -1).reshape(q.shape[:-1] + (3, 3))


def project(means3D, scales, rotations, viewmatrix, projmatrix, H, W,
            tanfovx, tanfovy, scale_modifier=1.0):
    """the per-Gaussian half (preprocess): EWA projection, 3-sigma radius, tile
    rectangle.  -> dict(pix [N,2], conic [N,3], tz [N], radii [N] int,
    visible [N] bool, rect (rminx, rmaxx, rminy, rmaxy), ndc [N,2])"""
    N = means3D.shape[0]
    dev, dt = means3D.device, means3D.dtype
    V = viewmatrix.reshape(4, 4).t()   # w2c
    P = projmatrix.reshape(4, 4).t()   # full projection
    ones = torch.ones(N, 1, dtype=dt, device=dev)
    ph = torch.cat([means3D, ones], 1)
    p_view = (ph @ V.t())[:, :3]
    p_hom = ph @ P.t()
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None]
    ndc.retain_grad() if ndc.requires_grad else None
    # 3-D covariance
    R = quat_to_rot(rotations)
    S2 = (scales * scale_modifier)**2
    Sigma = R @ torch.diag_embed(S2) @ R.transpose(1, 2)
    # EWA
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    # outside 1.3x the field of view the published backward multiplies the
    # x/y gradient by 0 and ignores the z dependence of the clamped value
    rx, ry = p_view[:, 0] / tz, p_view[:, 1] / tz
    tx = torch.where((rx < -limx) | (rx > limx),
                     (torch.clamp(rx, -limx, limx) * tz).detach(),
                     p_view[:, 0])
    ty = torch.where((ry < -limy) | (ry > limy),
                     (torch.clamp(ry, -limy, limy) * tz).detach(),
                     p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -fx * tx / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -fy * ty / (tz * tz)], -1)], 1)
    T = J @ V[:3, :3]
    cov = T @ Sigma @ T.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], -1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    pix = torch.stack([((ndc[:, 0] + 1) * W - 1) * 0.5,
                       ((ndc[:, 1] + 1) * H - 1) * 0.5], -1)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def tclamp(v, hi):
        return torch.clamp(torch.trunc(v / TILE), 0, hi)

    pd = pix.detach()
    rminx, rmaxx = tclamp(pd[:, 0] - radius, gx), tclamp(pd[:, 0] + radius +
                                                         TILE - 1, gx)
    rminy, rmaxy = tclamp(pd[:, 1] - radius, gy), tclamp(pd[:, 1] + radius +
                                                         TILE - 1, gy)
    visible = (tz.detach() > 0.2) & (det.detach() != 0) & \
        ((rmaxx - rminx) * (rmaxy - rminy) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).int()
    return dict(pix=pix, conic=conic, tz=tz, radii=radii, visible=visible,
                rect=(rminx, rmaxx, rminy, rmaxy), ndc=ndc)


def rasterize(means3D, colors, opacities, scales, rotations, viewmatrix,
              projmatrix, H, W, tanfovx, tanfovy, bg=None, scale_modifier=1.0,
              window=None):
    """viewmatrix / projmatrix are the [4,4] TRANSPOSED matrices the reference
    passes (common.py:599,605-606: w2c^T and (P w2c)^T).  Returns
    color [3,H,W], radii [N] int, depth [1,H,W], ndc [N,2] (retain_grad-able)."""
    dev, dt = means3D.device, means3D.dtype
    g = project(means3D, scales, rotations, viewmatrix, projmatrix, H, W,
                tanfovx, tanfovy, scale_modifier)
    pix, conic, tz, radii, visible, ndc = (g['pix'], g['conic'], g['tz'],
                                           g['radii'], g['visible'], g['ndc'])
    rminx, rmaxx, rminy, rmaxy = g['rect']
    order = torch.argsort(tz.detach(), stable=True)
    x0, y0, ww, wh = (0, 0, W, H) if window is None else window
    ys, xs = torch.meshgrid(torch.arange(y0, y0 + wh, device=dev, dtype=dt),
                            torch.arange(x0, x0 + ww, device=dev, dtype=dt),
                            indexing='ij')
    tile_x, tile_y = torch.floor(xs / TILE), torch.floor(ys / TILE)
    Tcur = torch.ones(wh, ww, dtype=dt, device=dev)
    done = torch.zeros(wh, ww, dtype=torch.bool, device=dev)
    C = torch.zeros(3, wh, ww, dtype=dt, device=dev)
    D = torch.zeros(wh, ww, dtype=dt, device=dev)
    # Gaussians whose tile rectangle touches the window (speed only)
    tx0, tx1 = x0 // TILE, (x0 + ww - 1) // TILE
    ty0, ty1 = y0 // TILE, (y0 + wh - 1) // TILE
    touch = visible & (rminx <= tx1) & (rmaxx > tx0) & (rminy <= ty1) & \
        (rmaxy > ty0)
    # the touching Gaussians in depth order, gathered ONCE (an index into an
    # N-sized tensor inside the loop costs an N-sized zero fill in backward)
    sel = order[touch[order]]
    pix_s, conic_s, op_s = pix[sel], conic[sel], opacities[sel, 0]
    col_s, tz_s = colors[sel], tz[sel]
    rx0, rx1, ry0, ry1 = rminx[sel], rmaxx[sel], rminy[sel], rmaxy[sel]
    for j in range(sel.shape[0]):
        in_rect = (tile_x >= rx0[j]) & (tile_x < rx1[j]) & \
            (tile_y >= ry0[j]) & (tile_y < ry1[j])
        dx, dy = pix_s[j, 0] - xs, pix_s[j, 1] - ys
        power = -0.5 * (conic_s[j, 0] * dx * dx + conic_s[j, 2] * dy * dy) - \
            conic_s[j, 1] * dx * dy
        a_raw = op_s[j] * torch.exp(power)
        # value clamped, derivative passed through (see the module docstring)
        alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
        ok = in_rect & ~done & (power.detach() <= 0) & \
            (alpha.detach() >= 1.0 / 255.0)
        test_T = Tcur * (1 - alpha)
        stop = ok & (test_T.detach() < 0.0001)
        done = done | stop
        use = ok & ~stop
        w = torch.where(use, alpha * Tcur, torch.zeros_like(alpha))
        C = C + col_s[j][:, None, None] * w
        D = D + tz_s[j] * w
        Tcur = torch.where(use, test_T, Tcur)
    if bg is not None:
        C = C + Tcur * bg.reshape(3, 1, 1)
    return C, radii, D.unsqueeze(0), ndc
