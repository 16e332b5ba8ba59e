"""TEST INFRASTRUCTURE ONLY — tile-culled evaluation of oracle/gs_oracle.py's
rasteriser ("parity unpinned", see that file): the same per-Gaussian
projection (gs_oracle.project), the same per-pixel blending rule, but every
16x16 tile only sees the Gaussians whose tile rectangle covers it and all
tiles of a chunk are blended at once as [tiles, K, 256] tensors — the dense
oracle walks every Gaussian over every pixel in a Python loop (O(N H W): 50
minutes for four 64x48 SplaTAM frames).  This is what lets the REFERENCE's
SplaTAM loop run at 160x120 over 20 frames for tests/golden/c1_splatam.npz
(oracle/make_golden_c1.py).  Gradients are autograd's on this forward.

Held to the dense oracle (values and every input gradient) by
tests/test_gs_tiled_oracle.py.

The sequential rule of the published kernel, vectorised along a tile's
depth-sorted list (index k):
    ok_k    = power_k <= 0 and alpha_k >= 1/255
    Tb_k    = prod_{j<k, ok_j} (1 - alpha_j)            (exclusive cumprod)
    stop_k  = ok_k and Tb_k (1 - alpha_k) < 1e-4        (the pixel is done
              BEFORE Gaussian k; nothing behind the first stop is used, so
              Tb_k is exact for every k up to it)
    use_k   = ok_k and no stop at any j <= k
    C = sum use_k alpha_k Tb_k colour_k,  D likewise with depth,
    T_final = prod_k (1 - use_k alpha_k)
"""
from __future__ import annotations

import torch

import gs_oracle

TILE = gs_oracle.TILE


def rasterize(means3D, colors, opacities, scales, rotations, viewmatrix,
              projmatrix, H, W, tanfovx, tanfovy, bg=None, scale_modifier=1.0,
              max_elems=2_000_000):
    """same contract as gs_oracle.rasterize (no ``window``): color [3,H,W],
    radii [N] int, depth [1,H,W], ndc [N,2]"""
    dev, dt = means3D.device, means3D.dtype
    g = gs_oracle.project(means3D, scales, rotations, viewmatrix, projmatrix,
                          H, W, tanfovx, tanfovy, scale_modifier)
    pix, conic, tz, radii, visible, ndc = (g['pix'], g['conic'], g['tz'],
                                           g['radii'], g['visible'], g['ndc'])
    rminx, rmaxx, rminy, rmaxy = g['rect']
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    n_tiles = gx * gy
    # visible Gaussians in depth order (stable: ties keep the index order)
    order = torch.argsort(tz.detach(), stable=True)
    sel = order[visible[order]]
    M = sel.shape[0]
    C = D = Tf = None
    if M > 0:
        pix_s, conic_s, op_s = pix[sel], conic[sel], opacities[sel, 0]
        col_s, tz_s = colors[sel], tz[sel]
        r = [v[sel].long() for v in (rminx, rmaxx, rminy, rmaxy)]
        tix = torch.arange(gx, device=dev)
        tiy = torch.arange(gy, device=dev)
        # cover[m, ty, tx]: Gaussian m's rectangle covers tile (ty, tx)
        cx = (tix[None, :] >= r[0][:, None]) & (tix[None, :] < r[1][:, None])
        cy = (tiy[None, :] >= r[2][:, None]) & (tiy[None, :] < r[3][:, None])
        cover = (cy[:, :, None] & cx[:, None, :]).reshape(M, n_tiles)
        counts = cover.sum(0)
        K = int(counts.max())
        # per tile the covering Gaussians, in depth order, padded with M
        # (argsort of "not covering" is stable: covering ones first, in order)
        idx = torch.argsort((~cover).to(torch.int8).t(), dim=1, stable=True)
        idx = idx[:, :K]
        valid = torch.arange(K, device=dev)[None, :] < counts[:, None]
        idx = torch.where(valid, idx, torch.zeros_like(idx))
        py, px = torch.meshgrid(torch.arange(TILE, device=dev, dtype=dt),
                                torch.arange(TILE, device=dev, dtype=dt),
                                indexing='ij')
        py, px = py.reshape(-1), px.reshape(-1)           # [256]
        chunk = max(1, int(max_elems // max(1, K * TILE * TILE)))
        cs, ds, ts = [], [], []
        for t0 in range(0, n_tiles, chunk):
            t1 = min(n_tiles, t0 + chunk)
            kk = max(1, int(counts[t0:t1].max()))
            ix = idx[t0:t1, :kk]                          # [T, kk]
            va = valid[t0:t1, :kk]
            tt = torch.arange(t0, t1, device=dev)
            ox = (tt % gx).to(dt) * TILE
            oy = (tt // gx).to(dt) * TILE
            xs = ox[:, None] + px[None, :]                # [T, 256]
            ys = oy[:, None] + py[None, :]
            gpix, gcon = pix_s[ix], conic_s[ix]           # [T, kk, .]
            dx = gpix[:, :, 0, None] - xs[:, None, :]     # [T, kk, 256]
            dy = gpix[:, :, 1, None] - ys[:, None, :]
            power = -0.5 * (gcon[:, :, 0, None] * dx * dx +
                            gcon[:, :, 2, None] * dy * dy) - \
                gcon[:, :, 1, None] * dx * dy
            a_raw = op_s[ix][:, :, None] * torch.exp(power)
            # value clamped, derivative passed through (gs_oracle docstring)
            alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
            ok = va[:, :, None] & (power.detach() <= 0) & \
                (alpha.detach() >= 1.0 / 255.0)
            om = torch.where(ok, 1 - alpha, torch.ones_like(alpha))
            incl = torch.cumprod(om, dim=1)
            Tb = torch.cat([torch.ones_like(incl[:, :1]), incl[:, :-1]], 1)
            stop = ok & ((Tb * (1 - alpha)).detach() < 0.0001)
            stopped = torch.cummax(stop.to(torch.int8), dim=1)[0] > 0
            use = ok & ~stopped
            w = torch.where(use, alpha * Tb, torch.zeros_like(alpha))
            c = torch.einsum('tkp,tkc->tcp', w, col_s[ix])      # [T, 3, 256]
            d = (w * tz_s[ix][:, :, None]).sum(1)               # [T, 256]
            tf = torch.where(use, 1 - alpha,
                             torch.ones_like(alpha)).prod(1)    # [T, 256]
            cs.append(c)
            ds.append(d)
            ts.append(tf)

        def image(v, ch):        # [tiles, ch, 256] -> [ch, gy*16, gx*16]
            return v.reshape(gy, gx, ch, TILE, TILE).permute(2, 0, 3, 1, 4) \
                .reshape(ch, gy * TILE, gx * TILE)
        C = image(torch.cat(cs), 3)
        D = image(torch.cat(ds)[:, None], 1)[0]
        Tf = image(torch.cat(ts)[:, None], 1)[0]
    else:
        C = torch.zeros(3, gy * TILE, gx * TILE, dtype=dt, device=dev)
        D = torch.zeros(gy * TILE, gx * TILE, dtype=dt, device=dev)
        Tf = torch.ones(gy * TILE, gx * TILE, dtype=dt, device=dev)
    C, D, Tf = C[:, :H, :W], D[:H, :W], Tf[:H, :W]
    if bg is not None:
        C = C + Tf * bg.reshape(3, 1, 1)
    return C, radii, D.unsqueeze(0), ndc
