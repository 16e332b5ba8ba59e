"""``diff_gaussian_rasterization``-compatible module backed by the HIP
rasteriser (csrc/gs_raster.hip).

Interface used by the reference (slam/common/common.py:606-618,
slam/model_components/gaussian_cloud_splatam.py:63-69,267-268):

    settings = GaussianRasterizationSettings(image_height, image_width,
        tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
        sh_degree, campos, prefiltered)
    color, radii, depth = GaussianRasterizer(raster_settings=settings)(
        means3D=..., means2D=..., opacities=..., colors_precomp=...,
        scales=..., rotations=...)

autograd reaches means3D, colors_precomp, opacities, scales, rotations and the
dummy ``means2D`` (its gradient = d loss / d ndc.xy, read by SplaTAM's
densification statistics, slam_external_splatam.py:99-103).  The depth output
carries no gradient (SplaTAM renders (z,1,z^2) as colours for that).  Only the
``colors_precomp`` / ``scales+rotations`` path is built (sh_degree 0,
no cov3D_precomp), which is what the reference uses."""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from .. import _lib


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


# The settings are constant for the life of a GaussianCloud (one object per
# cloud, gaussian_cloud_splatam.py:37), so the device->host copies of
# bg / viewmatrix / projmatrix are paid once per settings object, not on every
# forward and backward.  The cache keeps the settings object alive next to its
# camera so that an id() is never reused while its entry exists.
_CAM_CACHE: "dict[int, tuple]" = {}
_CAM_CACHE_MAX = 32


def _cam_key(rs):
    """the reference extension reads the settings' tensors on every call: a
    caller that updates viewmatrix / projmatrix / bg IN PLACE (same
    NamedTuple) must not keep rendering with the cached host copy"""
    return tuple((t.data_ptr(), t._version) for t in
                 (rs.viewmatrix, rs.projmatrix, rs.bg))


def _camera(rs: GaussianRasterizationSettings) -> _lib.GsCamera:
    hit = _CAM_CACHE.get(id(rs))
    key = _cam_key(rs)
    if hit is not None and hit[0] is rs and hit[2] == key:
        return hit[1]
    cam = _build_camera(rs)
    if len(_CAM_CACHE) >= _CAM_CACHE_MAX:
        _CAM_CACHE.pop(next(iter(_CAM_CACHE)))
    _CAM_CACHE[id(rs)] = (rs, cam, key)
    return cam


def _build_camera(rs: GaussianRasterizationSettings) -> _lib.GsCamera:
    cam = _lib.GsCamera()
    cam.image_height, cam.image_width = int(rs.image_height), \
        int(rs.image_width)
    cam.tanfovx, cam.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
    bg = rs.bg.detach().float().cpu().reshape(-1).tolist()
    vm = rs.viewmatrix.detach().float().cpu().reshape(-1).tolist()
    pm = rs.projmatrix.detach().float().cpu().reshape(-1).tolist()
    for i in range(3):
        cam.bg[i] = bg[i]
    cam.scale_modifier = float(rs.scale_modifier)
    for i in range(16):
        cam.viewmatrix[i] = vm[i]
        cam.projmatrix[i] = pm[i]
    return cam


# tile-band sharding (engine/dist.py, SplaTAM mapping over ranks): (first tile
# row, one past the last) this process rasterises, or None = the whole image
BAND = None

# bench.py: per-launch HIP-event timing, key -> [(start, end)], plus the
# (pixel, contributing Gaussian) pair count of every forward pass
PROFILE = None


class _Timed:
    def __init__(self, key):
        self.key = key if PROFILE is not None else None

    def __enter__(self):
        if self.key is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.key is not None:
            self.e1.record()
            PROFILE.setdefault(self.key, []).append((self.e0, self.e1))


class _Binning:
    """static capacity of the (Gaussian, tile) pair list per device.  A pass
    leaves its true pair count in a device scalar; it is copied to pinned
    host memory without waiting and read when the NEXT pass sizes its list
    (by then it has long arrived).  The host waits only when there is no
    usable history: first pass, or the number of Gaussians changed by more
    than 2 % (densification / pruning: once per frame, not per pass).

    SplaTAM's mapping alternates between views with different pair counts
    (same Gaussians transformed to the current frame or a keyframe): the
    capacity is therefore a HIGH-WATER mark — HEADROOM x the largest count
    seen while the number of Gaussians stayed within 2 % — and never follows
    a smaller view down.  A pass that still overflows (xrd_gs_bin drops the
    pairs beyond the capacity) is a hard condition: it is reported with a
    warning one pass later (the count arrives asynchronously), the capacity
    grows to fit it, and ``overflowed`` counts it.  ``exact = True`` sizes
    every pass from its own count with a host sync — what the reference
    extension does (it reads num_rendered back in every forward)."""
    HEADROOM = 1.3
    exact = False

    def __init__(self):
        self.state = {}
        self.overflowed = 0     # passes that dropped pairs

    def capacity(self, dev, n, tiles):
        key = str(dev)
        st = self.state.get(key)
        if st is not None and torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture: the list size is frozen at what the
            # eager passes before it established (the pass count of every
            # replay is folded into ``peak``, see check_replays)
            return st['cap']
        total = None
        if st is not None and not self.exact:
            if st['event'] is not None:
                st['event'].synchronize()      # previous pass: done long ago
                st['event'] = None
                st['last'] = int(st['host'][0])
                if st['last'] > st['last_cap']:
                    self.overflowed += 1
                    import warnings
                    warnings.warn(
                        f'Gaussian rasteriser: the previous pass produced '
                        f'{st["last"]} (Gaussian, tile) pairs for a list of '
                        f'{st["last_cap"]}: pairs were dropped from its '
                        'image and gradients.  Capacity raised; set '
                        'xrdslam_amd.compat.diff_gaussian_rasterization.'
                        '_BIN.exact = True for a per-pass exact size.')
            same_n = abs(n - st['n']) <= 0.02 * max(st['n'], 1)
            if st['last'] is not None and same_n:
                st['high'] = max(st.get('high', 0), st['last'])
                total = st['high']
            else:
                st['high'] = 0
        if total is None:
            total = int(tiles.sum().item()) if n > 0 else 0   # host sync
            if st is not None:
                st['high'] = max(st.get('high', 0), total)
        if st is not None and st.get('floor') is not None:
            fn, ftotal = st['floor']
            if abs(n - fn) <= 0.02 * max(fn, 1):
                total = max(total, ftotal)       # see reserve()
            else:
                st['floor'] = None
        need = max(int(total * self.HEADROOM) + 1024, 1 << 16)
        if st is None:
            st = self.state[key] = {
                'cap': need, 'n': n, 'event': None, 'last': None,
                'last_cap': need, 'ws': None, 'high': total,
                'peak': torch.zeros(1, dtype=torch.int64, device=dev),
                'host': torch.zeros(1, dtype=torch.int64).pin_memory()}
        if need > st['cap'] or need < st['cap'] // 3:
            st['cap'] = need
        return st['cap']

    def reserve(self, dev, n, total):
        """the caller KNOWS a view of these ``n`` Gaussians produces ``total``
        pairs (SplaTAM sizes the list from the largest count over ALL frames
        of the mapping window before any iteration is captured: a captured
        iteration renders a different keyframe on every replay through the
        same frozen list).  Holds as a floor while n stays within 2 %."""
        need = max(int(total * self.HEADROOM) + 1024, 1 << 16)
        key = str(dev)
        st = self.state.get(key)
        if st is None:
            st = self.state[key] = {
                'cap': need, 'n': n, 'event': None, 'last': None,
                'last_cap': need, 'ws': None, 'high': total,
                'peak': torch.zeros(1, dtype=torch.int64, device=dev),
                'host': torch.zeros(1, dtype=torch.int64).pin_memory()}
        st['floor'] = (n, total)
        st['high'] = max(st.get('high', 0), total)
        if need > st['cap']:
            st['cap'] = need

    def workspace(self, dev, nbytes):
        st = self.state[str(dev)]
        if st['ws'] is None or st['ws'].numel() < nbytes:
            st['ws'] = torch.empty(int(nbytes * 1.2) + 256, dtype=torch.uint8,
                                   device=dev)
        return st['ws']

    def report(self, dev, n, cap, n_keys):
        st = self.state[str(dev)]
        if torch.cuda.is_current_stream_capturing():
            torch.maximum(st['peak'], n_keys, out=st['peak'])
            st['graph_cap'] = min(cap, st.get('graph_cap', cap))
            return
        st['n'], st['last_cap'] = n, cap
        if self.exact:
            st['event'] = None
            return
        st['host'].copy_(n_keys, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        st['event'] = ev


    def check_replays(self, dev):
        """after captured passes were replayed: the largest pair count any
        of them produced (one host read) against the frozen list size"""
        st = self.state.get(str(dev))
        if st is None or 'graph_cap' not in st:
            return
        peak = int(st['peak'].item())
        st['peak'].zero_()
        cap = st.pop('graph_cap')
        st['high'] = max(st.get('high', 0), peak)
        if peak > cap:
            self.overflowed += 1
            import warnings
            warnings.warn(
                f'Gaussian rasteriser: a replayed pass produced {peak} '
                f'(Gaussian, tile) pairs for a list of {cap}: pairs were '
                'dropped from its image and gradients.  Capacity raised for '
                'the following passes.')


_BIN = _Binning()


def _geometry(lib, cam, dev, st, m3, sc, rt, op):
    """preprocess + tile binning of one view: everything that does not
    depend on the colours"""
    H, W = cam.image_height, cam.image_width
    n = m3.shape[0]
    f = dict(dtype=torch.float32, device=dev)
    i = dict(dtype=torch.int32, device=dev)
    # xrd_gs_preprocess writes every row (zeros for culled Gaussians)
    depths = torch.empty(n, **f)
    xy = torch.empty(n, 2, **f)
    conic_o = torch.empty(n, 4, **f)
    radii = torch.empty(n, **i)
    rect = torch.empty(n, 4, **i)
    tiles = torch.empty(n, **i)
    with _Timed('gs_preprocess'):
        _lib.check(lib.xrd_gs_preprocess(
            C.byref(cam), n, _lib.ptr(m3), _lib.ptr(sc), _lib.ptr(rt),
            _lib.ptr(op), _lib.ptr(depths), _lib.ptr(xy),
            _lib.ptr(conic_o), _lib.ptr(radii), _lib.ptr(rect),
            _lib.ptr(tiles), st), 'xrd_gs_preprocess')
    if BAND is not None:
        # multi-GPU mapping: this rank rasterises tile rows [BAND[0], BAND[1])
        _lib.check(lib.xrd_gs_band_clip(
            n, int(BAND[0]), int(BAND[1]), _lib.ptr(rect), _lib.ptr(tiles),
            st), 'xrd_gs_band_clip')
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ranges = torch.empty(gx * gy, 2, **i)
    # tile binning on the stream (scan, key duplication, radix sort,
    # ranges: csrc/gs_bin.hip) into a static-capacity list; the capacity
    # follows the pair count of the previous pass, read back
    # asynchronously — no host sync in the pass
    cap = _BIN.capacity(dev, n, tiles)
    plist = torch.empty(cap, **i)
    n_keys = torch.empty(1, dtype=torch.int64, device=dev)
    ws = _BIN.workspace(dev, lib.xrd_gs_bin_ws_bytes(n, cap, W, H))
    # inverse map of the sort + the scan (the lane-per-Gaussian blend backward
    # sums a Gaussian's per-key gradient rows through them)
    # [0,cap): sorted position -> pre-sort pair index; [cap,2cap): pre-sort
    # pair index -> Gaussian id, -1 for a pair dropped by a full list
    key_pos = torch.empty(2 * cap, **i)
    offsets = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    with _Timed('gs_bin'):
        _lib.check(lib.xrd_gs_bin2(
            n, W, H, _lib.ptr(rect), _lib.ptr(tiles), _lib.ptr(depths),
            cap, _lib.ptr(ws), _lib.ptr(plist), _lib.ptr(ranges),
            _lib.ptr(n_keys), _lib.ptr(key_pos), _lib.ptr(offsets), st),
            'xrd_gs_bin2')
    _BIN.report(dev, n, cap, n_keys)
    return depths, xy, conic_o, radii, ranges, plist, n_keys, \
        (cap, key_pos, offsets)


def count_pairs(raster_settings, means3D, scales, rotations, opacities):
    """number of (Gaussian, tile) pairs a render of this view would bin, as a
    0-d int64 device tensor: the preprocess launch alone, no binning, no
    blend, no host sync"""
    lib = _lib.lib()
    cam = _camera(raster_settings)
    dev = means3D.device
    st = _lib.stream_ptr(dev)
    n = means3D.shape[0]
    f = dict(dtype=torch.float32, device=dev)
    i = dict(dtype=torch.int32, device=dev)
    depths, xy = torch.empty(n, **f), torch.empty(n, 2, **f)
    conic_o, radii = torch.empty(n, 4, **f), torch.empty(n, **i)
    rect, tiles = torch.empty(n, 4, **i), torch.empty(n, **i)
    _lib.check(lib.xrd_gs_preprocess(
        C.byref(cam), n, _lib.ptr(means3D.detach().float().contiguous()),
        _lib.ptr(scales.detach().float().contiguous()),
        _lib.ptr(rotations.detach().float().contiguous()),
        _lib.ptr(opacities.detach().float().contiguous()), _lib.ptr(depths),
        _lib.ptr(xy), _lib.ptr(conic_o), _lib.ptr(radii), _lib.ptr(rect),
        _lib.ptr(tiles), st), 'xrd_gs_preprocess')
    return tiles.sum(dtype=torch.int64)


# the blend: True = forward with per-bucket checkpoints + lane-per-Gaussian
# backward (csrc/gs_blend.hip); False = round 2's pixel-per-thread kernels
# (csrc/gs_raster.hip: wave reductions + atomics in the backward)
BUCKET_BLEND = True


def _blend_fwd(lib, cam, dev, st, H, W, ranges, plist, xy, ca, cb, conic_o,
               depths, cap):
    f = dict(dtype=torch.float32, device=dev)
    i = dict(dtype=torch.int32, device=dev)
    color_a = torch.empty(3, H, W, **f)
    color_b = torch.empty(3, H, W, **f) if cb is not None else None
    depth = torch.empty(1, H, W, **f)
    final_T = torch.empty(H, W, **f)
    n_contrib = torch.empty(H, W, **i)
    ckpt = None
    with _Timed('gs_render_fwd'):
        if BUCKET_BLEND:
            ckpt = torch.empty(lib.xrd_gs_blend_ckpt_floats(cap, W, H), **f)
            _lib.check(lib.xrd_gs_blend_fwd(
                C.byref(cam), _lib.ptr(ranges), _lib.ptr(plist), _lib.ptr(xy),
                _lib.ptr(ca), _lib.ptr(cb), _lib.ptr(conic_o),
                _lib.ptr(depths), _lib.ptr(color_a), _lib.ptr(color_b),
                _lib.ptr(depth), _lib.ptr(final_T), _lib.ptr(n_contrib),
                _lib.ptr(ckpt), st), 'xrd_gs_blend_fwd')
        elif cb is not None:
            _lib.check(lib.xrd_gs_render_fwd2(
                C.byref(cam), _lib.ptr(ranges), _lib.ptr(plist), _lib.ptr(xy),
                _lib.ptr(ca), _lib.ptr(cb), _lib.ptr(conic_o),
                _lib.ptr(depths), _lib.ptr(color_a), _lib.ptr(color_b),
                _lib.ptr(depth), _lib.ptr(final_T), _lib.ptr(n_contrib), st),
                'xrd_gs_render_fwd2')
        else:
            _lib.check(lib.xrd_gs_render_fwd(
                C.byref(cam), _lib.ptr(ranges), _lib.ptr(plist), _lib.ptr(xy),
                _lib.ptr(ca), _lib.ptr(conic_o), _lib.ptr(depths),
                _lib.ptr(color_a), _lib.ptr(depth), _lib.ptr(final_T),
                _lib.ptr(n_contrib), st), 'xrd_gs_render_fwd')
    return color_a, color_b, depth, final_T, n_contrib, ckpt


def _blend_bwd(lib, cam, dev, st, n, ranges, plist, xy, conic_o, ca, cb,
               final_T, n_contrib, color_a, color_b, ga, gb, ckpt, bins):
    """-> d_mean2D [n,2], d_conic [n,3], d_op [n,1], d_ca [n,3], d_cb"""
    f = dict(dtype=torch.float32, device=dev)
    with _Timed('gs_render_bwd'):
        if ckpt is not None:
            cap, key_pos, offsets = bins
            d_mean2D = torch.empty(n, 2, **f)
            d_conic = torch.empty(n, 3, **f)
            d_op = torch.empty(n, 1, **f)
            d_ca = torch.empty(n, 3, **f)
            d_cb = torch.empty(n, 3, **f) if cb is not None else None
            key_grad = torch.empty(cap, 12, **f)
            _lib.check(lib.xrd_gs_blend_bwd(
                C.byref(cam), n, cap, _lib.ptr(ranges), _lib.ptr(plist),
                _lib.ptr(key_pos), _lib.ptr(offsets), _lib.ptr(xy),
                _lib.ptr(conic_o), _lib.ptr(ca), _lib.ptr(cb),
                _lib.ptr(final_T), _lib.ptr(n_contrib), _lib.ptr(color_a),
                _lib.ptr(color_b), _lib.ptr(ga), _lib.ptr(gb),
                _lib.ptr(ckpt), _lib.ptr(key_grad), _lib.ptr(d_mean2D),
                _lib.ptr(d_conic), _lib.ptr(d_op), _lib.ptr(d_ca),
                _lib.ptr(d_cb), st), 'xrd_gs_blend_bwd')
            return d_mean2D, d_conic, d_op, d_ca, d_cb
        d_mean2D = torch.zeros(n, 2, **f)
        d_conic = torch.zeros(n, 3, **f)
        d_op = torch.zeros(n, 1, **f)
        d_ca = torch.zeros(n, 3, **f)
        if cb is not None:
            d_cb = torch.zeros(n, 3, **f)
            _lib.check(lib.xrd_gs_render_bwd2(
                C.byref(cam), _lib.ptr(ranges), _lib.ptr(plist), _lib.ptr(xy),
                _lib.ptr(conic_o), _lib.ptr(ca), _lib.ptr(cb),
                _lib.ptr(final_T), _lib.ptr(n_contrib), _lib.ptr(ga),
                _lib.ptr(gb), _lib.ptr(d_mean2D), _lib.ptr(d_conic),
                _lib.ptr(d_op), _lib.ptr(d_ca), _lib.ptr(d_cb), st),
                'xrd_gs_render_bwd2')
            return d_mean2D, d_conic, d_op, d_ca, d_cb
        _lib.check(lib.xrd_gs_render_bwd(
            C.byref(cam), _lib.ptr(ranges), _lib.ptr(plist), _lib.ptr(xy),
            _lib.ptr(conic_o), _lib.ptr(ca), _lib.ptr(final_T),
            _lib.ptr(n_contrib), _lib.ptr(ga), _lib.ptr(d_mean2D),
            _lib.ptr(d_conic), _lib.ptr(d_op), _lib.ptr(d_ca), st),
            'xrd_gs_render_bwd')
        return d_mean2D, d_conic, d_op, d_ca, None


def _geometry_backward(lib, cam, dev, st, n, m3, sc, rt, radii, d_mean2D,
                       d_conic, want_means2D=True):
    f = dict(dtype=torch.float32, device=dev)
    d_means = torch.empty(n, 3, **f)
    d_scales = torch.empty(n, 3, **f)
    d_rots = torch.empty(n, 4, **f)
    _lib.check(lib.xrd_gs_preprocess_bwd(
        C.byref(cam), n, _lib.ptr(m3), _lib.ptr(sc), _lib.ptr(rt),
        _lib.ptr(radii), _lib.ptr(d_mean2D), _lib.ptr(d_conic),
        _lib.ptr(d_means), _lib.ptr(d_scales), _lib.ptr(d_rots), st),
        'xrd_gs_preprocess_bwd')
    d_means2D = torch.cat([d_mean2D, torch.zeros(n, 1, **f)], 1) \
        if want_means2D else None
    return d_means, d_means2D, d_scales, d_rots


class _RasterizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, opacities, colors, scales, rotations,
                rs):
        lib = _lib.lib()
        dev = means3D.device
        st = _lib.stream_ptr(dev)
        cam = _camera(rs)
        H, W = cam.image_height, cam.image_width
        n = means3D.shape[0]
        m3 = means3D.detach().float().contiguous()
        sc = scales.detach().float().contiguous()
        rt = rotations.detach().float().contiguous()
        op = opacities.detach().float().contiguous()
        cl = colors.detach().float().contiguous()
        f = dict(dtype=torch.float32, device=dev)
        i = dict(dtype=torch.int32, device=dev)
        depths, xy, conic_o, radii, ranges, plist, n_keys, bins = _geometry(
            lib, cam, dev, st, m3, sc, rt, op)
        color, _, depth, final_T, n_contrib, ckpt = _blend_fwd(
            lib, cam, dev, st, H, W, ranges, plist, xy, cl, None, conic_o,
            depths, bins[0])
        if PROFILE is not None:
            PROFILE.setdefault('pairs', []).append(n_contrib.sum())
            PROFILE.setdefault('keys', []).append(n_keys.clone())
            PROFILE.setdefault('gaussians', []).append(n)
        ctx.rs, ctx.n, ctx.cap = rs, n, bins[0]
        ctx.has_ckpt = ckpt is not None
        extra = [ckpt, bins[1], bins[2], color] if ckpt is not None else []
        ctx.save_for_backward(m3, sc, rt, cl, xy, conic_o, radii, ranges,
                              plist, final_T, n_contrib, *extra)
        ctx.mark_non_differentiable(radii, depth)
        # gradients of the non-differentiable outputs arrive as None instead
        # of materialised zero tensors (one fill launch each)
        ctx.set_materialize_grads(False)
        return color, radii, depth

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth):
        lib = _lib.lib()
        (m3, sc, rt, cl, xy, conic_o, radii, ranges, plist, final_T,
         n_contrib) = ctx.saved_tensors[:11]
        ckpt = key_pos = offsets = color = None
        if ctx.has_ckpt:
            ckpt, key_pos, offsets, color = ctx.saved_tensors[11:]
        dev = m3.device
        st = _lib.stream_ptr(dev)
        cam = _camera(ctx.rs)
        n = ctx.n
        gc = g_color.float().contiguous() if g_color is not None else \
            torch.zeros(3, cam.image_height, cam.image_width,
                        dtype=torch.float32, device=dev)
        d_mean2D, d_conic, d_op, d_col, _ = _blend_bwd(
            lib, cam, dev, st, n, ranges, plist, xy, conic_o, cl, None,
            final_T, n_contrib, color, None, gc, None, ckpt,
            (ctx.cap, key_pos, offsets))
        d_means, d_means2D, d_scales, d_rots = _geometry_backward(
            lib, cam, dev, st, n, m3, sc, rt, radii, d_mean2D, d_conic,
            ctx.needs_input_grad[1])
        return d_means, d_means2D, d_op, d_col, d_scales, d_rots, None


class _RasterizeDualFn(torch.autograd.Function):
    """two colour sets over the same Gaussians in ONE pass each way
    (xrd_gs_render_fwd2 / _bwd2): (color_a, radii, depth, color_b).  What
    SplaTAM's two GaussianRasterizer calls per view compute
    (gaussian_cloud_splatam.py:63-69), with one preprocess, one binning and
    one blend instead of two."""

    @staticmethod
    def forward(ctx, means3D, means2D, opacities, colors_a, colors_b, scales,
                rotations, rs):
        lib = _lib.lib()
        dev = means3D.device
        st = _lib.stream_ptr(dev)
        cam = _camera(rs)
        H, W = cam.image_height, cam.image_width
        n = means3D.shape[0]
        m3 = means3D.detach().float().contiguous()
        sc = scales.detach().float().contiguous()
        rt = rotations.detach().float().contiguous()
        op = opacities.detach().float().contiguous()
        ca = colors_a.detach().float().contiguous()
        cb = colors_b.detach().float().contiguous()
        f = dict(dtype=torch.float32, device=dev)
        i = dict(dtype=torch.int32, device=dev)
        depths, xy, conic_o, radii, ranges, plist, n_keys, bins = _geometry(
            lib, cam, dev, st, m3, sc, rt, op)
        color_a, color_b, depth, final_T, n_contrib, ckpt = _blend_fwd(
            lib, cam, dev, st, H, W, ranges, plist, xy, ca, cb, conic_o,
            depths, bins[0])
        if PROFILE is not None:
            PROFILE.setdefault('pairs', []).append(n_contrib.sum())
            PROFILE.setdefault('keys', []).append(n_keys.clone())
            PROFILE.setdefault('gaussians', []).append(n)
        ctx.rs, ctx.n, ctx.cap = rs, n, bins[0]
        ctx.has_ckpt = ckpt is not None
        extra = [ckpt, bins[1], bins[2], color_a, color_b] \
            if ckpt is not None else []
        ctx.save_for_backward(m3, sc, rt, ca, cb, xy, conic_o, radii, ranges,
                              plist, final_T, n_contrib, *extra)
        ctx.mark_non_differentiable(radii, depth)
        # gradients of the non-differentiable outputs arrive as None instead
        # of materialised zero tensors (one fill launch each)
        ctx.set_materialize_grads(False)
        return color_a, radii, depth, color_b

    @staticmethod
    def backward(ctx, g_a, g_radii, g_depth, g_b):
        lib = _lib.lib()
        (m3, sc, rt, ca, cb, xy, conic_o, radii, ranges, plist, final_T,
         n_contrib) = ctx.saved_tensors[:12]
        ckpt = key_pos = offsets = color_a = color_b = None
        if ctx.has_ckpt:
            ckpt, key_pos, offsets, color_a, color_b = ctx.saved_tensors[12:]
        dev = m3.device
        st = _lib.stream_ptr(dev)
        cam = _camera(ctx.rs)
        n = ctx.n
        f = dict(dtype=torch.float32, device=dev)
        H, W = cam.image_height, cam.image_width
        ga = g_a.float().contiguous() if g_a is not None \
            else torch.zeros(3, H, W, **f)
        gb = g_b.float().contiguous() if g_b is not None \
            else torch.zeros(3, H, W, **f)
        d_mean2D, d_conic, d_op, d_ca, d_cb = _blend_bwd(
            lib, cam, dev, st, n, ranges, plist, xy, conic_o, ca, cb, final_T,
            n_contrib, color_a, color_b, ga, gb, ckpt,
            (ctx.cap, key_pos, offsets))
        d_means, d_means2D, d_scales, d_rots = _geometry_backward(
            lib, cam, dev, st, n, m3, sc, rt, radii, d_mean2D, d_conic,
            ctx.needs_input_grad[1])
        return d_means, d_means2D, d_op, d_ca, d_cb, d_scales, d_rots, None


def rasterize_dual(raster_settings, means3D, means2D, opacities, colors_a,
                   colors_b, scales, rotations):
    """-> (color_a [3,H,W], radii [N], depth [1,H,W], color_b [3,H,W]); equal
    to GaussianRasterizer(raster_settings) called once with colors_a and once
    with colors_b, the ``means2D`` gradient being the sum of both calls'"""
    if not means3D.is_cuda:
        raise _lib.XrdError('rasterize_dual needs CUDA tensors')
    return _RasterizeDualFn.apply(means3D, means2D, opacities, colors_a,
                                  colors_b, scales, rotations, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None,
                colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if shs is not None or colors_precomp is None:
            raise NotImplementedError('only colors_precomp is built (the '
                                      'reference uses sh_degree=0)')
        if cov3D_precomp is not None or scales is None or rotations is None:
            raise NotImplementedError('only the scales/rotations path is '
                                      'built')
        if not means3D.is_cuda:
            raise _lib.XrdError('GaussianRasterizer needs CUDA tensors')
        return _RasterizeFn.apply(means3D, means2D, opacities, colors_precomp,
                                  scales, rotations, self.raster_settings)
