"""Point-SLAM geometry path on the fused kernels (csrc/nice_render.hip:
point_geo_fwd / _bwd): neighbour interpolation of the geometric features +
the geometry decoder behind autograd.

    occ, has = geometry(decoder, p, neighbours, npc, radius)

replaces ``MLP_geometry._interpolate`` + the trunk
(slam/model_components/decoder_pointslam.py:162-273).  The decoder has the
structure of NICE-SLAM's ``MLP`` (5 x 32 ReLU, feature added after every layer,
skip after the third, Fourier features) — it is packed with the NICE 'middle'
layout, ``embedder._B`` scaled by 2 pi (its embedding is sin(2 pi p B))."""
from __future__ import annotations

import math

import torch

from .. import _lib
from . import nice as _nice

PROFILE = None   # bench.py: key -> [(start event, end event, points)]


def supported(dec) -> bool:
    try:
        return (dec.weighting == 'distance' and dec.c_dim == 32 and
                list(dec.skips) == [2] and len(dec.pts_linears) == 5 and
                dec.pts_linears[0].weight.shape == (32, 93) and
                dec.output_linear.weight.shape == (1, 32) and
                dec.embedder.mapping_size == 93 and not dec.embedder.concat)
    except AttributeError:
        return False


def pack(dec, device) -> torch.Tensor:
    """the geometry decoder in the kernels' fragment layout; cached until a
    parameter changes (the decoder is fixed in Point-SLAM's mapping)"""
    sd = {k: v for k, v in dec.state_dict().items()}
    key = tuple((k, v._version, v.data_ptr()) for k, v in sd.items()
                if k.startswith(('fc_c', 'pts_linears', 'output_linear')) or
                k == 'embedder._B')
    hit = getattr(dec, '_xrd_pack', None)
    if hit is not None and hit[0] == key and hit[1].device == \
            torch.device(device):
        return hit[1]
    sd = dict(sd)
    B = dec.embedder._B
    sd['embedder._B'] = (2 * math.pi) * B.detach().to(device)
    flat = _nice.flatten_state_dict(
        {k: v.detach().to(device) for k, v in sd.items()}, 'middle')
    packed = _nice.pack_decoder(flat, 'middle')
    dec._xrd_pack = (key, packed)
    return packed


class _GeoFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, feats, nbr, n_nb, cloud, fmask, radius, radius_all,
                min_nn, empty, packed):
        lib = _lib.lib()
        dev = p.device
        p = p.detach().float().reshape(-1, 3).contiguous()
        n = p.shape[0]
        f = feats.detach().float().contiguous()
        occ = torch.empty(n, dtype=torch.float32, device=dev)
        has = torch.empty(n, dtype=torch.uint8, device=dev)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        masks = torch.empty(n, 4, dtype=torch.int64, device=dev) if need \
            else None
        _lib.check(lib.xrd_point_geo_fwd(
            n, _lib.ptr(p), _lib.ptr(nbr), _lib.ptr(n_nb), _lib.ptr(cloud),
            _lib.ptr(f), _lib.ptr(fmask), _lib.ptr(radius), float(radius_all),
            int(min_nn), _lib.ptr(empty), _lib.ptr(packed), _lib.ptr(occ),
            _lib.ptr(has), _lib.ptr(masks), _lib.stream_ptr(dev)),
            'xrd_point_geo_fwd')
        ctx.args = (float(radius_all), int(min_nn))
        ctx.save_for_backward(p, f, nbr, n_nb, cloud, fmask, radius, empty,
                              packed, masks)
        ctx.mark_non_differentiable(has)
        # gradients of the non-differentiable outputs arrive as None instead
        # of materialised zero tensors (one fill launch each)
        ctx.set_materialize_grads(False)
        return occ, has

    @staticmethod
    def backward(ctx, g_occ, _g_has):
        lib = _lib.lib()
        p, f, nbr, n_nb, cloud, fmask, radius, empty, packed, masks = \
            ctx.saved_tensors
        radius_all, min_nn = ctx.args
        dev = p.device
        need_p, need_f = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_p = torch.empty_like(p) if need_p else None
        g_f = torch.zeros_like(f) if need_f else None
        _lib.check(lib.xrd_point_geo_bwd(
            p.shape[0], _lib.ptr(p), _lib.ptr(nbr), _lib.ptr(n_nb),
            _lib.ptr(cloud), _lib.ptr(f), _lib.ptr(fmask), _lib.ptr(radius),
            radius_all, min_nn, _lib.ptr(empty), _lib.ptr(packed),
            _lib.ptr(masks), _lib.ptr(g_occ.float().contiguous()),
            _lib.ptr(g_p), _lib.ptr(g_f), _lib.stream_ptr(dev)),
            'xrd_point_geo_bwd')
        return (g_p, g_f) + (None, ) * 9


def geometry(dec, p, neighbors, npc, dynamic_r_query):
    """-> (occupancy logit [n], has_neighbours [n] bool).  ``neighbors`` =
    (D, I, n_nb) of ``npc.find_neighbors_faiss`` for these points."""
    dev = p.device
    _, ids, n_nb = neighbors
    cloud = npc.cloud_tensor(dev).float().contiguous()
    fmask = getattr(npc, 'frustum_mask', None)
    fmask = None if fmask is None else \
        fmask.detach().reshape(-1).to(torch.uint8).contiguous()
    radius = None
    if dec.use_dynamic_radius and dynamic_r_query is not None:
        radius = dynamic_r_query.detach().float().reshape(-1).contiguous()
    empty = dec.empty_feature_fn(dec.c_dim, dev).float().contiguous()
    # tracking optimises the pose only: no gradient to the map is computed
    feats = npc.geo_feats if getattr(dec, 'map_gradients', True) \
        else npc.geo_feats.detach()
    occ, has = _GeoFn.apply(
        p.reshape(-1, 3), feats, ids.long().contiguous(),
        n_nb.int().contiguous(), cloud, fmask, radius,
        float(npc.get_radius_query()), dec.min_nn_num, empty,
        pack(dec, dev))
    return occ, has.bool()


# ---- colour path (csrc/point_color.hip) ------------------------------------------
def color_supported(dec) -> bool:
    try:
        return (dec.weighting == 'distance' and dec.c_dim == 32 and
                list(dec.skips) == [2] and len(dec.pts_linears) == 5 and
                dec.encode_rel_pos_in_col and not dec.use_view_direction and
                not dec.encode_exposure and
                isinstance(dec.actvn, torch.nn.Softplus) and
                dec.actvn.beta == 100 and dec.actvn.threshold == 20 and
                dec.pts_linears[0].weight.shape == (128, 40) and
                dec.pts_linears[3].weight.shape == (128, 168) and
                dec.output_linear.weight.shape == (3, 128) and
                dec.mlp_col_neighbor.linear1.weight.shape == (128, 52) and
                dec.embedder.mapping_size == 20 and dec.embedder.concat and
                dec.embedder_rel_pos.mapping_size == 10)
    except AttributeError:
        return False


def color_params(dec):
    """the trainable tensors in the kernels' flat order (= the order of
    ``dec.parameters()``: Adam steps the re-seated decoder with one launch)"""
    f = dec.mlp_col_neighbor
    out = [dec.embedder_rel_pos._B, f.linear1.weight, f.linear1.bias,
           f.linear2.weight, f.linear2.bias]
    for layer in dec.fc_c:
        out += [layer.weight, layer.bias]
    for layer in dec.pts_linears:
        out += [layer.weight, layer.bias]
    return out + [dec.output_linear.weight, dec.output_linear.bias]


def color_flat(dec, device):
    """the decoder as ONE flat buffer [parameters | embedder._B]: the
    parameters are re-seated as back-to-back views of it (values kept,
    idempotent), so packing is one gather of current values and the flat
    gradient of the kernels maps onto the parameters as views"""
    params = color_params(dec)
    p0 = params[0]
    flat = getattr(dec, '_xrd_flat', None)
    ok = flat is not None and flat.device == torch.device(device)
    nxt = flat.data_ptr() if ok else 0
    for p in params:
        ok = ok and p.data_ptr() == nxt and p.is_contiguous()
        nxt += 4 * p.numel()
    if ok:
        return flat
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1).float().to(device)
                          for p in params] +
                         [dec.embedder._B.detach().reshape(-1).float()
                          .to(device)])
        off = 0
        for p in params:
            p.data = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
    dec._xrd_flat = flat
    return flat


_COLOR_INDEX = {}


def _color_index(device):
    dev = torch.device(device)
    hit = _COLOR_INDEX.get(dev)
    if hit is None:
        lib = _lib.lib()
        idx = torch.empty(lib.xrd_point_color_pack_len(), dtype=torch.int32)
        _lib.check(lib.xrd_point_color_pack_index(_lib.ptr(idx)),
                   'xrd_point_color_pack_index')
        live = (idx >= 0).to(dev)
        hit = (idx.clamp(min=0).long().to(dev), live)
        _COLOR_INDEX[dev] = hit
    return hit


def pack_color(flat: torch.Tensor) -> torch.Tensor:
    idx, live = _color_index(flat.device)
    return torch.where(live, flat.detach()[idx], flat.new_zeros(()))


_COLOR_SCRATCH = {}


def _color_scratch(device, n):
    """(operands of the weight gradients, per-block partial products): grown,
    never shrunk, one pair per device (consumed on the stream that fills it)"""
    lib = _lib.lib()
    dev = torch.device(device)
    ops, ws = _COLOR_SCRATCH.get(dev, (None, None))
    need = lib.xrd_point_color_ops_floats(n)
    if ops is None or ops.numel() < need:
        ops = torch.empty(need, dtype=torch.float32, device=dev)
    if ws is None:
        ws = torch.empty(lib.xrd_point_color_ws_floats(), dtype=torch.float32,
                         device=dev)
    _COLOR_SCRATCH[dev] = (ops, ws)
    return ops, ws


class _ColFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, feats, flat, nbr, n_nb, cloud, radius, radius_all,
                min_nn, empty, *params):
        # ``params``: the tensors ``flat`` is the storage of (color_flat),
        # passed so that autograd routes the flat gradient to them
        lib = _lib.lib()
        dev = p.device
        p = p.detach().float().reshape(-1, 3).contiguous()
        n = p.shape[0]
        f = feats.detach().float().contiguous()
        packed = pack_color(flat)
        ctx.need_w = any(ctx.needs_input_grad[10:])
        need = any(ctx.needs_input_grad[:2]) or ctx.need_w
        rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
        save_c = save_h = save_y = None
        if need:
            save_c = torch.empty(n, 32, dtype=torch.float32, device=dev)
            save_h = torch.empty(5, n, 128, dtype=torch.float32, device=dev)
            save_y = torch.empty(n, 8, 32, dtype=torch.float32, device=dev)
        if PROFILE is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.xrd_point_color_fwd(
            n, _lib.ptr(p), _lib.ptr(nbr), _lib.ptr(n_nb), _lib.ptr(cloud),
            _lib.ptr(f), _lib.ptr(radius), float(radius_all), int(min_nn),
            _lib.ptr(empty), _lib.ptr(packed), _lib.ptr(rgb),
            _lib.ptr(save_c), _lib.ptr(save_h), _lib.ptr(save_y),
            _lib.stream_ptr(dev)), 'xrd_point_color_fwd')
        if PROFILE is not None:
            e1.record()
            PROFILE.setdefault('color_fwd', []).append((e0, e1, n))
        ctx.args = (float(radius_all), int(min_nn), flat.numel(),
                    [t.shape for t in params])
        ctx.save_for_backward(p, f, nbr, n_nb, cloud, radius, packed, rgb,
                              save_c, save_h, save_y)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        lib = _lib.lib()
        p, f, nbr, n_nb, cloud, radius, packed, rgb, save_c, save_h, save_y \
            = ctx.saved_tensors
        radius_all, min_nn, flat_len, shapes = ctx.args
        dev = p.device
        need_p, need_f = ctx.needs_input_grad[:2]
        need_w = ctx.need_w
        n = p.shape[0]
        g_p = torch.empty_like(p) if need_p else None
        g_f = torch.zeros_like(f) if need_f else None
        g_flat = ops = ws = None
        if need_w:
            g_flat = torch.empty(lib.xrd_point_color_grad_len(),
                                 dtype=torch.float32, device=dev)
            ops, ws = _color_scratch(dev, n)
        if PROFILE is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.xrd_point_color_bwd(
            n, _lib.ptr(p), _lib.ptr(nbr), _lib.ptr(n_nb), _lib.ptr(cloud),
            _lib.ptr(f), _lib.ptr(radius), radius_all, min_nn,
            _lib.ptr(packed), _lib.ptr(rgb), _lib.ptr(save_c),
            _lib.ptr(save_h), _lib.ptr(save_y),
            _lib.ptr(g_rgb.float().contiguous()), _lib.ptr(g_p),
            _lib.ptr(g_f), _lib.ptr(g_flat), _lib.ptr(ops), _lib.ptr(ws),
            _lib.stream_ptr(dev)), 'xrd_point_color_bwd')
        if PROFILE is not None:
            e1.record()
            PROFILE.setdefault('color_bwd_w' if need_w else 'color_bwd',
                               []).append((e0, e1, n))
        g_params = [None] * len(shapes)
        if need_w:
            off = 0
            for k, shp in enumerate(shapes):
                g_params[k] = g_flat[off:off + shp.numel()].view(shp)
                off += shp.numel()
        return (g_p, g_f) + (None, ) * 8 + tuple(g_params)


def color(dec, p, neighbors, npc, dynamic_r_query):
    """-> rgb [n,3].  ``neighbors`` = (D, I, n_nb) of the search for ``p``."""
    dev = p.device
    _, ids, n_nb = neighbors
    cloud = npc.cloud_tensor(dev).float().contiguous()
    radius = None
    if dec.use_dynamic_radius and dynamic_r_query is not None:
        radius = dynamic_r_query.detach().float().reshape(-1).contiguous()
    empty = dec.empty_feature_fn(dec.c_dim, dev).float().contiguous()
    feats, params = npc.col_feats, color_params(dec)
    if not getattr(dec, 'map_gradients', True):
        # tracking optimises the pose only: neither the feature scatter nor
        # the weight products are computed
        feats, params = feats.detach(), [t.detach() for t in params]
    return _ColFn.apply(p.reshape(-1, 3), feats, color_flat(dec, dev),
                        ids.long().contiguous(), n_nb.int().contiguous(),
                        cloud, radius, float(npc.get_radius_query()),
                        dec.min_nn_num, empty, *params)


# ---- mapping loss (csrc/point_loss.hip) -------------------------------------------
class _MapLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z_vals, target_d, target_rgb, point_mask, ray_valid,
                coef, w_color, min_valid):
        lib = _lib.lib()
        dev = raw.device
        n, S = z_vals.shape
        r = raw.detach().float().contiguous()
        assert r.shape == (n * S, 4)
        z = z_vals.detach().float().contiguous()
        td = target_d.detach().float().reshape(-1).contiguous()
        tc = None if target_rgb is None else \
            target_rgb.detach().float().contiguous()
        pm = point_mask.reshape(-1).to(torch.uint8).contiguous()
        rv = None if ray_valid is None else \
            ray_valid.reshape(-1).to(torch.uint8).contiguous()
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        g_raw = torch.empty_like(r)
        _lib.check(lib.xrd_point_map_loss(
            n, S, _lib.ptr(r), _lib.ptr(pm), _lib.ptr(z), _lib.ptr(td),
            _lib.ptr(tc), _lib.ptr(rv), float(coef), float(w_color),
            int(min_valid), _lib.ptr(loss), _lib.ptr(g_raw),
            _lib.stream_ptr(dev)), 'xrd_point_map_loss')
        ctx.save_for_backward(g_raw)
        return loss[0] + loss[1] if tc is not None else loss[0] + 0.0

    @staticmethod
    def backward(ctx, g):
        g_raw, = ctx.saved_tensors
        return (g_raw * g, ) + (None, ) * 8


class _TrackLossFn(torch.autograd.Function):
    """(depth [n], var [n], colour [n,3]) -> (geo, rgb) of ConvOnet2's
    tracking loss on a shape-preserving batch (xrd_point_track_loss)"""

    @staticmethod
    def forward(ctx, depth, var, color, target_d, target_rgb, ray_valid,
                handle_dynamic, use_color, w_color):
        lib = _lib.lib()
        dev = depth.device
        n = depth.shape[0]
        d = depth.detach().float().contiguous()
        v = var.detach().float().contiguous()
        c = color.detach().float().contiguous() if use_color else None
        td = target_d.detach().float().reshape(-1).contiguous()
        tc = target_rgb.detach().float().contiguous() if use_color else None
        rv = None if ray_valid is None else \
            ray_valid.reshape(-1).to(torch.uint8).contiguous()
        f = dict(dtype=torch.float32, device=dev)
        loss = torch.empty(2, **f)
        g_d = torch.empty(n, **f)
        g_c = torch.empty(n, 3, **f) if use_color else None
        _lib.check(lib.xrd_point_track_loss(
            n, int(bool(handle_dynamic)), int(bool(use_color)),
            float(w_color), _lib.ptr(d), _lib.ptr(v), _lib.ptr(c),
            _lib.ptr(td), _lib.ptr(tc), _lib.ptr(rv), _lib.ptr(loss),
            _lib.ptr(g_d), _lib.ptr(g_c), _lib.stream_ptr(dev)),
            'xrd_point_track_loss')
        ctx.save_for_backward(g_d, g_c)
        ctx.use_color = bool(use_color)
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, g_geo, g_rgb):
        g_d, g_c = ctx.saved_tensors
        # the kernel's gradients are those of geo + rgb; both upstream
        # factors are the same scalar in the optimisation loop (a plain sum)
        gd = g_d * g_geo
        gc = g_c * g_rgb if ctx.use_color and g_c is not None else None
        return gd, None, gc, None, None, None, None, None, None


def track_loss(depth, var, color, target_d, target_rgb, ray_valid,
               handle_dynamic, use_color, w_color):
    return _TrackLossFn.apply(depth, var, color, target_d, target_rgb,
                              ray_valid, handle_dynamic, use_color, w_color)


class _BatchFn(torch.autograd.Function):
    """batch filter + sample placement of a Point-SLAM iteration
    (xrd_point_batch): (rays_o, rays_d) -> (keep, radius, z_vals, pts,
    radius_pts); differentiable w.r.t. the rays through pts = o + dir z"""

    @staticmethod
    def forward(ctx, rays_o, rays_d, target_d, radius_stack, idx, geom, S,
                near, far):
        lib = _lib.lib()
        dev = rays_o.device
        ro = rays_o.detach().float().contiguous()
        rd = rays_d.detach().float().contiguous()
        td = target_d.detach().float().reshape(-1).contiguous()
        n = ro.shape[0]
        n_per, wcrop, hedge, wedge, width, hw = geom
        f = dict(dtype=torch.float32, device=dev)
        keep = torch.empty(n, dtype=torch.uint8, device=dev)
        z = torch.empty(n, S, **f)
        pts = torch.empty(n * S, 3, **f)
        rq = rq_pts = None
        if radius_stack is not None:
            rq = torch.empty(n, **f)
            rq_pts = torch.empty(n * S, 1, **f)
            assert idx.dtype == torch.int64 and idx.numel() == n
        _lib.check(lib.xrd_point_batch(
            n, int(S), _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(td),
            _lib.ptr(radius_stack), _lib.ptr(idx), int(n_per), int(wcrop),
            int(hedge), int(wedge), int(width), int(hw), float(near),
            float(far), _lib.ptr(keep), _lib.ptr(rq), _lib.ptr(z),
            _lib.ptr(pts), _lib.ptr(rq_pts), None, _lib.stream_ptr(dev)),
            'xrd_point_batch')
        ctx.save_for_backward(z)
        ctx.set_materialize_grads(False)
        keep = keep.view(torch.bool)
        outs = (keep, z, pts) if rq is None else (keep, z, pts, rq, rq_pts)
        ctx.mark_non_differentiable(*[o for o in outs if o is not pts])
        return outs

    @staticmethod
    def backward(ctx, *grads):
        g_pts = grads[2]
        if g_pts is None:
            return (None, ) * 9
        z, = ctx.saved_tensors
        n, S = z.shape
        g = g_pts.reshape(n, S, 3)
        return (g.sum(1), (g * z[..., None]).sum(1)) + (None, ) * 7


def batch(rays_o, rays_d, target_d, radius_stack, idx, geom, S, near, far):
    """-> dict(ray_valid, z_vals, pts [, batch_dynamic_r, rq_pts])"""
    out = _BatchFn.apply(rays_o, rays_d, target_d, radius_stack, idx, geom,
                         S, near, far)
    res = {'ray_valid': out[0], 'z_vals': out[1], 'pts': out[2]}
    if len(out) == 5:
        res['batch_dynamic_r'], res['rq_pts'] = out[3], out[4]
    return res


def map_loss(raw, z_vals, target_d, target_rgb, point_mask, ray_valid, coef,
             w_color, min_valid):
    """compositing + Point-SLAM's mapping loss (sum |d - depth| + w_color sum
    |rgb - colour| over the rays that count), gradient w.r.t. ``raw``"""
    return _MapLossFn.apply(raw, z_vals, target_d, target_rgb, point_mask,
                            ray_valid, coef, w_color, min_valid)


# ---- compositing (csrc/point_loss.hip) --------------------------------------------
class _CompositeFn(torch.autograd.Function):
    """raw [m,4] = [rgb, occupancy logit] rows, z_vals [n,S], point_mask [m]
    -> depth [n], var [n], colour [n,3] (raw2outputs_nerf_color2 with the
    no-neighbour override)"""

    @staticmethod
    def forward(ctx, raw, z_vals, point_mask, coef):
        lib = _lib.lib()
        dev = raw.device
        n, S = z_vals.shape
        r = raw.detach().float().contiguous()
        assert r.shape == (n * S, 4)
        z = z_vals.detach().float().contiguous()
        pm = point_mask.reshape(-1).to(torch.uint8).contiguous()
        depth = torch.empty(n, dtype=torch.float32, device=dev)
        var = torch.empty(n, dtype=torch.float32, device=dev)
        color = torch.empty(n, 3, dtype=torch.float32, device=dev)
        _lib.check(lib.xrd_point_composite_fwd(
            n, S, _lib.ptr(r), 4, r.data_ptr() + 12, 4, _lib.ptr(pm),
            _lib.ptr(z), float(coef), _lib.ptr(depth), _lib.ptr(var),
            _lib.ptr(color), _lib.stream_ptr(dev)), 'xrd_point_composite_fwd')
        ctx.coef = float(coef)
        ctx.save_for_backward(r, z, pm)
        return depth, var, color

    @staticmethod
    def backward(ctx, g_depth, g_var, g_color):
        lib = _lib.lib()
        r, z, pm = ctx.saved_tensors
        n, S = z.shape
        g_raw = torch.empty_like(r)

        def c(t):
            return None if t is None else t.float().contiguous()
        gd, gv, gc = c(g_depth), c(g_var), c(g_color)
        _lib.check(lib.xrd_point_composite_bwd(
            n, S, _lib.ptr(r), 4, r.data_ptr() + 12, 4, _lib.ptr(pm),
            _lib.ptr(z), ctx.coef, _lib.ptr(gd), _lib.ptr(gv), _lib.ptr(gc),
            _lib.ptr(g_raw), 4, g_raw.data_ptr() + 12, 4,
            _lib.stream_ptr(r.device)), 'xrd_point_composite_bwd')
        return g_raw, None, None, None


def composite(raw, z_vals, point_mask, coef):
    return _CompositeFn.apply(raw, z_vals, point_mask, coef)
