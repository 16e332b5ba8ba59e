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

PROFILE = None   # bench.py: key -> [(start, end)]


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
    occ, has = _GeoFn.apply(
        p.reshape(-1, 3), npc.geo_feats, ids.long().contiguous(),
        n_nb.int().contiguous(), cloud, fmask, radius,
        float(npc.get_radius_query()), dec.min_nn_num, empty,
        pack(dec, dev))
    return occ, has.bool()
