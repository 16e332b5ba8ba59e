"""Vox-Fusion fused "voxel features + decoder" (csrc/vox_render.hip) behind
autograd.

    sdf, rgb = points(decoder, xyz, voxel_idx, map_states, voxel_size)

replaces ``Decoder(get_features(samples, map_states, voxel_size))`` of the
reference (slam/models/sparse_voxel.py:230-238,
voxel_helpers_voxfusion.py:109-123, decoder_voxfusion.py:123-149) for the
model's default decoder (in_dim 16, width 128, depth 2, no positional
encoding).  PyTorch is device memory + autograd plumbing; the decoder's weight
gradients are contracted over the points by xrd_vox_dw (csrc/vox_dw.hip) on
the operands the forward / backward kernels leave in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib

_idx_cache = {}
# bench.py: per-launch HIP-event timing (key -> [(start, end) events])
PROFILE = None


class _Timed:
    def __init__(self, key):
        self.key = key if PROFILE is not None and \
            not torch.cuda.is_current_stream_capturing() else None

    def __enter__(self):
        if self.key is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.key is not None:
            self.e1.record()
            PROFILE.setdefault(self.key, []).append((self.e0, self.e1))


_PARAM_SHAPES = [(128, 16), (128, ), (128, 128), (128, ), (129, 128), (129, ),
                 (128, 144), (128, ), (3, 128), (3, )]
_PARAM_SIZES = [int(np.prod(s_)) for s_ in _PARAM_SHAPES]
_dw_ws = {}


def dw_workspace(device) -> torch.Tensor:
    """per-block partials of xrd_vox_dw (one buffer per device; the kernel
    pair that uses it runs on one stream)"""
    key = str(device)
    if key not in _dw_ws:
        _dw_ws[key] = torch.empty(_lib.lib().xrd_vox_dw_ws_floats(),
                                  dtype=torch.float32, device=device)
    return _dw_ws[key]


def pack_index(device) -> torch.Tensor:
    key = str(device)
    if key not in _idx_cache:
        lib = _lib.lib()
        idx = np.empty(lib.xrd_vox_pack_len(), dtype=np.int32)
        _lib.check(lib.xrd_vox_pack_index(idx.ctypes.data), 'vox_pack_index')
        idx64 = idx.astype(np.int64)
        idx64[idx64 < 0] = lib.xrd_vox_flat_len()  # slot holding 0
        _idx_cache[key] = torch.from_numpy(idx64).to(device)
    return _idx_cache[key]


def decoder_params(decoder):
    """the ten tensors in state_dict order, or None when the decoder is not
    the shape the kernels are built for"""
    try:
        if decoder.D != 2 or decoder.W != 128 or decoder.skips not in ([],
                                                                       [4]):
            return None
        if decoder.pe.embedding_size != 16 or type(decoder.pe).__name__ != \
                '_Identity':
            return None
        ps = [decoder.pts_linears[0].weight, decoder.pts_linears[0].bias,
              decoder.pts_linears[1].weight, decoder.pts_linears[1].bias,
              decoder.sdf_out.weight, decoder.sdf_out.bias,
              decoder.color_out[0].weight, decoder.color_out[0].bias,
              decoder.color_out[2].weight, decoder.color_out[2].bias]
    except AttributeError:
        return None
    shapes = [(128, 16), (128, ), (128, 128), (128, ), (129, 128), (129, ),
              (128, 144), (128, ), (3, 128), (3, )]
    if [tuple(p.shape) for p in ps] != shapes:
        return None
    return ps


class _VoxPointsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, emb, vox_idx, centres, vertex_idx, voxel_size,
                *params):
        lib = _lib.lib()
        dev = xyz.device
        st = _lib.stream_ptr(dev)
        P = xyz.shape[0]
        xyz = xyz.detach().float().contiguous()
        emb_c = emb.detach().float().contiguous()
        vox_idx = vox_idx.int().contiguous()
        centres = centres.float().contiguous()
        vertex_idx = vertex_idx.int().contiguous()
        flat = torch.cat([p.detach().reshape(-1).float() for p in params] +
                         [xyz.new_zeros(1)])
        packed = flat[pack_index(dev)]
        need_w = any(ctx.needs_input_grad[6:])
        need_bwd = need_w or ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        f = dict(dtype=torch.float32, device=dev)
        sdf = torch.empty(P, **f)
        rgb = torch.empty(P, 3, **f)
        sx = torch.empty(P, 16, **f) if need_w else None
        sh1, sh2, sf, shc = ((torch.empty(P, 128, **f) for _ in range(4))
                             if need_w else (None, ) * 4)
        masks = torch.empty(P, 3, 4, dtype=torch.int32, device=dev) \
            if need_bwd else None
        with _Timed(('vox_points_fwd', P, need_w)):
          _lib.check(lib.xrd_vox_points_fwd(
            P, _lib.ptr(xyz), _lib.ptr(vox_idx), _lib.ptr(centres),
            _lib.ptr(vertex_idx), _lib.ptr(emb_c), float(voxel_size),
            _lib.ptr(packed), _lib.ptr(sdf), _lib.ptr(rgb), _lib.ptr(sx),
            _lib.ptr(sh1), _lib.ptr(sh2), _lib.ptr(sf), _lib.ptr(shc),
            _lib.ptr(masks), None, st), 'xrd_vox_points_fwd')
        ctx.voxel_size, ctx.need_w = float(voxel_size), need_w
        ctx.save_for_backward(xyz, emb_c, vox_idx, centres, vertex_idx,
                              packed, rgb, masks, sx, sh1, sh2, sf, shc)
        return sdf, rgb

    @staticmethod
    def backward(ctx, g_sdf, g_rgb):
        lib = _lib.lib()
        (xyz, emb, vox_idx, centres, vertex_idx, packed, rgb, masks, sx, sh1,
         sh2, sf, shc) = ctx.saved_tensors
        dev = xyz.device
        P = xyz.shape[0]
        f = dict(dtype=torch.float32, device=dev)
        need_xyz, need_emb = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_w = ctx.need_w
        g_xyz = torch.empty(P, 3, **f) if need_xyz else None
        g_emb = torch.zeros_like(emb) if need_emb else None
        gc3 = torch.empty(P, 4, **f) if need_w else None
        ghc, gf, gh2, gh1 = ((torch.empty(P, 128, **f) for _ in range(4))
                             if need_w else (None, ) * 4)
        gs = g_sdf.float().contiguous() if g_sdf is not None else None
        gr = g_rgb.float().contiguous() if g_rgb is not None else None
        with _Timed(('vox_points_bwd', P, need_w)):
          _lib.check(lib.xrd_vox_points_bwd(
            P, _lib.ptr(xyz), _lib.ptr(vox_idx), _lib.ptr(centres),
            _lib.ptr(vertex_idx), _lib.ptr(emb), ctx.voxel_size,
            _lib.ptr(packed), _lib.ptr(rgb), _lib.ptr(masks), _lib.ptr(gs),
            _lib.ptr(gr), _lib.ptr(g_xyz), _lib.ptr(g_emb), _lib.ptr(gc3),
            _lib.ptr(ghc), _lib.ptr(gf), _lib.ptr(gh2), _lib.ptr(gh1), None,
            _lib.stream_ptr(dev)), 'xrd_vox_points_bwd')
        gp = [None] * 10
        if need_w:
            # dW = G^T A over the points: xrd_vox_dw (csrc/vox_dw.hip)
            flat = torch.empty(lib.xrd_vox_flat_len(), **f)
            with _Timed(('vox_dw', P, True)):
                _lib.check(lib.xrd_vox_dw(
                    P, None, _lib.ptr(sx), _lib.ptr(sh1), _lib.ptr(sh2),
                    _lib.ptr(sf), _lib.ptr(shc), _lib.ptr(gc3), _lib.ptr(ghc),
                    _lib.ptr(gf), _lib.ptr(gh2), _lib.ptr(gh1),
                    _lib.ptr(dw_workspace(dev)), _lib.ptr(flat),
                    _lib.stream_ptr(dev)), 'xrd_vox_dw')
            gp = [g.reshape(shp) if need else None for g, shp, need in zip(
                flat.split(_PARAM_SIZES), _PARAM_SHAPES,
                ctx.needs_input_grad[6:])]
        return (g_xyz, g_emb, None, None, None, None, *gp)


def points(decoder, xyz, voxel_idx, map_states, voxel_size):
    """-> {'sdf' [P], 'color' [P,3]} or None when the fused kernels do not
    cover this decoder / device (the caller then uses the modular path)"""
    if not xyz.is_cuda:
        return None
    ps = decoder_params(decoder)
    if ps is None:
        return None
    dev = xyz.device
    sdf, rgb = _VoxPointsFn.apply(
        xyz, map_states['voxel_vertex_emb'], voxel_idx,
        map_states['voxel_center_xyz'].to(dev),
        map_states['voxel_vertex_idx'].to(dev), voxel_size, *ps)
    return {'sdf': sdf, 'color': rgb}


# ---------------------------------------------------------------------------
# The ray side of an iteration with static capacities (csrc/vox_rays.hip):
# octree traversal -> hit sort -> inverse-CDF sampling -> point compaction ->
# [voxel features + decoder] -> compositing + the four loss terms, and the
# backward of all of it, as a fixed launch sequence without a host sync.
# ---------------------------------------------------------------------------
N_MAX_HITS = 50          # n_max of svo_ray_intersect (voxel_helpers :655)
OVERFLOW_BITS = {1: 'samples per ray (s_cap)', 2: 'points (p_cap)',
                 4: 'sample row with a hole'}


class RayWorkspace:
    """every buffer of one ray batch shape, allocated once: captured graphs
    keep their addresses, and the rows of the decoder operands beyond the live
    point count stay finite (the weight-gradient GEMMs run over the capacity)"""

    def __init__(self, n_rays, s_cap, p_cap, need_w, device):
        self.n, self.s_cap, self.p_cap, self.need_w = n_rays, s_cap, p_cap, \
            need_w
        # sharded mapping: makes the loss normalisers of the size record
        # batch-global between sampling and the loss (None = single process)
        self.meta_sync = None
        dev = device
        i = dict(dtype=torch.int32, device=dev)
        f = dict(dtype=torch.float32, device=dev)
        n, s, p = n_rays, s_cap, p_cap
        self.hit_idx = torch.zeros(n, N_MAX_HITS, **i)
        self.hit_min = torch.zeros(n, N_MAX_HITS, **f)
        self.hit_max = torch.zeros(n, N_MAX_HITS, **f)
        self.probs = torch.zeros(n, N_MAX_HITS, **f)
        self.steps = torch.zeros(n, **f)
        self.hit = torch.zeros(n, **i)
        self.rank = torch.zeros(n, **i)
        self.hit_rays = torch.zeros(n, **i)
        self.s_idx = torch.zeros(n, s, **i)
        self.s_depth = torch.zeros(n, s, **f)
        self.cnt = torch.zeros(n, **i)
        self.offs = torch.zeros(n + 1, **i)
        self.xyz = torch.zeros(p, 3, **f)
        self.vox = torch.zeros(p, **i)
        self.meta = torch.zeros(_lib.lib().xrd_vox_meta_len(), **i)
        self.acc = torch.zeros(4, dtype=torch.float64, device=dev)
        self.loss = torch.zeros(5, **f)
        self.scale = torch.zeros(4, **f)
        self.depth = torch.zeros(n, **f)
        self.rgb = torch.zeros(n, 3, **f)
        self.sdf_pt = torch.zeros(p, **f)
        self.rgb_pt = torch.zeros(p, 3, **f)
        self.masks = torch.zeros(p, 3, 4, **i)
        self.g_sdf = torch.zeros(p, **f)
        self.g_rgb = torch.zeros(p, 3, **f)
        self.g_xyz = torch.zeros(p, 3, **f)
        self.g_o = torch.zeros(n, 3, **f)
        self.g_d = torch.zeros(n, 3, **f)
        self.zero_n = torch.zeros(n, **f)
        if need_w:
            self.sx = torch.zeros(p, 16, **f)
            self.sh1, self.sh2, self.sf, self.shc = (
                torch.zeros(p, 128, **f) for _ in range(4))
            self.gc3 = torch.zeros(p, 4, **f)
            self.ghc, self.gf, self.gh2, self.gh1 = (
                torch.zeros(p, 128, **f) for _ in range(4))
        else:
            self.sx = self.sh1 = self.sh2 = self.sf = self.shc = None
            self.gc3 = self.ghc = self.gf = self.gh2 = self.gh1 = None

    @property
    def n_pts_dev(self):
        return C.c_void_p(self.meta.data_ptr() + 4 * 4)

    def overflow(self):
        """(bits, meta list) — one device->host copy.  ``bits`` covers EVERY
        launch since the last call (the sticky slots 11..13 of the record,
        csrc/vox_rays.hip), not only the last one; row_len (m[2]) and the
        wanted sample count (m[14]) are the maxima over those launches.  The
        sticky slots are cleared here."""
        m = self.meta.tolist()
        bits = m[5] | (8 if m[10] else 0) | m[11]
        m[2] = max(m[2], m[12])
        m[14] = max(m[14], m[13])
        if m[11] or m[12] or m[13]:
            self.meta[11:14].zero_()
        return bits, m


def flatten_decoder(params):
    """re-seat the ten decoder tensors as back-to-back views of ONE buffer (+
    a trailing zero the packing index points its padding at): packing becomes
    one gather, and Adam steps the decoder with one launch (its gradient comes
    out of xrd_vox_dw as one flat tensor too).  Idempotent; values kept."""
    p0 = params[0]
    flat = getattr(p0, '_xrd_flat', None)
    nxt = flat.data_ptr() if flat is not None else None
    ok = flat is not None
    for p in params:
        ok = ok and p.data_ptr() == nxt and p.is_contiguous()
        nxt = (nxt or 0) + 4 * p.numel()
    if ok:
        return flat
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1).float() for p in params] +
                         [p0.new_zeros(1)])
        off = 0
        for p in params:
            p.data = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
    p0._xrd_flat = flat
    return flat


# size-record slots (csrc/vox_rays.hip, enum Meta)
META_SUM = [1, 6, 7, 8]   # hit rays, free-space / band samples, usable depths
META_MAX = [3]            # longest sample row


def allreduce_meta(meta):
    """sharded mapping: the reference's losses are means over the hit rays /
    the PADDED [hit rays, longest row] array with batch-global balancing
    weights (sparse_voxel.py:103-143) — the counts are summed and the row
    length maximised over the ranks before the loss kernels read them"""
    import torch.distributed as dist
    cnt = meta[META_SUM].contiguous()
    mx = meta[META_MAX].contiguous()
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    meta[META_SUM] = cnt
    meta[META_MAX] = mx


def _pack(params, like):
    return flatten_decoder(params)[pack_index(like.device)]


def sample_rays(ws, ms, cfg, rays_o, rays_d, target_d, noise, keep=None):
    """hits, samples and points of a ray batch into ``ws``.  ``keep`` [n] u8
    (sharded mapping): the rays this rank renders — the others are sampled
    too (whole-batch regrouping and loss normalisers) but leave no points"""
    lib = _lib.lib()
    dev = rays_o.device
    centres = ms['voxel_center_xyz']
    children = ms['voxel_structure']
    with _Timed(('vox_sample_rays', ws.n)):
        _lib.check(lib.xrd_vox_sample_rays_shard(
            ws.n, N_MAX_HITS, ws.s_cap, ws.p_cap, centres.shape[0],
            _lib.ptr(centres), _lib.ptr(children), float(cfg.voxel_size),
            float(cfg.max_distance), float(cfg.step_size),
            float(cfg.training_trunc * cfg.data_sc_factor),
            float(cfg.max_dpeth), _lib.ptr(rays_o), _lib.ptr(rays_d),
            _lib.ptr(target_d), _lib.ptr(noise), _lib.ptr(keep),
            _lib.ptr(ws.hit_idx),
            _lib.ptr(ws.hit_min), _lib.ptr(ws.hit_max), _lib.ptr(ws.probs),
            _lib.ptr(ws.steps), _lib.ptr(ws.hit), _lib.ptr(ws.rank),
            _lib.ptr(ws.hit_rays), _lib.ptr(ws.s_idx), _lib.ptr(ws.s_depth),
            _lib.ptr(ws.cnt), _lib.ptr(ws.offs), _lib.ptr(ws.xyz),
            _lib.ptr(ws.vox), _lib.ptr(ws.meta), _lib.ptr(ws.acc),
            _lib.stream_ptr(dev)), 'xrd_vox_sample_rays_shard')


def _points_fwd(ws, ms, cfg, packed, save):
    lib = _lib.lib()
    dev = ws.xyz.device
    with _Timed(('vox_points_fwd', ws.p_cap, save)):
        _lib.check(lib.xrd_vox_points_fwd(
            ws.p_cap, _lib.ptr(ws.xyz), _lib.ptr(ws.vox),
            _lib.ptr(ms['voxel_center_xyz']), _lib.ptr(ms['voxel_vertex_idx']),
            _lib.ptr(ms['voxel_vertex_emb'].detach()), float(cfg.voxel_size),
            _lib.ptr(packed), _lib.ptr(ws.sdf_pt), _lib.ptr(ws.rgb_pt),
            _lib.ptr(ws.sx if save else None),
            _lib.ptr(ws.sh1 if save else None),
            _lib.ptr(ws.sh2 if save else None),
            _lib.ptr(ws.sf if save else None),
            _lib.ptr(ws.shc if save else None), _lib.ptr(ws.masks),
            ws.n_pts_dev, _lib.stream_ptr(dev)), 'xrd_vox_points_fwd')


def _render_fwd(ws, cfg, target_d, target_s, with_loss, z_min=None,
                weights=None):
    lib = _lib.lib()
    dev = ws.xyz.device
    tr = float(cfg.training_trunc * cfg.data_sc_factor)
    with _Timed(('vox_render_fwd', ws.n)):
        _lib.check(lib.xrd_vox_render_fwd(
            ws.n, ws.s_cap, ws.p_cap, tr, float(cfg.max_dpeth),
            _lib.ptr(ws.hit), _lib.ptr(ws.cnt), _lib.ptr(ws.offs),
            _lib.ptr(ws.s_depth), _lib.ptr(ws.sdf_pt), _lib.ptr(ws.rgb_pt),
            _lib.ptr(target_d), _lib.ptr(target_s), _lib.ptr(ws.meta),
            _lib.ptr(ws.depth), _lib.ptr(ws.rgb), _lib.ptr(z_min),
            _lib.ptr(weights), _lib.ptr(ws.acc) if with_loss else None,
            float(cfg.trainging_rgb_weight), float(cfg.trainging_depth_weight),
            float(cfg.trainging_sdf_weight), float(cfg.trainging_fs_weight),
            _lib.ptr(ws.loss), _lib.ptr(ws.scale), _lib.stream_ptr(dev)),
            'xrd_vox_render_fwd')


class _VoxRenderLossFn(torch.autograd.Function):
    """(rays, embeddings, decoder) -> the summed Vox-Fusion loss of the batch:
    SparseVoxel.get_outputs + get_loss_dict (sparse_voxel.py:103-143,160-275)
    as ~12 launches each way"""

    @staticmethod
    def forward(ctx, rays_o, rays_d, emb, target_d, target_s, noise, ws, ms,
                cfg, flat, *params):
        rays_o = rays_o.detach().float().contiguous()
        rays_d = rays_d.detach().float().contiguous()
        target_d = target_d.detach().float().reshape(-1).contiguous()
        target_s = target_s.detach().float().contiguous()
        need_w = any(ctx.needs_input_grad[10:])
        assert not need_w or ws.need_w
        packed = flat.detach()[pack_index(rays_o.device)]
        sample_rays(ws, ms, cfg, rays_o, rays_d, target_d, noise,
                    getattr(ws, 'keep', None))
        if ws.meta_sync is not None:
            ws.meta_sync(ws.meta)
        _points_fwd(ws, ms, cfg, packed, need_w)
        _render_fwd(ws, cfg, target_d, target_s, True)
        ctx.ws, ctx.ms, ctx.cfg, ctx.need_w = ws, ms, cfg, need_w
        ctx.save_for_backward(packed, target_d, target_s)
        ctx.mark_non_differentiable(ws.loss, ws.depth, ws.rgb)
        # gradients of the non-differentiable outputs arrive as None instead
        # of materialised zero tensors (one fill launch each)
        ctx.set_materialize_grads(False)
        return ws.loss[4].clone(), ws.loss, ws.depth, ws.rgb

    @staticmethod
    def backward(ctx, g_loss, *unused):
        lib = _lib.lib()
        ws, ms, cfg, need_w = ctx.ws, ctx.ms, ctx.cfg, ctx.need_w
        packed, target_d, target_s = ctx.saved_tensors
        dev = packed.device
        st = _lib.stream_ptr(dev)
        tr = float(cfg.training_trunc * cfg.data_sc_factor)
        need_o, need_d, need_emb = ctx.needs_input_grad[:3]
        g_up = g_loss.detach().float().contiguous()
        with _Timed(('vox_render_bwd', ws.n)):
            _lib.check(lib.xrd_vox_render_bwd(
                ws.n, ws.s_cap, ws.p_cap, tr, float(cfg.max_dpeth),
                _lib.ptr(ws.hit), _lib.ptr(ws.cnt), _lib.ptr(ws.offs),
                _lib.ptr(ws.s_depth), _lib.ptr(ws.sdf_pt),
                _lib.ptr(ws.rgb_pt), _lib.ptr(target_d), _lib.ptr(target_s),
                _lib.ptr(ws.meta), _lib.ptr(ws.scale), _lib.ptr(g_up),
                _lib.ptr(ws.g_sdf), _lib.ptr(ws.g_rgb), st),
                'xrd_vox_render_bwd')
        emb = ms['voxel_vertex_emb'].detach()
        g_emb = torch.zeros_like(emb) if need_emb else None
        need_xyz = need_o or need_d
        with _Timed(('vox_points_bwd', ws.p_cap, need_w)):
            _lib.check(lib.xrd_vox_points_bwd(
                ws.p_cap, _lib.ptr(ws.xyz), _lib.ptr(ws.vox),
                _lib.ptr(ms['voxel_center_xyz']),
                _lib.ptr(ms['voxel_vertex_idx']), _lib.ptr(emb),
                float(cfg.voxel_size), _lib.ptr(packed), _lib.ptr(ws.rgb_pt),
                _lib.ptr(ws.masks), _lib.ptr(ws.g_sdf), _lib.ptr(ws.g_rgb),
                _lib.ptr(ws.g_xyz if need_xyz else None), _lib.ptr(g_emb),
                _lib.ptr(ws.gc3 if need_w else None),
                _lib.ptr(ws.ghc if need_w else None),
                _lib.ptr(ws.gf if need_w else None),
                _lib.ptr(ws.gh2 if need_w else None),
                _lib.ptr(ws.gh1 if need_w else None), ws.n_pts_dev, st),
                'xrd_vox_points_bwd')
        g_o = g_d = None
        if need_xyz:
            with _Timed(('vox_ray_grads', ws.n)):
                _lib.check(lib.xrd_vox_ray_grads(
                    ws.n, ws.s_cap, ws.p_cap, _lib.ptr(ws.hit),
                    _lib.ptr(ws.cnt), _lib.ptr(ws.offs), _lib.ptr(ws.s_depth),
                    _lib.ptr(ws.g_xyz), _lib.ptr(ws.g_o), _lib.ptr(ws.g_d),
                    st), 'xrd_vox_ray_grads')
            g_o, g_d = ws.g_o, ws.g_d
        gp = [None] * 10
        if need_w:
            flat = torch.empty(lib.xrd_vox_flat_len(), dtype=torch.float32,
                               device=dev)
            with _Timed(('vox_dw', ws.p_cap, True)):
                _lib.check(lib.xrd_vox_dw(
                    ws.p_cap, ws.n_pts_dev, _lib.ptr(ws.sx), _lib.ptr(ws.sh1),
                    _lib.ptr(ws.sh2), _lib.ptr(ws.sf), _lib.ptr(ws.shc),
                    _lib.ptr(ws.gc3), _lib.ptr(ws.ghc), _lib.ptr(ws.gf),
                    _lib.ptr(ws.gh2), _lib.ptr(ws.gh1),
                    _lib.ptr(dw_workspace(dev)), _lib.ptr(flat), st),
                    'xrd_vox_dw')
            gp = [g.reshape(shp) if need else None for g, shp, need in zip(
                flat.split(_PARAM_SIZES), _PARAM_SHAPES,
                ctx.needs_input_grad[10:])]
        return (g_o if need_o else None, g_d if need_d else None, g_emb, None,
                None, None, None, None, None, None, *gp)


def render_loss(decoder, ws, ms, cfg, rays_o, rays_d, target_d, target_s,
                noise, map_grads=True):
    """-> (loss, loss terms [5] = rgb, depth, sdf, fs, sum; depth [N]; rgb
    [N,3]) or None when the decoder is not the shape the kernels cover.
    ``map_grads`` False (tracking): only the ray gradients are produced (the
    reference's autograd also fills the map gradients there, and never uses
    them)"""
    ps = decoder_params(decoder)
    if ps is None or not rays_o.is_cuda:
        return None
    emb = ms['voxel_vertex_emb']
    flat = flatten_decoder(ps)
    if not map_grads:
        ps = [p.detach() for p in ps]
        emb = emb.detach()
    return _VoxRenderLossFn.apply(rays_o, rays_d, emb, target_d, target_s,
                                  noise, ws, ms, cfg, flat, *ps)


@torch.no_grad()
def render(decoder, ws, ms, cfg, rays_o, rays_d, noise, z_min=None,
           weights=None):
    """inference: depth [N], rgb [N,3] of a ray batch (views of ``ws``)"""
    ps = decoder_params(decoder)
    if ps is None or not rays_o.is_cuda:
        return None
    rays_o = rays_o.detach().float().contiguous()
    rays_d = rays_d.detach().float().contiguous()
    # the target depth only feeds the loss counters
    sample_rays(ws, ms, cfg, rays_o, rays_d, ws.zero_n, noise)
    _points_fwd(ws, ms, cfg, _pack(ps, rays_o), False)  # ps: the module's own
    _render_fwd(ws, cfg, None, None, False, z_min=z_min, weights=weights)
    return ws.depth, ws.rgb
