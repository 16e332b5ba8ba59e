"""Host side of the fused Co-SLAM ray renderer (csrc/coslam_render.hip,
include/xrdslam_hip.h ``xrd_coslam_*``): one launch renders a batch of rays
(depth-guided sampling, hash grid + OneBlob, both MLPs, SDF compositing), one
launch back-propagates to rays, hash table and decoder weights.

It covers the reference's default Co-SLAM model only (oneGrid, HashGrid 16x2,
OneBlob 16 bins, hidden 32, geo feature 15, two layers per MLP, depth-guided
samples <= 48 per ray); ``supported()`` says whether a model qualifies — the
caller (slam/models/joint_encoding.py) otherwise uses the modular HIP
encodings with torch MLPs.  No CPU path."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib

_INDEX = {}

# bench.py: per-launch HIP-event timing on the launch stream, keyed
# (kernel, n_rays, ray_grads, map_grads); None = off
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


def _index(device):
    """(pack gather index into [flat, 0], dW gather index) on ``device``"""
    key = str(device)
    if key not in _INDEX:
        lib = _lib.lib()
        pack = np.zeros(lib.xrd_coslam_pack_len(), np.int32)
        dw = np.zeros(lib.xrd_coslam_flat_len(), np.int32)
        _lib.check(lib.xrd_coslam_index(pack.ctypes.data, dw.ctypes.data),
                   'xrd_coslam_index')
        pack = np.where(pack < 0, lib.xrd_coslam_flat_len(), pack)
        _INDEX[key] = (torch.from_numpy(pack.astype(np.int64)).to(device),
                       torch.from_numpy(dw.astype(np.int64)).to(device))
    return _INDEX[key]


def flat_decoder(decoder) -> torch.Tensor:
    """state_dict order: color_net.model.0, color_net.model.2,
    sdf_net.model.0, sdf_net.model.2 (differentiable concatenation)"""
    ws = [decoder.color_net.model[0].weight, decoder.color_net.model[2].weight,
          decoder.sdf_net.model[0].weight, decoder.sdf_net.model[2].weight]
    return torch.cat([w.reshape(-1) for w in ws])


def supported(model) -> bool:
    cfg = model.config
    enc = getattr(model.embed_fn, 'encoding_config', None)
    pos = getattr(model.embedpos_fn, 'encoding_config', None)
    if enc is None or pos is None or not cfg.oneGrid or cfg.tcnn_network:
        return False
    if not cfg.tcnn_encoding or cfg.training_n_importance > 0:
        return False
    if enc.get('otype') != 'HashGrid' or model.embed_fn.n_levels != 16 or \
            pos.get('otype') != 'OneBlob' or model.embedpos_fn.n_bins != 16:
        return False
    if (cfg.hidden_dim, cfg.hidden_dim_color, cfg.geo_feat_dim,
            cfg.num_layers, cfg.num_layers_color) != (32, 32, 15, 2, 2):
        return False
    return cfg.training_n_range_d + cfg.training_n_sample_d <= 48 and \
        cfg.training_n_range_d >= 1


class SceneTables:
    """per-model constant tables on the device (linspaces exactly as the
    reference builds them: torch.linspace on the CPU, then moved)"""

    def __init__(self, model, device):
        cfg = model.config
        self.t_near = torch.linspace(-cfg.training_range_d,
                                     cfg.training_range_d,
                                     steps=cfg.training_n_range_d).to(device)
        self.t_far = torch.linspace(cfg.cam_near, cfg.cam_far,
                                    steps=cfg.training_n_range_d).to(device)
        self.t_uniform = torch.linspace(
            cfg.cam_near, cfg.cam_far,
            max(cfg.training_n_sample_d, 1)).to(device)
        self.device = torch.device(device)
        self._ws = None

    def workspace(self, n_rays, n_extra=0):
        need = _lib.lib().xrd_coslam_bwd_ws_floats_extra(n_rays, n_extra)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.float32,
                                   device=self.device)
        return self._ws


def make_scene(model, tables, table_params, pack):
    cfg = model.config
    enc = model.embed_fn
    sc = _lib.CoslamScene()
    bb = model.bounding_box.detach().cpu().double().numpy()
    for d in range(3):
        sc.bound[2 * d], sc.bound[2 * d + 1] = bb[d, 0], bb[d, 1]
    for l in range(16):
        sc.lv_scale[l] = float(enc._scales[l])
        sc.lv_res[l] = int(enc._res[l])
        sc.lv_size[l] = int(enc._sizes[l])
        sc.lv_offset[l] = int(enc._offsets[l])
    sc.table = table_params.data_ptr()
    sc.pack = pack.data_ptr()
    sc.t_near = tables.t_near.data_ptr()
    sc.t_far = tables.t_far.data_ptr()
    sc.t_uniform = tables.t_uniform.data_ptr()
    sc.n_range_d = cfg.training_n_range_d
    sc.n_sample_d = cfg.training_n_sample_d
    sc.perturb = 1 if cfg.training_perturb > 0. else 0
    sc.white_bkgd = 1 if cfg.training_white_bkgd else 0
    sc.trunc = float(cfg.training_trunc)
    sc.sc_factor = float(cfg.data_sc_factor)
    return sc


def track_pack(model, dev, refresh=False):
    """the decoder's packed MFMA fragments for calls that do not train it
    (tracking): a static buffer, re-packed only when ``refresh`` finds the
    weights changed (or the owner invalidated ``model._track_pack_key`` after
    a mapping call whose captured optimiser steps torch's version counters do
    not see).  Inside the tracking iteration (eager or captured) the buffer is
    just read: the per-call concatenate / fill / gather launches are gone."""
    dev = torch.device(dev)
    buf = model.__dict__.get('_track_pack')
    ws = [model.decoder.color_net.model[0].weight,
          model.decoder.color_net.model[2].weight,
          model.decoder.sdf_net.model[0].weight,
          model.decoder.sdf_net.model[2].weight]
    if buf is None or buf.device != dev:
        buf = model._track_pack = torch.empty(
            _lib.lib().xrd_coslam_pack_len(), dtype=torch.float32, device=dev)
        model._track_pack_key = None
        refresh = True
    if refresh:
        key = tuple((w._version, w.data_ptr()) for w in ws)
        if key != model.__dict__.get('_track_pack_key'):
            pack_idx, _ = _index(dev)
            with torch.no_grad():
                flat = torch.cat([w.detach().reshape(-1).float() for w in ws])
                buf.copy_(torch.cat([flat, flat.new_zeros(1)])[pack_idx])
            model._track_pack_key = key
    return buf


class _CoslamRenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, target_d, rnd, table, flat, model,
                tables, pack=None):
        lib = _lib.lib()
        dev = rays_o.device
        if pack is None:
            pack_idx, _ = _index(dev)
            with torch.no_grad():
                pack = torch.cat([flat.detach().float(),
                                  flat.new_zeros(1)])[pack_idx].contiguous()
        ro = rays_o.detach().float().contiguous()
        rd = rays_d.detach().float().contiguous()
        td = target_d.detach().float().reshape(-1).contiguous()
        rn = None if rnd is None else rnd.detach().float().contiguous()
        n = ro.shape[0]
        sc = make_scene(model, tables, table.detach(), pack)
        S = sc.n_range_d + sc.n_sample_d
        z_vals = torch.empty(n, S, dtype=torch.float32, device=dev)
        raw = torch.empty(n, S, 4, dtype=torch.float32, device=dev)
        maps = torch.empty(n, 8, dtype=torch.float32, device=dev)
        with _Timed(('coslam_fwd', n, False, False)):
            _lib.check(lib.xrd_coslam_render_fwd(
                C.byref(sc), n, _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(td),
                _lib.ptr(rn), _lib.ptr(z_vals), _lib.ptr(raw), _lib.ptr(maps),
                _lib.stream_ptr(dev)), 'xrd_coslam_render_fwd')
        ctx.sc, ctx.tables, ctx.model = sc, tables, model
        # (the smoothness term's lattice joins THIS backward's table scatter
        # when it runs first, see _SmoothFn)
        # (only a forward that CAN be followed by a backward arms the
        # hand-over: a render under no_grad left the flag set, and the next
        # smoothness backward parked a table gradient nobody picked up)
        model._render_bwd_pending = bool(table.requires_grad and
                                         torch.is_grad_enabled())
        model._smooth_stash = None
        ctx.save_for_backward(ro, rd, z_vals, raw, table, pack)
        ctx.mark_non_differentiable(z_vals)
        # gradients of the non-differentiable outputs arrive as None instead
        # of materialised zero tensors (one fill launch each)
        ctx.set_materialize_grads(False)
        return maps, z_vals, raw

    @staticmethod
    def backward(ctx, g_maps, _gz, g_raw):
        lib = _lib.lib()
        ro, rd, z_vals, raw, table, pack = ctx.saved_tensors
        dev = ro.device
        n = ro.shape[0]
        need_rays = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_map = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
        g_o = g_d = None
        if need_rays:
            # back to back: the kernel's zero-fill of both is one launch
            g_od = torch.empty(2, n, 3, dtype=torch.float32, device=dev)
            g_o, g_d = g_od[0], g_od[1]
        g_table = g_dw = None
        if need_map:
            g_table = torch.empty_like(table)  # fully overwritten
            g_dw = torch.empty(lib.xrd_coslam_dw_len(), dtype=torch.float32,
                               device=dev)
        g_maps = torch.zeros(n, 8, dtype=torch.float32, device=dev) \
            if g_maps is None else g_maps.float().contiguous()
        if g_raw is not None:
            g_raw = g_raw.float().contiguous()
        model = ctx.model
        model._render_bwd_pending = False
        stash, model._smooth_stash = getattr(model, '_smooth_stash', None), \
            None
        ex = ed = None
        n_extra = 0
        if need_map and stash is not None:
            ex, ed = stash
            n_extra = ex.shape[0]
        ws = _lib.ptr(ctx.tables.workspace(n, n_extra)) if need_map else None
        with _Timed(('coslam_bwd', n, bool(need_rays), bool(need_map))):
            _lib.check(lib.xrd_coslam_render_bwd_extra(
                C.byref(ctx.sc), n, _lib.ptr(ro), _lib.ptr(rd),
                _lib.ptr(z_vals), _lib.ptr(raw), _lib.ptr(g_maps),
                _lib.ptr(g_raw), _lib.ptr(g_o), _lib.ptr(g_d),
                _lib.ptr(g_table), _lib.ptr(g_dw), n_extra, _lib.ptr(ex),
                _lib.ptr(ed), ws, _lib.stream_ptr(dev)),
                'xrd_coslam_render_bwd_extra')
        g_flat = None
        if need_map:
            g_flat = g_dw[_index(dev)[1]]
        return g_o, g_d, None, None, g_table, g_flat, None, None, None


class _SmoothFn(torch.autograd.Function):
    """JointEncoding.smoothness (joint_encoding.py:165-197) x ``scale`` on the
    kernels of xrd_hashgrid_tv: lattice points, hash features, TV loss and
    d loss / d features in three launches.  The backward does NOT scatter into
    the table itself when a fused render backward of the same iteration is
    still to come: it leaves (points, d features) for it, and the table
    gradient of both leaves in ONE scatter launch.  If no render backward is
    pending it scatters on its own (xrd_hashgrid_bwd)."""

    @staticmethod
    def forward(ctx, table, model, side, voxel, margin, scale, r_off,
                r_shift):
        lib = _lib.lib()
        enc = model.embed_fn
        dev = table.device
        L = enc.n_output_dims // 2
        P = side**3
        f = dict(dtype=torch.float32, device=dev)
        pts, feat = torch.empty(P, 3, **f), torch.empty(P, 2 * L, **f)
        dfeat = torch.empty(P, 2 * L, **f)
        loss = torch.empty((), dtype=torch.float64, device=dev)
        bb = np.ascontiguousarray(
            model.bounding_box.detach().cpu().double().numpy().reshape(-1))
        t = table.detach()
        _lib.check(lib.xrd_hashgrid_tv(
            L, enc._scales.ctypes.data, enc._res.ctypes.data,
            enc._sizes.ctypes.data, enc._offsets.ctypes.data, _lib.ptr(t),
            int(side), bb.ctypes.data, float(voxel), float(margin),
            _lib.ptr(r_off), _lib.ptr(r_shift), float(scale), _lib.ptr(pts),
            _lib.ptr(feat), _lib.ptr(dfeat), _lib.ptr(loss),
            _lib.stream_ptr(dev)), 'xrd_hashgrid_tv')
        ctx.model, ctx.L = model, L
        ctx.save_for_backward(pts, dfeat, t)
        return loss.float()

    @staticmethod
    def backward(ctx, g):
        pts, dfeat, t = ctx.saved_tensors
        model = ctx.model
        d = dfeat * g.float()
        if getattr(model, '_render_bwd_pending', False):
            model._smooth_stash = (pts, d)
            return (None, ) * 8
        lib = _lib.lib()
        enc = model.embed_fn
        g_table = torch.zeros_like(t)
        _lib.check(lib.xrd_hashgrid_bwd(
            ctx.L, enc._scales.ctypes.data, enc._res.ctypes.data,
            enc._sizes.ctypes.data, enc._offsets.ctypes.data, pts.shape[0],
            _lib.ptr(pts), _lib.ptr(t), _lib.ptr(d.contiguous()),
            _lib.ptr(g_table), None, _lib.stream_ptr(t.device)),
            'xrd_hashgrid_bwd')
        return (g_table, ) + (None, ) * 7


def smoothness(model, side, voxel, margin, scale):
    """the smoothness term of the mapping loss, already multiplied by
    ``scale`` (= its weight); draws the reference's two random vectors with
    the model's generator hook (same RNG consumption as the torch path)"""
    dev = model.embed_fn.params.device
    bb = model._bbox(dev)
    volume = bb[:, 1] - bb[:, 0]
    offset_max = volume - (side * voxel) - 2 * margin
    r_off = model._rand((3, ), offset_max).double().contiguous()
    r_shift = model._rand((1, 1, 1, 3), volume).double().reshape(3) \
        .contiguous()
    return _SmoothFn.apply(model.embed_fn.params, model, int(side),
                           float(voxel), float(margin), float(scale), r_off,
                           r_shift)


def render(model, tables, rays_o, rays_d, target_d, rnd, train_map=True):
    """-> dict like JointEncoding.render_rays (joint_encoding.py:250-344).
    ``train_map=False`` (tracking: only the pose is stepped) skips the hash
    table / decoder gradients, which nobody consumes."""
    if train_map:
        maps, z_vals, raw = _CoslamRenderFn.apply(
            rays_o, rays_d, target_d, rnd, model.embed_fn.params,
            flat_decoder(model.decoder), model, tables)
    else:
        # the first call after the weights changed re-packs (a caller that
        # replays captured tracking iterations refreshes before the replays:
        # CoSLAM.pre_precessing)
        pack = track_pack(model, rays_o.device,
                          refresh=not torch.cuda.is_current_stream_capturing())
        maps, z_vals, raw = _CoslamRenderFn.apply(
            rays_o, rays_d, target_d, rnd, model.embed_fn.params.detach(),
            None, model, tables, pack)
    return {'rgb': maps[:, 0:3], 'depth': maps[:, 3], 'disp_map': maps[:, 6],
            'acc_map': maps[:, 5], 'depth_var': maps[:, 4], 'z_vals': z_vals,
            'raw': raw, '_maps': maps}


class _CoslamLossFn(torch.autograd.Function):
    """total of the rgb / depth / sdf / free-space terms of
    JointEncoding.get_loss_dict, two launches, gradients produced with it.
    ``sharded``: this rank holds a shard of the mapping batch — the seven
    batch sums are all-reduced between the two launches so that normalisers
    and the batch-global balancing weights are those of the whole batch."""

    @staticmethod
    def forward(ctx, maps, z_vals, raw, target_d, target_rgb, cfgv, sharded,
                n_live=None):
        lib = _lib.lib()
        dev = maps.device
        n, S = z_vals.shape
        m = maps.detach().float().contiguous()
        r = raw.detach().float().contiguous()
        z = z_vals.detach().float().contiguous()
        td = target_d.detach().float().reshape(-1).contiguous()
        tc = target_rgb.detach().float().contiguous()
        loss5 = torch.empty(5, dtype=torch.float32, device=dev)
        g_maps = torch.empty(n, 8, dtype=torch.float32, device=dev)
        g_raw = torch.empty(n, S, 4, dtype=torch.float32, device=dev)
        stats = torch.empty(n, 8, dtype=torch.float32, device=dev)
        w_rgb, w_d, w_sdf, w_fs, trunc, dtrunc, miss = [float(v) for v in cfgv]
        st = _lib.stream_ptr(dev)
        if n_live is not None:
            # capacity batch of a persistent mapping graph: the live ray
            # count is read on the device
            if sharded:
                raise _lib.XrdError('live-count batches are not sharded')
            assert n_live.dtype == torch.int32 and n_live.is_cuda
            _lib.check(lib.xrd_coslam_loss_live(
                n, S, w_rgb, w_d, w_sdf, w_fs, trunc, dtrunc, miss,
                _lib.ptr(m), _lib.ptr(z), _lib.ptr(r), _lib.ptr(td),
                _lib.ptr(tc), _lib.ptr(n_live), _lib.ptr(loss5),
                _lib.ptr(g_maps), _lib.ptr(g_raw), _lib.ptr(stats), st),
                'xrd_coslam_loss_live')
            ctx.save_for_backward(g_maps, g_raw)
            ctx.mark_non_differentiable(loss5)
            ctx.set_materialize_grads(False)
            return loss5[0], loss5
        _lib.check(lib.xrd_coslam_loss_stats(
            n, S, trunc, dtrunc, miss, _lib.ptr(m), _lib.ptr(z), _lib.ptr(r),
            _lib.ptr(td), _lib.ptr(tc), _lib.ptr(stats), st),
            'xrd_coslam_loss_stats')
        totals, n_total = None, n
        if sharded:
            import torch.distributed as dist
            t = torch.cat([stats[:, :7].double().sum(0),
                           torch.tensor([float(n)], dtype=torch.float64,
                                        device=dev)])
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            totals, n_total = t[:7].contiguous(), int(round(float(t[7])))
        _lib.check(lib.xrd_coslam_loss_grads(
            n, S, w_rgb, w_d, w_sdf, w_fs, trunc, dtrunc, miss, _lib.ptr(m),
            _lib.ptr(z), _lib.ptr(r), _lib.ptr(td), _lib.ptr(tc),
            _lib.ptr(stats), _lib.ptr(totals), n_total, _lib.ptr(loss5),
            _lib.ptr(g_maps), _lib.ptr(g_raw), st), 'xrd_coslam_loss_grads')
        ctx.save_for_backward(g_maps, g_raw)
        ctx.mark_non_differentiable(loss5)
        ctx.set_materialize_grads(False)
        return loss5[0], loss5

    @staticmethod
    def backward(ctx, g, _g5):
        g_maps, g_raw = ctx.saved_tensors
        return g * g_maps, None, g * g_raw, None, None, None, None, None


def loss(model, outputs, target_d, target_rgb, sharded=False, n_live=None):
    """-> (total with autograd, loss5 = [total, rgb, depth, sdf, fs]).  With
    ``sharded`` every rank gets the GLOBAL loss value and the gradient of its
    own rays; summing the gradients over ranks gives the single-GPU one."""
    cfg = model.config
    cfgv = (cfg.trainging_rgb_weight, cfg.trainging_depth_weight,
            cfg.trainging_sdf_weight, cfg.trainging_fs_weight,
            cfg.training_trunc * cfg.data_sc_factor, cfg.cam_depth_trunc,
            cfg.training_rgb_missing)
    return _CoslamLossFn.apply(outputs['_maps'], outputs['z_vals'],
                               outputs['raw'], target_d, target_rgb, cfgv,
                               bool(sharded), n_live)
