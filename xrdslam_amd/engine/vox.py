"""Vox-Fusion fused "voxel features + decoder" (csrc/vox_render.hip) behind
autograd.

    sdf, rgb = points(decoder, xyz, voxel_idx, map_states, voxel_size)

replaces ``Decoder(get_features(samples, map_states, voxel_size))`` of the
reference (slam/models/sparse_voxel.py:230-238,
voxel_helpers_voxfusion.py:109-123, decoder_voxfusion.py:123-149) for the
model's default decoder (in_dim 16, width 128, depth 2, no positional
encoding).  PyTorch is device memory + autograd plumbing; the decoder's weight
gradients are five GEMMs over the points on the operands the backward kernel
writes (rocBLAS through ``torch.mm``: plain library GEMMs)."""
from __future__ import annotations

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
            _lib.ptr(masks), st), 'xrd_vox_points_fwd')
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
            _lib.ptr(ghc), _lib.ptr(gf), _lib.ptr(gh2), _lib.ptr(gh1),
            _lib.stream_ptr(dev)), 'xrd_vox_points_bwd')
        gp = [None] * 10
        if need_w:
            # dW = G^T A over the points (plain GEMMs)
            gout = torch.cat([gc3[:, 3:4], gf], 1)          # [P,129]
            gp = [gh1.t() @ sx, gh1.sum(0), gh2.t() @ sh1, gh2.sum(0),
                  gout.t() @ sh2, gout.sum(0),
                  ghc.t() @ torch.cat([sf, sx], 1), ghc.sum(0),
                  gc3[:, :3].t() @ shc, gc3[:, :3].sum(0)]
            gp = [g if need else None
                  for g, need in zip(gp, ctx.needs_input_grad[6:])]
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
