"""``tinycudann``-compatible ``Encoding`` backed by the HIP kernels.

Interface the reference relies on (slam/model_components/encodings_coslam.py,
slam/models/joint_encoding.py:212-234,439-481): ``Encoding(n_input_dims,
encoding_config, dtype=torch.float)`` -> ``nn.Module`` with ``.n_output_dims``,
one flat fp32 ``params`` Parameter, ``forward(x[N,D] in [0,1]) -> [N,out]``,
differentiable with respect to the parameters AND the inputs (tracking needs
d/dx).  Supported otypes: ``HashGrid`` / ``Grid`` (Hash or Dense, Linear
interpolation) and ``OneBlob`` — the ones Co-SLAM's defaults use; the others
raise ``NotImplementedError``.  Inputs may be float64 (the reference hands over
the f64-normalised coordinates): they are cast to f32 like
tiny-cuda-nn does."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib


class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, enc):
        lib = _lib.lib()
        x32 = x.detach().float().contiguous()
        n = x32.shape[0]
        y = torch.empty(n, enc.n_output_dims, dtype=torch.float32,
                        device=x32.device)
        _lib.check(lib.xrd_hashgrid_fwd(
            enc.n_levels, enc._scales.ctypes.data, enc._res.ctypes.data,
            enc._sizes.ctypes.data, enc._offsets.ctypes.data, n,
            _lib.ptr(x32), _lib.ptr(params.detach()), _lib.ptr(y),
            _lib.stream_ptr(x32.device)), 'xrd_hashgrid_fwd')
        ctx.enc = enc
        ctx.x_dtype = x.dtype
        ctx.save_for_backward(x32, params)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.lib()
        x32, params = ctx.saved_tensors
        enc = ctx.enc
        n = x32.shape[0]
        dy = dy.float().contiguous()
        dparams = torch.zeros_like(params) if ctx.needs_input_grad[1] else None
        dx = torch.empty_like(x32) if ctx.needs_input_grad[0] else None
        _lib.check(lib.xrd_hashgrid_bwd(
            enc.n_levels, enc._scales.ctypes.data, enc._res.ctypes.data,
            enc._sizes.ctypes.data, enc._offsets.ctypes.data, n,
            _lib.ptr(x32), _lib.ptr(params.detach()), _lib.ptr(dy),
            _lib.ptr(dparams), _lib.ptr(dx), _lib.stream_ptr(x32.device)),
            'xrd_hashgrid_bwd')
        if dx is not None:
            dx = dx.to(ctx.x_dtype)
        return dx, dparams, None


class _OneBlobFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_bins):
        lib = _lib.lib()
        x32 = x.detach().float().contiguous()
        n, d = x32.shape
        y = torch.empty(n, d * n_bins, dtype=torch.float32, device=x32.device)
        _lib.check(lib.xrd_oneblob_fwd(n, d, n_bins, _lib.ptr(x32),
                                       _lib.ptr(y),
                                       _lib.stream_ptr(x32.device)),
                   'xrd_oneblob_fwd')
        ctx.n_bins, ctx.x_dtype = n_bins, x.dtype
        ctx.save_for_backward(x32)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.lib()
        x32, = ctx.saved_tensors
        n, d = x32.shape
        dx = torch.empty_like(x32)
        _lib.check(lib.xrd_oneblob_bwd(n, d, ctx.n_bins, _lib.ptr(x32),
                                       _lib.ptr(dy.float().contiguous()),
                                       _lib.ptr(dx),
                                       _lib.stream_ptr(x32.device)),
                   'xrd_oneblob_bwd')
        return dx.to(ctx.x_dtype), None


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, dtype=torch.float,
                 seed=1337):
        super().__init__()
        cfg = dict(encoding_config)
        self.n_input_dims = int(n_input_dims)
        self.otype = cfg['otype']
        self.dtype = dtype
        self.encoding_config = cfg
        if self.otype in ('HashGrid', 'Grid'):
            if self.n_input_dims != 3:
                raise NotImplementedError('grid encodings: 3-D inputs only')
            if cfg.get('interpolation', 'Linear') != 'Linear':
                raise NotImplementedError('only Linear interpolation')
            dense = cfg.get('type', 'Hash') == 'Dense'
            if cfg.get('n_features_per_level', 2) != 2:
                raise NotImplementedError('n_features_per_level must be 2')
            self.n_levels = int(cfg.get('n_levels', 16))
            L = self.n_levels
            self._scales = np.zeros(L, np.float32)
            self._res = np.zeros(L, np.uint32)
            self._sizes = np.zeros(L, np.uint32)
            self._offsets = np.zeros(L, np.uint32)
            total = C.c_uint32(0)
            _lib.check(_lib.lib().xrd_hashgrid_levels(
                L, int(cfg.get('base_resolution', 16)),
                float(cfg.get('per_level_scale', 2.0)),
                int(cfg.get('log2_hashmap_size', 19)), int(dense),
                self._scales.ctypes.data, self._res.ctypes.data,
                self._sizes.ctypes.data, self._offsets.ctypes.data,
                C.byref(total)), 'xrd_hashgrid_levels')
            self.n_output_dims = 2 * L
            # tiny-cuda-nn initialises grid parameters U(-1e-4, 1e-4)
            self.params = nn.Parameter(
                (torch.rand(int(total.value) * 2) * 2 - 1) * 1e-4)
        elif self.otype == 'OneBlob':
            self.n_bins = int(cfg.get('n_bins', 16))
            self.n_output_dims = self.n_input_dims * self.n_bins
            self.params = nn.Parameter(torch.zeros(0))
        else:
            raise NotImplementedError(
                f"tinycudann shim: otype '{self.otype}' is not built (Co-SLAM "
                'defaults use HashGrid and OneBlob)')

    def level_resolutions(self):
        return [int(r) for r in self._res]

    def forward(self, x):
        _require_device(x)
        if self.otype == 'OneBlob':
            return _OneBlobFn.apply(x, self.n_bins)
        return _HashGridFn.apply(x, self.params, self)


def _require_device(x):
    if not x.is_cuda:
        raise _lib.XrdError('tinycudann shim: CUDA tensors only '
                            '(no CPU fallback)')


class Network(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError(
            'tcnn.Network (FullyFusedMLP) is only used when tcnn_network=True '
            '(default False, joint_encoding.py:35); not built')
