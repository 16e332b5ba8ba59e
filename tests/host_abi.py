"""TEST INFRASTRUCTURE ONLY — never imported by the product path.

A host-side stand-in for the C-ABI library object: the entry points the
``compat`` shims call are evaluated on HOST pointers by the oracles
(oracle/tcnn_oracle.py; oracle/svo_oracle.c through tests/svo_util.py), every
other symbol (level tables, sizes, the host C++ octree) goes to the real
``libxrdslam_hip.so``.  With it the shims run unchanged on CPU tensors, so the
REFERENCE's own model code can be executed end to end on top of
``xrdslam_amd.compat`` in the build container: what is exercised is the shim
modules and the call protocol of the boundary (argument order, dtypes, who
allocates what, accumulate-or-overwrite) — the kernels behind the same entry
points are what the ``-m gpu`` suites check against the same goldens.

    with host_abi.installed():   # patches xrdslam_amd._lib.lib / stream_ptr
        ...
"""
import contextlib
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(os.path.dirname(HERE), 'oracle'), HERE]
import svo_util  # noqa: E402
import tcnn_oracle as to  # noqa: E402


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, int):
        return p
    return int(p.value or 0)


def _view(p, shape, ctype=C.c_float):
    """numpy view of host memory at pointer ``p`` (0 elements: empty array)"""
    n = int(np.prod(shape))
    dt = np.dtype(ctype)
    if n == 0:
        return np.zeros(shape, dt)
    a = _addr(p)
    assert a, 'null pointer for a non-empty array'
    return np.ctypeslib.as_array(C.cast(a, C.POINTER(ctype)),
                                 shape=(n, )).reshape(shape)


def _levels(n_levels, scales, res, sizes, offsets):
    sc = _view(scales, (n_levels, ))
    rs, sz, of = (_view(p, (n_levels, ), C.c_uint32)
                  for p in (res, sizes, offsets))
    lv = [(float(a), int(b), int(c), int(d))
          for a, b, c, d in zip(sc, rs, sz, of)]
    return lv, int(of[-1]) + int(sz[-1])


class HostAbi:
    """proxy of the ctypes library: oracle-backed compute entry points"""

    def __init__(self, real):
        self._real = real
        self.calls = []          # names of the compute entry points used

    def __getattr__(self, name):
        return getattr(self._real, name)

    # -- tiny-cuda-nn encodings ------------------------------------------------
    def xrd_hashgrid_fwd(self, n_levels, scales, res, sizes, offsets, n, x,
                         params, y, stream):
        self.calls.append('xrd_hashgrid_fwd')
        lv, total = _levels(n_levels, scales, res, sizes, offsets)
        xt = torch.from_numpy(_view(x, (n, 3)))
        pt = torch.from_numpy(_view(params, (total * 2, )))
        _view(y, (n, 2 * n_levels))[...] = \
            to.hashgrid_forward(xt, pt, lv).numpy()
        return 0

    def xrd_hashgrid_bwd(self, n_levels, scales, res, sizes, offsets, n, x,
                         params, dy, dparams, dx, stream):
        self.calls.append('xrd_hashgrid_bwd')
        lv, total = _levels(n_levels, scales, res, sizes, offsets)
        want_p, want_x = bool(_addr(dparams)), bool(_addr(dx))
        xt = torch.from_numpy(_view(x, (n, 3)).copy()).requires_grad_(want_x)
        pt = torch.from_numpy(_view(params, (total * 2, )).copy()) \
            .requires_grad_(want_p)
        g = torch.from_numpy(_view(dy, (n, 2 * n_levels)).copy())
        with torch.enable_grad():
            to.hashgrid_forward(xt, pt, lv).backward(g)
        if want_p:      # the header: dparams is ACCUMULATED into
            _view(dparams, (total * 2, ))[...] += pt.grad.numpy()
        if want_x:
            _view(dx, (n, 3))[...] = xt.grad.numpy()
        return 0

    def xrd_oneblob_fwd(self, n, dims, n_bins, x, y, stream):
        self.calls.append('xrd_oneblob_fwd')
        xt = torch.from_numpy(_view(x, (n, dims)))
        _view(y, (n, dims * n_bins))[...] = \
            to.oneblob_forward(xt, n_bins).numpy()
        return 0

    def xrd_oneblob_bwd(self, n, dims, n_bins, x, dy, dx, stream):
        self.calls.append('xrd_oneblob_bwd')
        xt = torch.from_numpy(_view(x, (n, dims)).copy()).requires_grad_(True)
        g = torch.from_numpy(_view(dy, (n, dims * n_bins)).copy())
        with torch.enable_grad():
            to.oneblob_forward(xt, n_bins).backward(g)
        _view(dx, (n, dims))[...] = xt.grad.numpy()
        return 0

    # -- sparse_voxels `grid` kernels ------------------------------------------
    def xrd_svo_intersect(self, B, N, M, voxelsize, n_max, shared, ray_start,
                          ray_dir, points, children, idx, mn, mx, hits,
                          stream):
        self.calls.append('xrd_svo_intersect')
        pshape = (N, 3) if shared else (B, N, 3)
        cshape = (N, 9) if shared else (B, N, 9)
        pts = _view(points, pshape)
        ch = _view(children, cshape, C.c_int32)
        if shared:      # one octree for every batch row
            pts = np.repeat(pts[None], B, axis=0)
            ch = np.repeat(ch[None], B, axis=0)
        o_idx, o_mn, o_mx, _ = svo_util.svo_intersect_oracle(
            np.ascontiguousarray(_view(ray_start, (B, M, 3))),
            np.ascontiguousarray(_view(ray_dir, (B, M, 3))),
            np.ascontiguousarray(pts), np.ascontiguousarray(ch),
            float(voxelsize), int(n_max))
        _view(idx, (B, M, n_max), C.c_int32)[...] = o_idx
        _view(mn, (B, M, n_max))[...] = o_mn
        _view(mx, (B, M, n_max))[...] = o_mx
        return 0

    def xrd_inverse_cdf_sampling(self, G, R, P, S, fixed_step, pts_idx,
                                 min_depth, max_depth, noise, probs, steps,
                                 sidx, sdep, sdis, stream):
        self.calls.append('xrd_inverse_cdf_sampling')
        a = [np.ascontiguousarray(_view(pts_idx, (G, R, P), C.c_int32)),
             np.ascontiguousarray(_view(min_depth, (G, R, P))),
             np.ascontiguousarray(_view(max_depth, (G, R, P))),
             np.ascontiguousarray(_view(noise, (G, R, S))),
             np.ascontiguousarray(_view(probs, (G, R, P))),
             np.ascontiguousarray(_view(steps, (G, R)))]
        o_idx, o_dep, o_dis = svo_util.inverse_cdf_oracle(*a,
                                                          float(fixed_step))
        _view(sidx, (G, R, S), C.c_int32)[...] = o_idx
        _view(sdep, (G, R, S))[...] = o_dep
        _view(sdis, (G, R, S))[...] = o_dis
        return 0


    # -- Point-SLAM neighbour search (exact, brute force behind the grid API) ---
    def xrd_knn_cell_ids(self, n, points, origin, cell, dims, cell_ids,
                         stream):
        self.calls.append('xrd_knn_cell_ids')
        p = _view(points, (n, 3))
        o = _view(origin, (3, ))
        d = _view(dims, (3, ), C.c_int32)
        inv = np.float32(1.0) / np.float32(cell)
        c = np.floor((p - o[None]) * inv).astype(np.int64)
        c = np.clip(c, 0, d[None].astype(np.int64) - 1)
        _view(cell_ids, (n, ), C.c_int64)[...] = \
            (c[:, 2] * d[1] + c[:, 1]) * d[0] + c[:, 0]
        return 0

    def xrd_knn_cell_ranges(self, n, sorted_ids, start, end, stream):
        self.calls.append('xrd_knn_cell_ranges')
        ids = _view(sorted_ids, (n, ), C.c_int64)
        # (the caller's arrays are pre-zeroed and sized by the grid: only the
        # occupied cells are written, like the kernel)
        first = np.flatnonzero(np.r_[True, ids[1:] != ids[:-1]])
        last = np.r_[first[1:], n]
        for c, a, b in zip(ids[first], first, last):
            C.cast(_addr(start), C.POINTER(C.c_int32))[int(c)] = int(a)
            C.cast(_addr(end), C.POINTER(C.c_int32))[int(c)] = int(b)
        return 0

    def xrd_knn_search_count(self, m, queries, sorted_points, sorted_ids,
                             origin, cell, dims, cell_start, cell_end, k,
                             max_radius, out_d2, out_idx, radius_q, radius_all,
                             n_within, stream):
        self.calls.append('xrd_knn_search_count')
        d = _view(dims, (3, ), C.c_int32)
        ncell = int(d[0]) * int(d[1]) * int(d[2])
        n = int(_view(cell_end, (ncell, ), C.c_int32).max())
        q = _view(queries, (m, 3))
        pts = _view(sorted_points, (n, 3))
        ids = _view(sorted_ids, (n, ), C.c_int32).astype(np.int64)
        D = np.full((m, k), np.finfo(np.float32).max, np.float32)
        I = np.full((m, k), -1, np.int64)
        diff = q[:, None, :] - pts[None, :, :]
        d2 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + \
            diff[..., 2] * diff[..., 2]
        lim = np.float32(max_radius) * np.float32(max_radius)
        for r in range(m):
            ok = np.flatnonzero(d2[r] <= lim)
            order = ok[np.lexsort((ids[ok], d2[r][ok]))][:k]
            D[r, :order.size] = d2[r][order]
            I[r, :order.size] = ids[order]
        _view(out_d2, (m, k))[...] = D
        _view(out_idx, (m, k), C.c_int64)[...] = I
        if _addr(n_within):
            rq = _view(radius_q, (m, )) if _addr(radius_q) else \
                np.full(m, np.float32(radius_all), np.float32)
            _view(n_within, (m, ), C.c_int32)[...] = \
                ((D < (rq * rq)[:, None]) & (I >= 0)).sum(1)
        return 0

    def xrd_knn_search(self, m, queries, sorted_points, sorted_ids, origin,
                       cell, dims, cell_start, cell_end, k, max_radius, out_d2,
                       out_idx, stream):
        return self.xrd_knn_search_count(
            m, queries, sorted_points, sorted_ids, origin, cell, dims,
            cell_start, cell_end, k, max_radius, out_d2, out_idx, None, 0.0,
            None, stream)


@contextlib.contextmanager
def installed():
    """route xrdslam_amd._lib through the host backend (and drop the stream
    argument: there is no HIP stream on this box).  The product shims refuse
    host tensors unconditionally; every switch that lets the reference's model
    code reach this backend with host tensors is patched HERE, for the
    duration of the block, and nowhere inside ``xrdslam_amd/``."""
    from xrdslam_amd import _lib
    from xrdslam_amd.compat import faiss as c_faiss
    from xrdslam_amd.compat import grid as c_grid
    from xrdslam_amd.compat import tinycudann as c_tcnn
    real_lib, real_stream = _lib.lib, _lib.stream_ptr
    real = (c_grid._require_device, c_tcnn._require_device,
            c_faiss.index_cpu_to_gpu)
    proxy = HostAbi(real_lib())
    _lib.lib = lambda: proxy
    _lib.stream_ptr = lambda device=None: None
    c_grid._require_device = lambda t, name: None
    c_tcnn._require_device = lambda x: None

    def host_index(resource, device_id, index):
        index._device = 'cpu'
        return index
    c_faiss.index_cpu_to_gpu = host_index
    try:
        yield proxy
    finally:
        _lib.lib, _lib.stream_ptr = real_lib, real_stream
        (c_grid._require_device, c_tcnn._require_device,
         c_faiss.index_cpu_to_gpu) = real
