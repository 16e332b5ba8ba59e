"""Device-resident exact kNN index over a 3-D point cloud (uniform grid),
the engine behind the ``faiss`` shim (xrdslam_amd/compat/faiss.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib


PROFILE = None   # bench.py: [(start event, end event, queries, points)]


class GridKNN:
    """exact k=8 nearest neighbours within ``max_radius``"""

    def __init__(self, max_radius: float = 0.16, device='cuda:0'):
        self.max_radius = float(max_radius)
        self.device = torch.device(device)
        self.points = torch.zeros(0, 3, device=self.device)
        self._dirty = True

    @property
    def ntotal(self):
        return int(self.points.shape[0])

    def add(self, pts: torch.Tensor):
        pts = pts.detach().to(self.device, torch.float32).reshape(-1, 3)
        self.points = torch.cat([self.points, pts], 0)
        self._dirty = True

    def reset(self):
        self.points = torch.zeros(0, 3, device=self.device)
        self._dirty = True

    def _build(self):
        lib = _lib.lib()
        n = self.ntotal
        cell = self.max_radius
        lo = self.points.min(0).values - 1e-3
        hi = self.points.max(0).values + 1e-3
        self._origin = np.ascontiguousarray(lo.cpu().numpy(), np.float32)
        ext = (hi - lo).cpu().numpy()
        self._dims = np.ascontiguousarray(
            np.maximum(np.ceil(ext / cell), 1).astype(np.int32))
        ncell = int(np.prod(self._dims.astype(np.int64)))
        st = _lib.stream_ptr(self.device)
        cid = torch.empty(n, dtype=torch.int64, device=self.device)
        _lib.check(lib.xrd_knn_cell_ids(
            n, _lib.ptr(self.points), self._origin.ctypes.data, cell,
            self._dims.ctypes.data, _lib.ptr(cid), st), 'xrd_knn_cell_ids')
        cid_s, order = torch.sort(cid, stable=True)
        self._sorted_pts = self.points[order].contiguous()
        self._sorted_ids = order.int().contiguous()
        self._start = torch.zeros(ncell, dtype=torch.int32, device=self.device)
        self._end = torch.zeros(ncell, dtype=torch.int32, device=self.device)
        _lib.check(lib.xrd_knn_cell_ranges(
            n, _lib.ptr(cid_s), _lib.ptr(self._start), _lib.ptr(self._end),
            st), 'xrd_knn_cell_ranges')
        self._dirty = False

    def search(self, queries: torch.Tensor, k: int = 8):
        """-> (squared distances [m,k] f32 ascending, ids [m,k] i64; FLT_MAX/-1
        where fewer than k points lie within max_radius)"""
        return self.search_count(queries, k, None)[:2]

    def search_count(self, queries: torch.Tensor, k: int = 8, radius=None):
        """search + (radius: float or per-query tensor [m]) the number of
        neighbours strictly inside the radius, int32 [m] (None without a
        radius) — one launch"""
        lib = _lib.lib()
        q = queries.detach().to(self.device, torch.float32).reshape(
            -1, 3).contiguous()
        m = q.shape[0]
        if self.ntotal == 0 or m == 0:
            D = torch.full((m, k), torch.finfo(torch.float32).max,
                           device=self.device)
            I = torch.full((m, k), -1, dtype=torch.int64, device=self.device)
            cnt = None if radius is None else torch.zeros(
                m, dtype=torch.int32, device=self.device)
            return D, I, cnt
        # the kernel writes every entry (FLT_MAX / -1 where there is none)
        D = torch.empty(m, k, dtype=torch.float32, device=self.device)
        I = torch.empty(m, k, dtype=torch.int64, device=self.device)
        cnt = rq = None
        r_all = 0.0
        if radius is not None:
            cnt = torch.empty(m, dtype=torch.int32, device=self.device)
            if torch.is_tensor(radius):
                rq = radius.detach().to(self.device, torch.float32).reshape(
                    -1).contiguous()
                assert rq.numel() == m
            else:
                r_all = float(radius)
        if self._dirty:
            self._build()
        if PROFILE is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        # (the count output is not optional in the kernel; the scratch keeps a
        # name so that it outlives the call)
        within = cnt if cnt is not None else torch.empty(
            m, dtype=torch.int32, device=self.device)
        _lib.check(lib.xrd_knn_search_count(
            m, _lib.ptr(q), _lib.ptr(self._sorted_pts),
            _lib.ptr(self._sorted_ids), self._origin.ctypes.data,
            self.max_radius, self._dims.ctypes.data, _lib.ptr(self._start),
            _lib.ptr(self._end), k, self.max_radius, _lib.ptr(D), _lib.ptr(I),
            _lib.ptr(rq), r_all, _lib.ptr(within),
            _lib.stream_ptr(self.device)), 'xrd_knn_search_count')
        if PROFILE is not None:
            e1.record()
            PROFILE.append((e0, e1, m, self.ntotal))
        return D, I, cnt
