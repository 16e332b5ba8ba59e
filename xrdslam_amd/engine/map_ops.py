"""Map maintenance between iterations on the device (csrc/map_ops.hip): row
compaction by one mask, first-occurrence voxel flags, Point-SLAM's dynamic
radii, point insertion and frustum mask.  Thin callers of the C ABI; the only
host read-back of each is the ONE size it returns."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib

_PINNED = {}


def _read_count(count):
    """device i32 -> int through a pinned word (one blocking copy)"""
    key = count.device
    host = _PINNED.get(key)
    if host is None:
        host = _PINNED[key] = torch.empty(1, dtype=torch.int32,
                                          pin_memory=True)
    host.copy_(count, non_blocking=False)
    return int(host[0])


def compact_rows(keep: torch.Tensor, arrays, want_count=True):
    """rows of every tensor in ``arrays`` (same leading size n, 4-byte
    elements, contiguous) where ``keep`` [n] is set, order preserved.
    -> (list of narrowed outputs, count).  The outputs are prefixes of buffers
    sized n (no second allocation, no second copy)."""
    lib = _lib.lib()
    dev = keep.device
    n = int(keep.shape[0])
    keep = keep.reshape(-1)
    if keep.dtype == torch.bool:
        keep = keep.view(torch.uint8)
    assert keep.dtype == torch.uint8 and keep.is_contiguous()
    srcs, dsts, words = [], [], []
    for a in arrays:
        assert a.shape[0] == n and a.element_size() == 4 and a.device == dev
        a = a.detach()
        a = a if a.is_contiguous() else a.contiguous()
        srcs.append(a)
        dsts.append(torch.empty_like(a))
        words.append(a.numel() // n if n else 1)
    k = len(srcs)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(max(int(lib.xrd_compact_ws_ints(n)), 1),
                     dtype=torch.int32, device=dev)
    src_p = (C.c_void_p * max(k, 1))(*[_lib.ptr(a) for a in srcs])
    dst_p = (C.c_void_p * max(k, 1))(*[_lib.ptr(a) for a in dsts])
    w_p = (C.c_int32 * max(k, 1))(*words)
    _lib.check(lib.xrd_compact_rows(
        n, _lib.ptr(keep), k, src_p, dst_p, w_p, _lib.ptr(ws),
        _lib.ptr(count), _lib.stream_ptr(dev)), 'xrd_compact_rows')
    if not want_count:
        return dsts, count
    c = _read_count(count)
    return [d.narrow(0, 0, c) for d in dsts], c


class _VoxelTable:
    """open-addressing scratch of xrd_voxel_first_flags, kept between calls"""
    bufs = {}

    @classmethod
    def get(cls, dev, n):
        size = 1 << max(int(2 * n - 1).bit_length(), 4)
        b = cls.bufs.get(dev)
        if b is None or b[0].numel() < size:
            b = (torch.empty(size, dtype=torch.int64, device=dev),
                 torch.empty(size, dtype=torch.int32, device=dev),
                 torch.empty(1, dtype=torch.int32, device=dev))
            cls.bufs[dev] = b
        return b, size


def voxel_first_flags(voxels: torch.Tensor):
    """voxels [n,3] i32 -> u8 [n]: 1 at the first occurrence of every distinct
    row, and the device error word (0 = fine)"""
    lib = _lib.lib()
    dev = voxels.device
    voxels = voxels.contiguous()
    assert voxels.dtype == torch.int32 and voxels.shape[1] == 3
    n = int(voxels.shape[0])
    (keys, rows, err), size = _VoxelTable.get(dev, max(n, 1))
    first = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.check(lib.xrd_voxel_first_flags(
        n, _lib.ptr(voxels), _lib.ptr(first), _lib.ptr(keys), _lib.ptr(rows),
        size, _lib.ptr(err), _lib.stream_ptr(dev)), 'xrd_voxel_first_flags')
    return first, err


def distinct_voxels(voxels: torch.Tensor):
    """distinct rows of voxels [n,3] i32 in first-occurrence order"""
    first, err = voxel_first_flags(voxels)
    (out, ), _ = compact_rows(first, [voxels.contiguous()])
    code = int(err.item())
    if code:
        raise _lib.XrdError(
            f'xrd_voxel_first_flags: device error {code} '
            '(1: voxel coordinate outside +-2^20, 2: table full)')
    return out


def point_dynamic_radius(rgb: torch.Tensor, thresh, add_max, add_min,
                         query_ratio):
    """rgb [H,W,3] f32 device -> (r_add, r_query) [H,W] f64"""
    lib = _lib.lib()
    assert rgb.dtype == torch.float32 and rgb.dim() == 3 and rgb.shape[2] == 3
    rgb = rgb.contiguous()
    H, W = int(rgb.shape[0]), int(rgb.shape[1])
    r_add = torch.empty(H, W, dtype=torch.float64, device=rgb.device)
    r_query = torch.empty_like(r_add)
    _lib.check(lib.xrd_point_dynamic_radius(
        H, W, _lib.ptr(rgb), float(thresh), float(add_max), float(add_min),
        float(query_ratio), _lib.ptr(r_add), _lib.ptr(r_query),
        _lib.stream_ptr(rgb.device)), 'xrd_point_dynamic_radius')
    return r_add, r_query


def point_sensor_points(rays_o, rays_d, depth):
    """o + d * depth, [n,3]"""
    lib = _lib.lib()
    o, d = rays_o.float().contiguous(), rays_d.float().contiguous()
    z = depth.float().reshape(-1).contiguous()
    n = int(o.shape[0])
    pts = torch.empty(n, 3, device=o.device)
    _lib.check(lib.xrd_point_sensor_points(
        n, _lib.ptr(o), _lib.ptr(d), _lib.ptr(z), _lib.ptr(pts),
        _lib.stream_ptr(o.device)), 'xrd_point_sensor_points')
    return pts


def point_insert(rays_o, rays_d, depth, color, pts_gt, n_within, lin,
                 fix_interval, near_end, far_end):
    """-> (kept sensor points [c,3], their colours * 255 [c,3], points along
    the kept rays [c * n_add, 3], c)"""
    lib = _lib.lib()
    dev = rays_o.device
    o, d = rays_o.float().contiguous(), rays_d.float().contiguous()
    z = depth.float().reshape(-1).contiguous()
    col = color.float().contiguous()
    lin = lin.float().contiguous()
    n, n_add = int(o.shape[0]), int(lin.numel())
    pos = torch.empty(n, 3, device=dev)
    rgb = torch.empty(n, 3, device=dev)
    pts = torch.empty(n * n_add, 3, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    if n_within is not None:
        n_within = n_within.int().contiguous()
        assert n_within.numel() == n
    _lib.check(lib.xrd_point_insert(
        n, _lib.ptr(o), _lib.ptr(d), _lib.ptr(z), _lib.ptr(col),
        _lib.ptr(pts_gt.contiguous()),
        None if n_within is None else _lib.ptr(n_within), _lib.ptr(lin),
        n_add, int(bool(fix_interval)), float(near_end), float(far_end),
        _lib.ptr(pos), _lib.ptr(rgb), _lib.ptr(pts), _lib.ptr(count),
        _lib.stream_ptr(dev)), 'xrd_point_insert')
    c = _read_count(count)
    return pos[:c], rgb[:c], pts[:c * n_add], c


def point_frustum_mask(points, w2c, depth, H, W, fx, fy, cx, cy, edge):
    """points [n,3] f32, w2c [4,4] / [3,4] f64 device, depth [H,W] f32 ->
    bool [n]"""
    lib = _lib.lib()
    dev = points.device
    pts = points.detach().float().contiguous()
    n = int(pts.shape[0])
    w = w2c.detach().to(dev, torch.float64)[:3].contiguous()
    img = depth.detach().reshape(H, W).float().contiguous()
    ws_f = torch.empty(max(n, 1) * 3, device=dev)
    ws_d = torch.empty(max(n, 1), dtype=torch.float64, device=dev)
    ws_i = torch.empty(1, dtype=torch.int32, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.check(lib.xrd_point_frustum_mask(
        n, _lib.ptr(pts), _lib.ptr(w), _lib.ptr(img), int(H), int(W),
        float(fx), float(fy), float(cx), float(cy), int(edge), _lib.ptr(ws_f),
        _lib.ptr(ws_d), _lib.ptr(ws_i), _lib.ptr(mask), _lib.stream_ptr(dev)),
        'xrd_point_frustum_mask')
    return mask.view(torch.bool)
