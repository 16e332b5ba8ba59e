"""``Octree`` with the method surface of the reference's TorchScript class
``torch.classes.svo.Octree`` (third_party/sparse_octree/src/bindings.cpp:8-31),
backed by the flat-array octree of the C-ABI library (csrc/octree.cpp).

    tree = Octree(); tree.init(256, 16, 0.2); tree.insert(vox_int32_cpu)
    voxels, children, features = tree.get_centres_and_children()

Tensors are CPU tensors like the reference returns them.  Pickling replays the
inserted batches after resetting the process-global id counter, which is what
the reference's unpickle constructor does (octree.cpp:20-28)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib


def _vox(t):
    if isinstance(t, torch.Tensor):
        if t.dtype != torch.int32:
            # the reference's accessor<int,2>() throws on other dtypes
            raise RuntimeError('expected scalar type Int but found '
                               f'{t.dtype}')
        t = t.detach().cpu().contiguous().numpy()
    a = np.ascontiguousarray(t, dtype=np.int32)
    return a


class Octree:
    def __init__(self):
        self._h = None
        self._batches = []
        self._args = None

    def init(self, grid_dim: int, feat_dim: int, voxel_size: float):
        self._args = (int(grid_dim), int(feat_dim), float(voxel_size))
        self._h = _lib.lib().xrd_octree_create(*self._args)
        if not self._h:
            raise RuntimeError('xrd_octree_create failed')

    def __del__(self):
        try:
            if self._h:
                _lib.lib().xrd_octree_destroy(self._h)
        except Exception:
            pass

    def _need(self):
        if not self._h:
            print('Octree not initialized!')
            return False
        return True

    def insert(self, vox):
        if not self._need():
            return
        a = _vox(vox)
        if a.ndim != 2 or a.shape[1] != 3:
            print(f'Point dimensions mismatch: inputs are {a.shape[-1]} '
                  'expect 3')
            return
        created = C.c_int(0)
        _lib.check(_lib.lib().xrd_octree_insert(
            self._h, a.ctypes.data, a.shape[0], C.byref(created)),
            'xrd_octree_insert')
        if created.value:
            self._batches.append(a.copy())

    def try_insert(self, vox) -> float:
        a = _vox(vox)
        if a.ndim != 2 or a.shape[1] != 3:
            return -1.0
        return float(_lib.lib().xrd_octree_try_insert(self._h, a.ctypes.data,
                                                      a.shape[0]))

    def has_voxel(self, xyz) -> bool:
        a = _vox(xyz).reshape(-1)
        if a.shape[0] != 3:
            return False
        return bool(_lib.lib().xrd_octree_has_voxel(self._h, a.ctypes.data))

    def count_nodes(self) -> int:
        return int(_lib.lib().xrd_octree_count_nodes(self._h))

    def count_leaf_nodes(self) -> int:
        return int(_lib.lib().xrd_octree_count_leaf_nodes(self._h))

    def get_features(self, pts):
        return None  # empty body in the reference (octree.cpp:212-214)

    def get_voxels(self):
        n = _lib.lib().xrd_octree_get_voxels(self._h, None, 0)
        out = np.zeros((n, 4), np.float32)
        _lib.lib().xrd_octree_get_voxels(self._h, out.ctypes.data, n)
        return torch.from_numpy(out)

    def get_leaf_voxels(self):
        n = _lib.lib().xrd_octree_get_leaf_voxels(self._h, None, 0)
        out = np.zeros((n, 3), np.float32)
        _lib.lib().xrd_octree_get_leaf_voxels(self._h, out.ctypes.data, n)
        return torch.from_numpy(out)

    def get_centres_and_children(self):
        T = self.count_nodes()
        vox = np.empty((T, 4), np.float32)
        ch = np.empty((T, 8), np.float32)
        ft = np.empty((T, 8), np.int32)
        _lib.check(_lib.lib().xrd_octree_export(
            self._h, vox.ctypes.data, ch.ctypes.data, ft.ctypes.data),
            'xrd_octree_export')
        return torch.from_numpy(vox), torch.from_numpy(ch), \
            torch.from_numpy(ft)

    # pickling = (size, feat_dim, voxel_size, all_pts) like bindings.cpp:23-31
    def __getstate__(self):
        return {'args': self._args, 'batches': self._batches}

    def __setstate__(self, st):
        self._h, self._batches, self._args = None, [], st['args']
        _lib.lib().xrd_octree_reset_id_counter()
        self.init(*st['args'])
        for b in st['batches']:
            self.insert(b)


def reset_id_counter():
    _lib.lib().xrd_octree_reset_id_counter()
