"""``faiss``-shaped module for Point-SLAM's neighbour queries, backed by the
exact grid kNN (xrdslam_amd/engine/knn.py).  Only what
slam/model_components/neural_point_cloud.py:46-52,214-218,255 touches:
StandardGpuResources, IndexFlatL2, IndexIVFFlat, METRIC_L2, index_cpu_to_gpu,
and on the index: train / add / search(x, 8) / is_trained / ntotal / nprobe.

numpy in, numpy out like faiss; torch CUDA tensors are accepted too and then
returned as tensors (no host round trip).  Neighbours are exact within
``XRD_KNN_RADIUS`` (default 0.16 m = Point-SLAM's largest query radius): FAISS
IVF(nlist 400, nprobe 4) is approximate, so results are not comparable to the
reference's bit for bit."""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import _lib
from ..engine.knn import GridKNN

METRIC_L2 = 1


class StandardGpuResources:
    pass


class IndexFlatL2:
    def __init__(self, d):
        if d != 3:
            raise NotImplementedError('3-D points only')
        self.d = d


class IndexIVFFlat:
    def __init__(self, quantizer, d, nlist, metric=METRIC_L2):
        if d != 3 or metric != METRIC_L2:
            raise NotImplementedError('3-D points, L2 metric only')
        self.d, self.nlist, self.nprobe = d, nlist, 1
        self.is_trained = False
        self._knn = None
        self._device = 'cuda:0'

    def _index(self):
        if self._knn is None:
            self._knn = GridKNN(float(os.environ.get('XRD_KNN_RADIUS', 0.16)),
                                self._device)
        return self._knn

    @property
    def ntotal(self):
        return self._index().ntotal

    def train(self, x):
        self.is_trained = True  # nothing to learn: the grid is exact

    def add(self, x):
        t = x if torch.is_tensor(x) else torch.from_numpy(
            np.ascontiguousarray(x, np.float32))
        self._index().add(t)

    def search(self, x, k):
        as_tensor = torch.is_tensor(x)
        t = x if as_tensor else torch.from_numpy(
            np.ascontiguousarray(x, np.float32))
        D, I = self._index().search(t, k)
        if as_tensor:
            return D, I
        return D.cpu().numpy(), I.cpu().numpy()


def index_cpu_to_gpu(resource, device_id, index):
    index._device = f'cuda:{int(device_id)}'
    return index
