"""TEST INFRASTRUCTURE ONLY.  CPU stand-in for the ``faiss`` module as
Point-SLAM uses it (neural_point_cloud.py:46-52,214-218,255): EXACT brute-force
8-NN (squared L2, ascending; 3.4e38 / -1 padding when fewer points exist).  The
reference's IndexIVFFlat(400 lists, 4 probes) is approximate; the product's
grid kNN and this stand-in are both exact, so they agree with each other and
with FAISS whenever FAISS finds the true neighbours."""
import types

import numpy as np

METRIC_L2 = 1
# trajectory-level runs (oracle/make_golden_c1.py) switch to a k-d tree: the
# same exact neighbours (ties between equidistant points may come in another
# order), ~100 x faster than the dense distance matrix + full sort below,
# which stays the checker of the kernel-level goldens (bit-exact order)
FAST = False


class StandardGpuResources:
    pass


class IndexFlatL2:
    def __init__(self, d):
        self.d = d


class IndexIVFFlat:
    def __init__(self, quantizer, d, nlist, metric=METRIC_L2):
        self.pts = np.zeros((0, 3), np.float32)
        self.is_trained = False
        self.nprobe = 1

    @property
    def ntotal(self):
        return self.pts.shape[0]

    def train(self, x):
        self.is_trained = True

    def add(self, x):
        self.pts = np.concatenate([self.pts, np.asarray(x, np.float32)
                                   .reshape(-1, 3)], 0)

    def search(self, x, k):
        x = np.asarray(x, np.float32).reshape(-1, 3)
        m, n = x.shape[0], self.pts.shape[0]
        D = np.full((m, k), 3.4028235e38, np.float32)
        ids = np.full((m, k), -1, np.int64)
        if n == 0 or m == 0:
            return D, ids
        if FAST:
            from scipy.spatial import cKDTree
            if getattr(self, '_tree_n', -1) != n:
                self._tree, self._tree_n = cKDTree(self.pts.astype(
                    np.float64)), n
            kk = min(k, n)
            dd, ii = self._tree.query(x.astype(np.float64), k=kk, workers=-1)
            dd, ii = dd.reshape(m, kk), ii.reshape(m, kk)
            D[:, :kk] = (dd**2).astype(np.float32)
            ids[:, :kk] = ii
            return D, ids
        d2 = ((x[:, None, :].astype(np.float64) -
               self.pts[None, :, :].astype(np.float64))**2).sum(-1)
        order = np.argsort(d2, axis=1, kind='stable')[:, :k]
        kk = order.shape[1]
        D[:, :kk] = np.take_along_axis(d2, order, 1).astype(np.float32)
        ids[:, :kk] = order
        return D, ids


def index_cpu_to_gpu(resource, device_id, index):
    return index


def module():
    m = types.ModuleType('faiss')
    for k, v in dict(METRIC_L2=METRIC_L2,
                     StandardGpuResources=StandardGpuResources,
                     IndexFlatL2=IndexFlatL2, IndexIVFFlat=IndexIVFFlat,
                     index_cpu_to_gpu=index_cpu_to_gpu).items():
        setattr(m, k, v)
    return m


class TorchKNN:
    """same search with the interface of xrdslam_amd.engine.knn.GridKNN, for
    running the product's NeuralPointCloud mirror on the CPU in tests"""

    def __init__(self, device='cpu'):
        self.index = IndexIVFFlat(None, 3, 1)
        self.device = device

    @property
    def ntotal(self):
        return self.index.ntotal

    def add(self, pts):
        self.index.add(pts.detach().cpu().numpy())

    def search(self, q, k=8):
        import torch
        D, ids = self.index.search(q.detach().cpu().numpy(), k)
        return torch.from_numpy(D), torch.from_numpy(ids)
