"""GPU parity of the grid kNN (faiss shim) against exact brute force (torch):
ids exact (ties -> smaller id), squared distances within 1e-6; neighbours
beyond the radius are reported as (FLT_MAX, -1)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FMAX = torch.finfo(torch.float32).max


def brute(points, queries, k, radius):
    d2 = ((queries[:, None, :].double() - points[None].double())**2).sum(-1)
    d2 = d2.float()
    d2 = torch.where(d2 > radius * radius, torch.full_like(d2, FMAX), d2)
    # sort by (distance, id): stable sort on distance keeps ids ascending
    vals, idx = torch.sort(d2, dim=1, stable=True)
    vals, idx = vals[:, :k], idx[:, :k]
    if vals.shape[1] < k:  # fewer points than k: pad like the index does
        pad = k - vals.shape[1]
        vals = torch.cat([vals, torch.full((vals.shape[0], pad), FMAX)], 1)
        idx = torch.cat([idx, torch.full((idx.shape[0], pad), -1)], 1)
    idx = torch.where(vals == FMAX, torch.full_like(idx, -1), idx)
    return vals, idx


@pytest.mark.parametrize('n,m', [(5000, 3000), (7, 50), (1, 4)])
def test_knn_matches_brute_force(n, m):
    from xrdslam_amd.compat import faiss
    g = torch.Generator().manual_seed(n)
    pts = torch.rand(n, 3, generator=g) * torch.tensor([2.0, 1.5, 1.0])
    pts[::7] = pts[::7].round(decimals=1)  # duplicates / exact ties
    q = torch.rand(m, 3, generator=g) * torch.tensor([2.2, 1.6, 1.1]) - 0.05
    index = faiss.index_cpu_to_gpu(
        faiss.StandardGpuResources(), 0,
        faiss.IndexIVFFlat(faiss.IndexFlatL2(3), 3, 400, faiss.METRIC_L2))
    index.nprobe = 4
    assert not index.is_trained
    index.train(pts.numpy())
    index.add(pts[:n // 2].numpy())
    index.add(pts[n // 2:].numpy())
    assert index.ntotal == n and index.is_trained
    D, I = index.search(q.numpy(), 8)
    rd, ri = brute(pts, q, 8, 0.16)
    assert D.shape == (m, 8) and I.dtype == np.int64
    # float32 kernel vs float64 brute force: distances 1e-6, ids may swap only
    # between numerically tied candidates
    found = ri >= 0
    assert np.array_equal(I >= 0, found.numpy())
    assert np.allclose(D[found.numpy()], rd[found].numpy(), rtol=1e-5,
                       atol=1e-9)
    same = (torch.from_numpy(I) == ri)
    if not bool(same.all()):
        bad = ~same
        # every mismatch must be a distance tie at float32 resolution
        dd = torch.from_numpy(D)[bad]
        assert torch.allclose(dd, rd[bad], rtol=1e-5, atol=1e-9)
    assert float(same.float().mean()) > 0.999
    # tensors in -> tensors out, no host copy
    Dt, It = index.search(q.cuda(), 8)
    assert Dt.is_cuda and torch.equal(It.cpu(), torch.from_numpy(I))


def test_knn_empty_index_and_far_queries():
    from xrdslam_amd.engine.knn import GridKNN
    idx = GridKNN(0.16)
    D, I = idx.search(torch.zeros(3, 3, device='cuda'))
    assert (I == -1).all()
    idx.add(torch.zeros(1, 3))
    D, I = idx.search(torch.tensor([[5.0, 5.0, 5.0], [0.05, 0.0, 0.0]]))
    assert I[0].tolist() == [-1] * 8 and I[1, 0] == 0 and (I[1, 1:] == -1).all()
    assert abs(float(D[1, 0]) - 0.0025) < 1e-7


def test_search_count_equals_the_torch_count():
    """xrd_knn_search_count: the neighbours strictly inside the query's own
    radius, counted in the search launch = (D < r^2).sum(-1) of
    NeuralPointCloud.find_neighbors_faiss"""
    import torch
    from xrdslam_amd.engine.knn import GridKNN
    g = torch.Generator().manual_seed(2)
    pts = torch.rand(20000, 3, generator=g) * 2
    q = torch.rand(5000, 3, generator=g) * 2
    knn = GridKNN(0.16, 'cuda:0')
    knn.add(pts.cuda())
    D, I = knn.search(q.cuda(), 8)
    r = (0.02 + 0.1 * torch.rand(5000, generator=g)).cuda()
    D2, I2, c = knn.search_count(q.cuda(), 8, r)
    assert torch.equal(D, D2) and torch.equal(I, I2)
    assert torch.equal(c, (D < r.reshape(-1, 1)**2).sum(-1).int())
    _, _, c2 = knn.search_count(q.cuda(), 8, 0.08)
    assert torch.equal(c2, (D < 0.08**2).sum(-1).int())
