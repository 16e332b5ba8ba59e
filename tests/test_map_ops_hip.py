"""Map-maintenance kernels (csrc/map_ops.hip) against the
torch / numpy formulations they replace — which are the code paths pinned to
the reference on the CPU (tests/test_reference_host_parity.py,
tests/test_pointslam_host.py, tests/test_splatam_host.py).  Index and byte
work is compared bit for bit; the f64 radii to the last bits."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _note(line):
    path = os.environ.get('XRD_PARITY_REPORT')
    if path:
        with open(path, 'a') as f:
            f.write(line + '\n')


# ------------------------------------------------------------------ compaction
@pytest.mark.parametrize('n', [0, 1, 63, 1023, 1024, 1025, 5000, 307200])
@pytest.mark.parametrize('density', [0.0, 0.37, 1.0])
def test_compact_rows_equals_boolean_indexing(n, density):
    from xrdslam_amd.engine.map_ops import compact_rows
    g = torch.Generator().manual_seed(n + int(100 * density))
    keep = (torch.rand(n, generator=g) < density).to(DEV)
    arrays = [torch.randn(n, 3, generator=g).to(DEV),
              torch.randn(n, generator=g).to(DEV),
              torch.randint(-9, 9, (n, 4), generator=g, dtype=torch.int32)
              .to(DEV),
              torch.randn(n, 1, generator=g).to(DEV),
              torch.randn(n, 3, 3, generator=g).to(DEV)]
    out, count = compact_rows(keep, arrays)
    assert count == int(keep.sum())
    for o, a in zip(out, arrays):
        assert o.dtype == a.dtype and o.shape[1:] == a.shape[1:]
        assert torch.equal(o, a[keep])


def test_compact_rows_count_only_and_bad_arguments():
    from xrdslam_amd import _lib
    from xrdslam_amd.engine.map_ops import compact_rows
    keep = (torch.arange(4099, device=DEV) % 3 == 0)
    out, count = compact_rows(keep, [])
    assert out == [] and count == 1367
    lib = _lib.lib()
    assert lib.xrd_compact_rows(8, _lib.ptr(keep.view(torch.uint8)), 25, None,
                                None, None, None, _lib.ptr(keep), None) == 1


def test_splatam_remove_points_kernel_equals_index_select():
    """GaussianCloud.remove_points on the GPU (xrd_compact_rows) against the
    index_select formulation of the CPU path: parameters, Adam moments,
    statistics, and the re-keying of the optimiser state"""
    from xrdslam_amd.slam.model_components.gaussian_cloud_splatam import \
        GaussianCloud
    n = 20011
    g = torch.Generator().manual_seed(5)
    shapes = {'means3D': 3, 'rgb_colors': 3, 'unnorm_rotations': 4,
              'logit_opacities': 1, 'log_scales': 1}
    base = {k: torch.randn(n, w, generator=g) for k, w in shapes.items()}
    moments = {k: (torch.randn(n, w, generator=g),
                   torch.rand(n, w, generator=g)) for k, w in shapes.items()}
    stats = {k: torch.rand(n, generator=g) for k in
             ('means2D_gradient_accum', 'denom', 'max_2D_radius', 'timestep')}
    remove = torch.rand(n, generator=g) < 0.23
    got = {}
    for dev in ('cpu', DEV):
        cloud = GaussianCloud.__new__(GaussianCloud)
        torch.nn.Module.__init__(cloud)
        cloud.device = dev
        cloud.params = {k: torch.nn.Parameter(v.clone().to(dev))
                        for k, v in base.items()}
        cloud.variables = {k: v.clone().to(dev) for k, v in stats.items()}
        opt = {}
        for k, p in cloud.params.items():
            o = torch.optim.Adam([p], lr=1e-3)
            o.state[p] = {'step': torch.tensor(3.),
                          'exp_avg': moments[k][0].clone().to(dev),
                          'exp_avg_sq': moments[k][1].clone().to(dev)}
            opt[k] = o
        cloud.remove_points(remove.to(dev), opt)
        got[dev] = (cloud, opt)
    (c_cpu, o_cpu), (c_gpu, o_gpu) = got['cpu'], got[DEV]
    kept = int((~remove).sum())
    for k in shapes:
        assert c_gpu.params[k].shape[0] == kept
        assert torch.equal(c_gpu.params[k].detach().cpu(),
                           c_cpu.params[k].detach())
        assert c_gpu.params[k] is o_gpu[k].param_groups[0]['params'][0]
        assert c_gpu.params[k].requires_grad
        # the sliced moments stay keyed to the OLD parameter object
        (s_gpu, ), (s_cpu, ) = o_gpu[k].state.values(), o_cpu[k].state.values()
        for m in ('exp_avg', 'exp_avg_sq'):
            assert torch.equal(s_gpu[m].cpu(), s_cpu[m])
        assert c_gpu.params[k] not in o_gpu[k].state
    for k in stats:
        assert torch.equal(c_gpu.variables[k].cpu(), c_cpu.variables[k])


# --------------------------------------------------------------- voxel dedup
def _voxel_rows(seed, n, spread):
    """image-like: runs of equal voxels with revisits, negative coordinates"""
    g = torch.Generator().manual_seed(seed)
    base = torch.randint(-spread, spread, (max(n // 7, 1), 3), generator=g,
                         dtype=torch.int32)
    pick = torch.randint(0, base.shape[0], (n, ), generator=g)
    if n % 64 == 0:     # runs of equal rows like neighbouring depth pixels
        pick = torch.sort(pick.reshape(-1, 64), dim=1).values.reshape(-1)
    return base[pick]


@pytest.mark.parametrize('n,spread', [(1, 3), (64, 2), (1000, 4),
                                      (307200, 12), (307200, 400)])
def test_distinct_voxels_equal_unique_in_first_occurrence_order(n, spread):
    from xrdslam_amd.engine.map_ops import distinct_voxels, voxel_first_flags
    from xrdslam_amd.slam.models.sparse_voxel import SparseVoxel
    vox = _voxel_rows(n + spread, n, spread).to(DEV)
    want = SparseVoxel.distinct_voxels_torch(vox)
    got = distinct_voxels(vox)
    assert got.dtype == torch.int32 and torch.equal(got, want)
    first, err = voxel_first_flags(vox)
    assert int(err) == 0 and int(first.sum()) == want.shape[0]


def test_distinct_voxels_reports_out_of_range_coordinates():
    from xrdslam_amd import _lib
    from xrdslam_amd.engine.map_ops import distinct_voxels
    vox = torch.tensor([[0, 0, 0], [1 << 20, 0, 0]], dtype=torch.int32,
                       device=DEV)
    with pytest.raises(_lib.XrdError):
        distinct_voxels(vox)


def test_sparse_voxel_insert_points_same_octree_as_torch_dedup():
    """SparseVoxel.insert_points through the kernels and through
    torch.unique build the same octree (node order included)"""
    from xrdslam_amd.compat import svo as _svo
    from xrdslam_amd.engine.map_ops import distinct_voxels
    from xrdslam_amd.slam.models.sparse_voxel import SparseVoxel
    g = torch.Generator().manual_seed(9)
    pts = (torch.rand(50000, 3, generator=g) * 2 - 1).to(DEV) * \
        torch.tensor([2.0, 1.5, 1.0], device=DEV)
    vox = torch.div(pts, 0.2, rounding_mode='floor').int()
    trees = []
    for rows in (distinct_voxels(vox), SparseVoxel.distinct_voxels_torch(vox)):
        _svo.reset_id_counter()    # node ids come from a process-wide counter
        tree = _svo.Octree()
        tree.init(256, 16, 0.2)
        tree.insert(rows.cpu().int())
        trees.append(tree.get_centres_and_children())
    for a, b in zip(*trees):
        assert torch.equal(a, b)


# ------------------------------------------------------------- dynamic radii
def _radius_self(dev):
    from xrdslam_amd.slam.algorithms.point_slam import PointSLAM
    algo = PointSLAM.__new__(PointSLAM)
    algo.model = types.SimpleNamespace(device=dev)
    algo.config = types.SimpleNamespace(
        pointcloud_radius_query_ratio=2.0,
        pointcloud_color_grad_threshold=0.15, pointcloud_radius_add_max=0.08,
        pointcloud_radius_add_min=0.02, mapping_frustum_edge=-4)
    return algo


@pytest.mark.parametrize('kind', ['noise', 'smooth', 'flat', 'edges'])
def test_dynamic_radius_kernel_equals_host_numpy(kind):
    """xrd_point_dynamic_radius against cal_dynamic_radius' numpy path (the
    one pinned to the reference): same f64 arithmetic in the same order —
    equal to the last bits (a contracted multiply-add inside np.interp's
    compiled loop would be the only source of a 1-ulp difference)"""
    from xrdslam_amd.slam.algorithms.point_slam import PointSLAM
    from xrdslam_amd.slam.common.frame import Frame
    H, W = 480, 640
    g = torch.Generator().manual_seed(3)
    if kind == 'noise':
        img = torch.rand(H, W, 3, generator=g) * 0.2
    elif kind == 'smooth':
        y, x = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W),
                              indexing='ij')
        img = torch.stack((x * y, 0.5 + 0.3 * torch.sin(7 * x), y), -1)
    elif kind == 'flat':
        img = torch.full((H, W, 3), 0.25)
    else:
        img = (torch.rand(H // 8, W // 8, 3, generator=g) > 0.5).float() \
            .repeat_interleave(8, 0).repeat_interleave(8, 1)
    img = img.numpy().astype(np.float32)
    algo = _radius_self(DEV)
    frame = Frame(0, img, np.ones((H, W), np.float32))
    ka, kq = PointSLAM.cal_dynamic_radius(algo, img, frame=frame)
    ha, hq = PointSLAM.cal_dynamic_radius_host(algo, img)
    assert ka.dtype == torch.float64 and ka.shape == (H, W) and ka.is_cuda
    for name, k, h in (('r_add', ka, ha), ('r_query', kq, hq)):
        rel = float(((k - h).abs() / h.abs()).max())
        _note(f'map_ops/dynamic_radius/{kind}/{name}\trel_max={rel:.3e}\t'
              f'bit_equal={bool(torch.equal(k, h))}')
        assert rel < 1e-15
    if kind == 'flat':
        assert float(ka.min()) == 0.08 and float(kq.max()) == 0.16
    if kind == 'edges':
        assert float(ka.min()) == 0.02 and float(ka.max()) == 0.08
    # computed once per frame
    assert PointSLAM.cal_dynamic_radius(algo, img, frame=frame)[0] is ka


# ----------------------------------------------------------- point insertion
def _cloud(dev, fix_interval, device_insert):
    from xrdslam_amd.slam.model_components.neural_point_cloud import \
        NeuralPointCloud
    npc = NeuralPointCloud(
        c_dim=8, nn_num=8, radius_add=0.04, cuda_id=0, radius_min=0.02,
        radius_query=0.08, fix_interval_when_add_along_ray=fix_interval,
        use_dynamic_radius=True, N_surface=5, N_add=3,
        near_end_surface=0.98, far_end_surface=1.02, device=dev)
    npc.device_insert = device_insert
    gen = torch.Generator().manual_seed(11)
    npc.feature_init_fn = lambda n, c: torch.zeros(n, c).normal_(
        0, 0.1, generator=gen)
    return npc


def _ray_batch(seed, n, with_holes):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(1, 3, generator=g).repeat(n, 1) * 0.1
    d = torch.nn.functional.normalize(
        torch.randn(n, 3, generator=g) * torch.tensor([0.4, 0.3, 1.0]), dim=1)
    depth = 1.0 + 2.0 * torch.rand(n, generator=g)
    if with_holes:
        depth[torch.rand(n, generator=g) < 0.1] = 0.0
    color = torch.rand(n, 3, generator=g)
    radius = (0.02 + 0.06 * torch.rand(n, generator=g)).double()
    return [t.to(DEV) for t in (o, d, depth, color, radius)]


@pytest.mark.parametrize('fix_interval', [False, True])
def test_add_neural_points_kernels_equal_torch_statements(fix_interval):
    """three frames' worth of insertions into two clouds — one through
    xrd_point_sensor_points / xrd_knn_search_count / xrd_point_insert, one
    through the reference's statement sequence in torch ops — stay identical:
    sensor points, colours, neural points (bit for bit), counts, features"""
    a = _cloud(DEV, fix_interval, True)
    b = _cloud(DEV, fix_interval, False)
    first = _ray_batch(20, 6000, True)
    # the second batch revisits half of the first one's rays (their space is
    # occupied now) next to new ones
    fresh = _ray_batch(21, 3000, False)
    again = [torch.cat((f[:1500], t[:1500])) for f, t in zip(first, fresh)]
    again[2] = again[2].clamp_min(0.5)          # (all valid: per-ray radii)
    calls = [(first, False, False), (again, True, False),
             (_ray_batch(22, 1500, False), True, True),
             (_ray_batch(23, 2500, True), False, True)]
    kept = []
    for (o, d, depth, color, radius), dyn, grad in calls:
        rets = [c.add_neural_points(o, d, depth, color, is_pts_grad=grad,
                                    dynamic_radius=radius if dyn else None)
                for c in (a, b)]
        assert int(rets[0]) == int(rets[1])
        kept.append(int(rets[0]))
        assert a.pts_num() == b.pts_num() and a.index_ntotal() == \
            b.index_ntotal()
        assert torch.equal(a._cloud, b._cloud)
        assert torch.equal(a._input_pos, b._input_pos)
        assert torch.equal(a._input_rgb, b._input_rgb)
        assert torch.equal(a.geo_feats, b.geo_feats)
    assert kept[0] == int((first[2] > 0).sum())  # empty cloud: every valid ray
    assert 1000 < kept[1] < 2500                # revisited rays were dropped
    assert a.pts_num() == 3 * sum(kept)


# -------------------------------------------------------------- frustum mask
def test_point_frustum_mask_kernel_equals_torch_formulation():
    """xrd_point_frustum_mask against get_mask_from_c2w_torch (pinned to the
    reference's cv2.remap formulation on the CPU).  The f64 camera transform
    is evaluated in a fixed order here and by a library GEMM there: a point
    within an ulp of an image or depth border could flip — none does on these
    20 000 points."""
    from xrdslam_amd.slam.algorithms.point_slam import PointSLAM
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.utils.opt_pose import quaternion_to_matrix
    H, W = 480, 640
    g = torch.Generator().manual_seed(17)
    algo = _radius_self(DEV)
    algo.camera = Camera(600., 601., 319.5, 239.5, W, H)
    pts = torch.randn(20000, 3, generator=g) * torch.tensor([2.5, 2.0, 2.5])
    algo.model.neural_point_cloud = types.SimpleNamespace(
        cloud_tensor=lambda dev: pts.to(dev))
    depth = 1.0 + 2.0 * torch.rand(H, W, generator=g)
    depth[100:180, 200:330] = 0.0
    depth = depth.to(DEV)
    total = 0
    for seed, edge in ((1, -4), (2, 20), (3, 0)):
        algo.config.mapping_frustum_edge = edge
        q = torch.randn(4, generator=torch.Generator().manual_seed(seed))
        c2w = torch.eye(4)
        c2w[:3, :3] = quaternion_to_matrix(q / q.norm())
        c2w[:3, 3] = torch.tensor([0.1 * seed, -0.2, 0.3])
        got = PointSLAM.get_mask_from_c2w(algo, c2w, depth)
        want = PointSLAM.get_mask_from_c2w_torch(algo, c2w, depth)
        assert got.dtype == torch.bool and got.shape == want.shape
        diff = int((got != want).sum())
        _note(f'map_ops/frustum_mask/edge{edge}\tmismatches={diff}\t'
              f'selected={int(want.sum())}')
        assert diff == 0
        total += int(want.sum())
    assert total > 500
    # no depth anywhere: every projected pixel takes the (zero) maximum
    algo.config.mapping_frustum_edge = -4
    zero = torch.zeros(H, W, device=DEV)
    assert torch.equal(PointSLAM.get_mask_from_c2w(algo, c2w, zero),
                       PointSLAM.get_mask_from_c2w_torch(algo, c2w, zero))
