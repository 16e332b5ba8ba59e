"""CPU tests of the host-side mirror of the reference plugin surface."""
import math

import numpy as np
import pytest
import torch

from xrdslam_amd.data.synthetic import SyntheticRoom, look_at
from xrdslam_amd.slam.common.camera import Camera
from xrdslam_amd.slam.common.frame import Frame
from xrdslam_amd.slam.engine.optimizers import (AdamOptimizerConfig,
                                                Optimizers)
from xrdslam_amd.slam.engine.schedulers import (LRconfig,
                                                NiceSLAMSchedulerConfig)
from xrdslam_amd.slam.utils.opt_pose import (OptimizablePose,
                                             matrix_to_quaternion,
                                             quaternion_to_matrix)


def test_sincos_cody_waite_accuracy():
    """the device sin/cos (csrc/common.h sincos_cw), emulated in numpy with the
    same constants and fma order, stays < 1.5e-7 up to 3e4 rad"""
    f = np.float32

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) +
                c.astype(np.float64)).astype(f)

    x = np.random.default_rng(0).uniform(-3e4, 3e4, 500000).astype(f)
    k = np.rint(x * f(0.636619772367581)).astype(f)
    r = fma(k, np.full_like(x, -1.57079601e+00), x)
    r = fma(k, np.full_like(x, -3.13916473e-07), r)
    r = fma(k, np.full_like(x, -5.39030253e-15), r)
    s = (r * r).astype(f)
    ps = fma(s, np.full_like(x, -1.9515295891e-4), np.full_like(x, 8.3321608736e-3))
    ps = fma(s, ps, np.full_like(x, -1.6666654611e-1))
    sn = fma((r * s).astype(f), ps, r)
    pc = fma(s, np.full_like(x, 2.443315711809948e-5), np.full_like(x, -1.388731625493765e-3))
    pc = fma(s, pc, np.full_like(x, 4.166664568298827e-2))
    cs = fma((s * s).astype(f), pc, fma(s, np.full_like(x, -0.5), np.full_like(x, 1.0)))
    n = k.astype(np.int64) & 3
    S = np.where(n == 0, sn, np.where(n == 1, cs, np.where(n == 2, -sn, -cs)))
    C = np.where(n == 0, cs, np.where(n == 1, -sn, np.where(n == 2, -cs, sn)))
    assert np.abs(S - np.sin(x.astype(np.float64))).max() < 1.5e-7
    assert np.abs(C - np.cos(x.astype(np.float64))).max() < 1.5e-7


@pytest.mark.parametrize('rot_rep', ['axis_angle', 'quat'])
@pytest.mark.parametrize('separate', [False, True])
def test_pose_roundtrip_and_gradient(rot_rep, separate):
    Rt = torch.tensor(look_at([0.3, -0.2, 0.5], [1.0, 2.0, 0.1]),
                      dtype=torch.float32)
    pose = OptimizablePose.from_matrix(Rt, separate_LR=separate,
                                       rot_rep=rot_rep)
    assert torch.allclose(pose.matrix(), Rt, atol=1e-5)
    n_par = len(list(pose.parameters()))
    assert n_par == (2 if separate else 1)
    pose.matrix()[:3, :].sum().backward()
    assert all(p.grad is not None for p in pose.parameters())


def test_quaternion_matrix_consistency():
    q = torch.tensor([0.3, -0.5, 0.1, 0.8])
    R = quaternion_to_matrix(q)
    assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-6)
    q2 = matrix_to_quaternion(R)
    assert torch.allclose(quaternion_to_matrix(q2), R, atol=1e-6)
    assert OptimizablePose.axis_angle_to_rotation_matrix(
        torch.zeros(3)).equal(torch.eye(3))


def test_frame_params_and_inconsistent_pose():
    d = np.ones((4, 6), np.float32)
    c = np.zeros((4, 6, 3), np.float32)
    f = Frame(0, c, d, init_pose=np.eye(4, dtype=np.float32),
              separate_LR=True, rot_rep='quat')
    assert [tuple(p.shape) for p in f.get_params()] == [(4, ), (3, )]
    bad = np.eye(4, dtype=np.float32)
    bad[:3, :3] *= 2.0  # not a rotation
    with pytest.raises(ValueError):
        Frame(0, c, d, init_pose=bad)


def test_optimizers_missing_group_raises():
    p = torch.nn.Parameter(torch.zeros(3))
    with pytest.raises(RuntimeError):
        Optimizers({'a': {'optimizer': AdamOptimizerConfig()}}, {'b': [p]})


def test_nice_scheduler_stage_factors_and_accum():
    cfg = NiceSLAMSchedulerConfig(coarse=False, max_steps=60,
                                  stage_lr=LRconfig(coarse=9., middle=0.1,
                                                    fine=0.01, color=0.001))
    sch = cfg.setup()
    assert [sch.factor(s) for s in (0, 24, 25, 36, 37, 59)] == \
        [0.1, 0.1, 0.01, 0.01, 0.001, 0.001]
    cfg.coarse = True
    assert sch.factor(5) == 9.
    # accum_step: step only every k-th iteration (optimizers.py:157-162)
    p = torch.nn.Parameter(torch.ones(2))
    oc = AdamOptimizerConfig(lr=0.1, accum_step=3)
    opt = Optimizers({'g': {'optimizer': oc, 'scheduler': None}}, {'g': [p]})
    for step in range(3):
        opt.zero_grad_all()
        (p.sum() * 2).backward()
        before = p.detach().clone()
        opt.optimizer_step_all(step)
        moved = not torch.equal(before, p.detach())
        assert moved == (step == 2)


def test_synthetic_room_is_consistent():
    ds = SyntheticRoom([[-2, 2], [-2, 2], [-1, 2]], H=24, W=32, fx=16., fy=16.,
                       cx=15.5, cy=11.5, n_frames=4, shrink=0.2)
    a, b = ds[1], ds[1]
    assert np.array_equal(a['depth'], b['depth'])  # seeded
    assert a['depth'].shape == (24, 32) and a['rgb'].shape == (24, 32, 3)
    frac0 = (a['depth'] == 0).mean()
    assert 0.0 < frac0 < 0.1
    # back-projected points lie on the room walls / spheres (inside the box)
    d = a['depth']
    c2w = a['c2w']
    j, i = np.meshgrid(np.arange(24.), np.arange(32.), indexing='ij')
    dirs = np.stack([(i - 15.5) / 16., -(j - 11.5) / 16., -np.ones_like(i)], -1)
    pts = c2w[:3, 3] + (dirs @ c2w[:3, :3].T) * d[..., None]
    ok = d > 0
    assert (pts[ok] >= ds.lo - 1e-3).all() and (pts[ok] <= ds.hi + 1e-3).all()


def test_convonet_bound_and_grid_shapes_cpu():
    """App. B.1: f32-contaminated bound -> (71,75,63) fine grid, not (72,76,64)"""
    from xrdslam_amd.slam.configs.input_config import nice_slam_config
    cfg = nice_slam_config()
    from xrdslam_amd.slam.models.conv_onet import ConvOnet
    bb = torch.from_numpy(np.array(cfg.mapping_bound))
    cfg.model.coarse = True
    torch.manual_seed(0)
    m = ConvOnet(cfg.model, Camera(320., 320., 319.5, 239.5, 640, 480), bb)
    assert abs(float(m.bounding_box[0, 1]) - 6.0199995) < 1e-6
    assert tuple(m.grid_c['grid_fine'].shape) == (1, 32, 63, 75, 71)
    assert tuple(m.grid_c['grid_middle'].shape) == (1, 32, 31, 37, 35)
    assert tuple(m.grid_c['grid_coarse'].shape) == (1, 32, 10, 12, 11)
    assert m.decoder.color_decoder.flat.numel() == 15899


def test_unbuilt_model_options_are_refused_loudly():
    """options of the reference models the kernels are not built for fail at
    construction instead of being ignored (NICE-SLAM: the second,
    inverse-CDF sampling pass of conv_onet.py:498-512)"""
    import pytest
    from xrdslam_amd.slam.configs.input_config import nice_slam_config
    from xrdslam_amd.slam.models.conv_onet import ConvOnet
    cfg = nice_slam_config()
    cfg.model.rendering_n_importance = 8
    bb = torch.from_numpy(np.array(cfg.mapping_bound))
    with pytest.raises(NotImplementedError, match='n_importance'):
        ConvOnet(cfg.model, Camera(320., 320., 319.5, 239.5, 640, 480), bb)
    cfg = nice_slam_config()
    cfg.model.rendering_perturb = 1.0
    cfg.model.rendering_lindisp = True
    with pytest.raises(NotImplementedError, match='lindisp.*perturb'):
        ConvOnet(cfg.model, Camera(320., 320., 319.5, 239.5, 640, 480), bb)


def test_synthetic_room_preload_is_transparent():
    """preloaded frames are the frames generated on demand; items are copies
    (callers may edit the dict), and on the CPU no device images are added"""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    bound = [[-2.0, 2.0], [-2.4, 1.8], [-1.6, 2.0]]
    a = SyntheticRoom(bound, H=24, W=32, fx=16., fy=16., cx=15.5, cy=11.5,
                      n_frames=8)
    b = SyntheticRoom(bound, H=24, W=32, fx=16., fy=16., cx=15.5, cy=11.5,
                      n_frames=8).preload(range(-1, 20))
    assert sorted(b._cache) == list(range(8))
    for k in (0, 5):
        x, y = a[k], b[k]
        assert set(x) == set(y) == {'index', 'rgb', 'depth', 'c2w'}
        for key in ('rgb', 'depth', 'c2w'):
            assert np.array_equal(x[key], y[key])
    item = b[3]
    item['c2w'] = None
    assert b[3]['c2w'] is not None


def test_trajectory_evaluation_recovers_a_similarity_transform(tmp_path):
    """eval.tar round trip and the closed-form alignment (utils/eval_traj.py;
    pinned against the reference's evaluate_ate in
    tests/test_reference_host_parity.py)"""
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    from xrdslam_amd.slam.utils import eval_traj as et
    g = torch.Generator().manual_seed(1)
    q = torch.randn(4, generator=g)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w),
                       2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                       2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w),
                       1 - 2 * (x * x + y * y)]])
    gt, est = [], []
    for k in range(25):
        p = torch.eye(4)
        p[:3, 3] = torch.tensor([0.1 * k, np.sin(0.3 * k), 0.02 * k * k])
        gt.append(p)
        e = torch.eye(4)
        e[:3, 3] = (R.T @ (p[:3, 3] - torch.tensor([1., 2., 3.]))) / 0.8
        est.append(e)

    class Algo:
        def get_gt_c2w_list_ori(self): return gt
        def get_gt_c2w_list(self): return gt
        def get_estimate_c2w_list(self): return est

    slam = SequentialSLAM.__new__(SequentialSLAM)
    slam.algorithm = Algo()
    st = slam.trajectory_stats(align=True, correct_scale=True)
    assert st['compared_pose_pairs'] == 25
    assert st['absolute_translational_error.rmse'] < 1e-6
    assert abs(st['scale'] - 0.8) < 1e-6
    assert np.allclose(st['rot'], R.numpy(), atol=1e-6)
    assert np.allclose(st['trans'], [1., 2., 3.], atol=1e-5)
    rigid = slam.trajectory_stats(align=True)
    assert rigid['scale'] == 1.0
    assert rigid['absolute_translational_error.rmse'] > 1e-3
    raw = slam.trajectory_stats(align=False)
    assert abs(raw['absolute_translational_error.rmse'] -
               slam.ate_rmse()) < 1e-6
    path = str(tmp_path / 'eval.tar')
    slam.save_eval_tar(path)
    again = et.evaluate_eval_tar(path, correct_scale=True)
    assert abs(again['scale'] - 0.8) < 1e-6


def test_pose_conversion_round_trips():
    """quaternion -> axis-angle -> Rodrigues matrix == quaternion -> matrix, and
    matrix -> quaternion -> matrix is the identity (the property pytorch3d's
    conversions are defined by; sign-standardised quaternion: angle <= pi)"""
    from xrdslam_amd.slam.utils import opt_pose as mp
    g = torch.Generator().manual_seed(8)
    q = torch.randn(300, 4, generator=g, dtype=torch.float64)
    q = torch.cat([q, torch.tensor([[1., 0, 0, 0], [1., 1e-12, 0, 0],
                                    [0., 1, 0, 0], [-0.3, 0.2, 0.1, 0.9]],
                                   dtype=torch.float64)])
    q = q / q.norm(dim=1, keepdim=True)
    R = mp.quaternion_to_matrix(q)
    assert torch.allclose(R @ R.transpose(-1, -2),
                          torch.eye(3, dtype=torch.float64).expand_as(R),
                          atol=1e-12)
    aa = mp.quaternion_to_axis_angle(q)
    for k in range(q.shape[0]):
        Rk = mp.OptimizablePose.axis_angle_to_rotation_matrix(aa[k])
        assert torch.allclose(Rk, R[k], atol=1e-9), k
        qk = mp.matrix_to_quaternion(R[k])
        assert qk[0] >= 0
        assert torch.allclose(mp.quaternion_to_matrix(qk), R[k], atol=1e-9)
        # from_matrix (what Frame.set_pose uses) reproduces the matrix
        Rt = torch.eye(4, dtype=torch.float64)
        Rt[:3, :3] = R[k]
        Rt[:3, 3] = torch.tensor([0.1, -0.2, 0.3], dtype=torch.float64)
        for rep in ('axis_angle', 'quat'):
            pose = mp.OptimizablePose.from_matrix(Rt, separate_LR=True,
                                                  rot_rep=rep)
            assert torch.allclose(pose.matrix().double(), Rt, atol=1e-6), \
                (k, rep)
            if rep == 'axis_angle':
                assert float(pose.data_r.detach().norm()) <= np.pi + 1e-6
