"""GPU: the pose hand-over between frames on the device (xrd_pose_from_matrix,
xrd_pose_predict, SequentialSLAM(device_poses=True)) against the host chain of
the reference's tracker (numpy constant-velocity start, frame.py / opt_pose.py
conversions on the CPU)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rigid(rng, angle_scale=1.0):
    from scipy.spatial.transform import Rotation
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = Rotation.from_rotvec(rng.normal(size=3) * angle_scale) \
        .as_matrix().astype(np.float32)
    m[:3, 3] = rng.normal(size=3).astype(np.float32) * 2
    return m


def _branch_cases():
    """rotations that exercise each of the four quaternion branches (largest
    component r, i, j, k), the identity, and half turns"""
    from scipy.spatial.transform import Rotation
    out = [np.eye(3)]
    for axis in np.eye(3):
        for ang in (np.pi, np.pi - 1e-3, 3.0, 1e-9, 1e-4):
            out.append(Rotation.from_rotvec(axis * ang).as_matrix())
    out.append(Rotation.from_rotvec(np.array([1., 1., 1.]) / 3**.5 * 3.1)
               .as_matrix())
    return [r.astype(np.float32) for r in out]


@pytest.mark.parametrize('rot_rep', ['quat', 'axis_angle'])
def test_pose_from_matrix_equals_the_host_conversion(rot_rep):
    from xrdslam_amd.engine import slam_ops
    from xrdslam_amd.slam.utils.opt_pose import OptimizablePose
    rng = np.random.default_rng(3)
    mats = [_rigid(rng, s) for s in (0.01, 0.3, 1.0, 2.0, 3.0) for _ in
            range(8)]
    for r in _branch_cases():
        m = _rigid(rng)
        m[:3, :3] = r
        mats.append(m)
    worst = 0.0
    for m in mats:
        host = OptimizablePose.from_matrix(torch.from_numpy(m),
                                           separate_LR=False, rot_rep=rot_rep)
        want = host.data.detach().numpy()
        got = slam_ops.pose_from_matrix(torch.from_numpy(m).to(DEV),
                                        rot_rep).cpu().numpy()
        assert got.shape == want.shape
        # a half turn has two equivalent representations (q and -q agree up
        # to the sign convention r >= 0, which is ill-conditioned at r = 0):
        # compare through the matrix there
        back = OptimizablePose(torch.from_numpy(got), separate_LR=False,
                               rot_rep=rot_rep).matrix().detach().numpy()
        assert np.abs(back - m).max() < 2e-6, (m, got)
        if abs(np.trace(m[:3, :3]) + 1) > 1e-2:   # away from half turns
            worst = max(worst, np.abs(got - want).max())
    assert worst < 2e-6, worst


def test_pose_predict_equals_the_numpy_formula():
    from xrdslam_amd.engine import slam_ops
    from xrdslam_amd.slam.pipeline import predict_current_pose
    rng = np.random.default_rng(5)
    for _ in range(20):
        prev2 = _rigid(rng)
        step = _rigid(rng, 0.05)
        step[:3, 3] *= 0.01
        prev = (step @ prev2).astype(np.float32)
        est = [torch.from_numpy(prev2), torch.from_numpy(prev)]
        want = predict_current_pose(2, None, est)
        got = slam_ops.pose_predict(est[1].to(DEV), est[0].to(DEV)).cpu() \
            .numpy()
        assert np.abs(got - want).max() < 5e-6, (got, want)


def _run(device_poses, n_frames):
    import random

    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import cadence, coslam_config
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)
    bound = [[-3, 3], [-4, 2.5], [-2, 2.5]]
    cam = Camera(fx=150., fy=150., cx=79.5, cy=59.5, width=160, height=120)
    cfg = coslam_config(bound)
    cfg.mapping_first_n_iters = 100
    cfg.tracking_Wedge = cfg.tracking_Hedge = 5
    cfg.mapping_sample = 768
    algo = cfg.setup(camera=cam, device=DEV)
    algo.use_graphs = True
    data = SyntheticRoom(bound, H=120, W=160, fx=150., fy=150., cx=79.5,
                         cy=59.5, n_frames=200, device=DEV)
    cad = cadence['co-slam']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every, pose_device=DEV,
                          device_poses=device_poses)
    for k in range(n_frames):
        slam.step(k)
    assert slam._device_chain() == device_poses
    return algo, slam


def test_device_pose_chain_tracks_like_the_host_chain():
    """the same Co-SLAM run with the pose chain on the device and through the
    host: same estimates up to the float noise the 10 Adam steps a frame
    amplify, same trajectory error; the tracking call returns a device tensor
    (no host copy) in device mode"""
    a, sa = _run(True, 16)
    b, sb = _run(False, 16)
    ea = torch.stack([p.detach().cpu() for p in a.get_estimate_c2w_list()])
    eb = torch.stack([p.detach().cpu() for p in b.get_estimate_c2w_list()])
    assert ea.shape == eb.shape == (16, 4, 4)
    assert all(p.is_cuda for p in a.get_estimate_c2w_list()[1:])
    assert float((ea - eb).abs().max()) < 2e-3
    assert abs(sa.ate_rmse() - sb.ate_rmse()) < 2e-3
    assert sa.ate_rmse() < 0.03
    assert a.device_track_result and not b.device_track_result


def test_initial_pose_check_of_device_poses_is_deferred_not_dropped():
    """frame.py:24-29 checks every frame's initial pose against its
    parameterisation; for a pose that is already on the device the comparison
    stays there (no host read per frame) and is raised by the next trajectory
    reader (Frame.raise_if_inconsistent)"""
    from xrdslam_amd.slam.common.frame import Frame
    d = np.ones((4, 6), np.float32)
    c = np.zeros((4, 6, 3), np.float32)
    Frame.raise_if_inconsistent()          # start clean
    good = torch.from_numpy(_rigid(np.random.default_rng(3))).to(DEV)
    Frame(0, c, d, init_pose=good, separate_LR=True, rot_rep='quat',
          device=DEV)
    Frame.raise_if_inconsistent()          # consistent: nothing raised
    bad = good.clone()
    bad[:3, :3] *= 2.0                     # not a rotation
    Frame(1, c, d, init_pose=bad, separate_LR=True, rot_rep='quat',
          device=DEV)                      # no host read here ...
    Frame(2, c, d, init_pose=good, separate_LR=True, rot_rep='quat',
          device=DEV)
    with pytest.raises(ValueError):
        Frame.raise_if_inconsistent()      # ... raised here
    Frame.raise_if_inconsistent()          # and cleared


def test_nan_initial_pose_is_not_erased_by_the_next_frame():
    """the deviation is folded over frames into ONE device float: a NaN pose at
    frame k must survive frame k+1's launch with a finite deviation (a plain
    ``!(e <= err)`` update lets the finite value overwrite the NaN)"""
    from xrdslam_amd.slam.common.frame import Frame
    d = np.ones((4, 6), np.float32)
    c = np.zeros((4, 6, 3), np.float32)
    Frame.raise_if_inconsistent()
    good = torch.from_numpy(_rigid(np.random.default_rng(5))).to(DEV)
    bad = good.clone()
    bad[0, 0] = float('nan')
    for rep in ('quat', 'axis_angle'):
        Frame(1, c, d, init_pose=bad, separate_LR=True, rot_rep=rep,
              device=DEV)
        Frame(2, c, d, init_pose=good, separate_LR=True, rot_rep=rep,
              device=DEV)
        with pytest.raises(ValueError):
            Frame.raise_if_inconsistent()
        Frame.raise_if_inconsistent()      # cleared


def test_deferred_pose_check_is_polled_without_a_host_wait():
    """Frame.poll_inconsistent (called by the pipeline at mapping frames)
    copies the deviation out asynchronously and raises it on a LATER poll;
    reset_pose_check clears what another run left behind"""
    from xrdslam_amd.slam.common.frame import Frame
    d = np.ones((4, 6), np.float32)
    c = np.zeros((4, 6, 3), np.float32)
    Frame.reset_pose_check()
    good = torch.from_numpy(_rigid(np.random.default_rng(9))).to(DEV)
    bad = good.clone()
    bad[:3, :3] *= 1.5
    Frame(0, c, d, init_pose=good, separate_LR=True, rot_rep='quat',
          device=DEV)
    Frame.poll_inconsistent()              # starts a copy of a clean value
    torch.cuda.synchronize()
    Frame.poll_inconsistent()              # clean: nothing raised
    Frame(1, c, d, init_pose=bad, separate_LR=True, rot_rep='quat',
          device=DEV)
    torch.cuda.synchronize()
    Frame.poll_inconsistent()              # consumes the clean copy, re-arms
    torch.cuda.synchronize()
    with pytest.raises(ValueError):
        Frame.poll_inconsistent()          # the bad deviation arrives
    Frame(2, c, d, init_pose=bad, separate_LR=True, rot_rep='quat',
          device=DEV)
    Frame.reset_pose_check()               # a new run: nothing carried over
    Frame.raise_if_inconsistent()


def test_deferred_pose_check_covers_the_bottom_row():
    """the reference compares the full 4x4 (frame.py:24-29): a matrix whose
    last row is not 0 0 0 1 is inconsistent even with a perfect rotation"""
    from xrdslam_amd.slam.common.frame import Frame
    d = np.ones((4, 6), np.float32)
    c = np.zeros((4, 6, 3), np.float32)
    Frame.reset_pose_check()
    good = torch.from_numpy(_rigid(np.random.default_rng(11))).to(DEV)
    bad = good.clone()
    bad[3, 3] = 2.0
    Frame(1, c, d, init_pose=bad, separate_LR=True, rot_rep='quat',
          device=DEV)
    with pytest.raises(ValueError):
        Frame.raise_if_inconsistent()
