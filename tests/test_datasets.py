"""File datasets and the prefetching loader on synthetic
Replica- / TUM-formatted folders written with PIL; the pure-numpy parts of the
reference readers (pose convention, time-stamp association) are executed from
/root/reference where it exists."""
import os
import sys

import numpy as np
import pytest
import torch

from xrdslam_amd.data import datasets as ds
from xrdslam_amd.data.synthetic import SyntheticRoom

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 48, 64
CAM = dict(fx=60., fy=60., cx=31.5, cy=23.5)


def _room(n=5):
    return SyntheticRoom([[-3, 3], [-4, 2.5], [-2, 2.5]], H=H, W=W,
                         n_frames=n, device='cpu', **CAM)


def _write_replica(path, room, crop=0, down=1):
    from PIL import Image
    os.makedirs(os.path.join(path, 'results'))
    with open(os.path.join(path, 'devices.yaml'), 'w') as f:
        f.write(f'cam:\n  H: {H}\n  W: {W}\n  fx: {CAM["fx"]}\n  fy: '
                f'{CAM["fy"]}\n  cx: {CAM["cx"]}\n  cy: {CAM["cy"]}\n'
                f'  png_depth_scale: 6553.5\n  crop_edge: {crop}\n'
                f'  downsample_factor: {down}\n')
    lines = []
    for k in range(room.n_frames):
        it = room[k]
        rgb = np.clip(np.rint(it['rgb'] * 255), 0, 255).astype(np.uint8)
        Image.fromarray(rgb).save(
            os.path.join(path, 'results', f'frame{k:06d}.jpg'), quality=100,
            subsampling=0)
        d16 = np.clip(np.rint(it['depth'] * 6553.5), 0, 65535).astype(
            np.uint16)
        Image.fromarray(d16).save(os.path.join(path, 'results',
                                               f'depth{k:06d}.png'))
        cv = it['c2w'].copy()          # OpenGL -> dataset convention
        cv[:3, 1] *= -1
        cv[:3, 2] *= -1
        lines.append(' '.join(f'{v:.9e}' for v in cv.reshape(-1)))
    with open(os.path.join(path, 'traj.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')


def test_replica_reader_roundtrip(tmp_path):
    room = _room()
    _write_replica(str(tmp_path / 'r'), room)
    data = ds.get_dataset(str(tmp_path / 'r'), 'replica')
    assert len(data) == room.n_frames
    cam = data.get_camera()
    assert (cam.width, cam.height, cam.fx, cam.cx) == (W, H, 60., 31.5)
    for k in (0, 3):
        a, b = data[k], room[k]
        assert a['depth'].dtype == np.float32 and a['depth'].shape == (H, W)
        assert np.abs(a['depth'] - b['depth']).max() < 1.0 / 6553.5
        assert (a['depth'] == 0).sum() == (b['depth'] == 0).sum()
        assert np.abs(a['rgb'] - b['rgb']).max() < 0.03   # JPEG q=100
        assert np.allclose(a['c2w'], b['c2w'], atol=1e-6)


def test_crop_and_downsample_follow_the_reference_intrinsics(tmp_path):
    room = _room(2)
    _write_replica(str(tmp_path / 'r'), room, crop=4, down=2)
    data = ds.Replica(str(tmp_path / 'r'))
    cam = data.get_camera()
    assert (cam.height, cam.width) == ((H - 8) // 2, (W - 8) // 2)
    assert cam.fx == 30. and cam.cx == (31.5 - 4) / 2
    it = data[1]
    assert it['rgb'].shape == (cam.height, cam.width, 3)
    # nearest for depth: every value is one of the cropped source pixels
    src = data.__class__(str(tmp_path / 'r'))
    src.crop_edge, src.downsample_factor = 4, 1
    full = src[1]['depth']
    assert np.array_equal(it['depth'], full[::2, ::2])


def test_resamplers_known_answers():
    img = np.arange(16, dtype=np.float64).reshape(4, 4)
    # 4 -> 2: centres of the output fall between source pixels 0|1 and 2|3
    out = ds.resize_bilinear(img, 2, 2)
    assert np.allclose(out, [[2.5, 4.5], [10.5, 12.5]])
    assert np.array_equal(ds.resize_nearest(img, 2, 2), [[0, 2], [8, 10]])
    assert ds.resize_bilinear(img, 4, 4) is img
    # undistortion with zero coefficients is the identity
    rgb = np.random.default_rng(0).random((6, 8, 3))
    assert np.allclose(ds.undistort(rgb, 5., 5., 3.5, 2.5, [0, 0, 0, 0, 0]),
                       rgb)
    # a positive k1 pulls samples from farther out: the centre stays put
    und = ds.undistort(rgb, 5., 5., 4.0, 3.0, [0.1, 0, 0, 0, 0])
    assert np.allclose(und[3, 4], rgb[3, 4])


def _write_tum(path, room):
    from PIL import Image
    from scipy.spatial.transform import Rotation
    os.makedirs(os.path.join(path, 'rgb'))
    os.makedirs(os.path.join(path, 'depth'))
    with open(os.path.join(path, 'devices.yaml'), 'w') as f:
        f.write(f'cam:\n  H: {H}\n  W: {W}\n  fx: 60.0\n  fy: 60.0\n'
                '  cx: 31.5\n  cy: 23.5\n  png_depth_scale: 5000.0\n')
    rgb_l, dep_l, gt_l = [], [], ['# timestamp tx ty tz qx qy qz qw']
    for k in range(room.n_frames):
        it = room[k]
        t = 100.0 + k * 0.05
        Image.fromarray(np.clip(np.rint(it['rgb'] * 255), 0, 255).astype(
            np.uint8)).save(os.path.join(path, 'rgb', f'{t:.6f}.png'))
        Image.fromarray(np.clip(np.rint(it['depth'] * 5000), 0, 65535).astype(
            np.uint16)).save(os.path.join(path, 'depth', f'{t + .01:.6f}.png'))
        rgb_l.append(f'{t:.6f} rgb/{t:.6f}.png')
        dep_l.append(f'{t + .01:.6f} depth/{t + .01:.6f}.png')
        cv = it['c2w'].copy()
        cv[:3, 1] *= -1
        cv[:3, 2] *= -1
        q = Rotation.from_matrix(cv[:3, :3]).as_quat()
        gt_l.append(f'{t + .004:.6f} ' + ' '.join(f'{v:.9f}' for v in
                                                  list(cv[:3, 3]) + list(q)))
    for name, rows in (('rgb.txt', rgb_l), ('depth.txt', dep_l),
                       ('groundtruth.txt', gt_l)):
        with open(os.path.join(path, name), 'w') as f:
            f.write('\n'.join(rows) + '\n')


def test_tum_reader(tmp_path):
    room = _room(6)
    _write_tum(str(tmp_path / 't'), room)
    data = ds.get_dataset(str(tmp_path / 't'), 'tumrgbd')
    assert len(data) == 6          # 20 Hz < frame_rate 32: every frame kept
    it = data[2]
    assert np.abs(it['depth'] - room[2]['depth']).max() < 1.0 / 5000
    assert np.allclose(it['c2w'], room[2]['c2w'], atol=1e-5)
    assert np.abs(it['rgb'] - room[2]['rgb']).max() < 0.003  # PNG: exact/255
    assert len(ds.TUM_RGBD(str(tmp_path / 't'), frame_rate=12)) == 3


@pytest.mark.skipif(not os.path.isdir('/root/reference/slam'),
                    reason='reference tree not present')
def test_association_and_pose_convention_match_the_reference():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ref_harness
    ref_harness.install()
    from slam.common.datasets import TUM_RGBD, Replica
    rng = np.random.default_rng(0)
    t_img = np.sort(rng.uniform(0, 10, 200))
    t_dep = np.sort(t_img + rng.normal(0, 0.05, 200))
    t_pose = np.sort(rng.uniform(0, 10, 900))
    want = TUM_RGBD.associate_frames(None, t_img, t_dep, t_pose)
    assert ds.associate_frames(t_img, t_dep, t_pose) == \
        [tuple(int(x) for x in w) for w in want]
    want2 = TUM_RGBD.associate_frames(None, t_img, t_dep, None)
    assert ds.associate_frames(t_img, t_dep, None) == \
        [tuple(int(x) for x in w) for w in want2]

    class Dummy:
        n_img = 3
    import tempfile
    rows = rng.normal(size=(3, 16))
    with tempfile.NamedTemporaryFile('w', suffix='.txt', delete=False) as f:
        f.write('\n'.join(' '.join(f'{v:.9e}' for v in r) for r in rows))
    d = Dummy()
    Replica.load_poses(d, f.name)
    for k in range(3):
        ours = ds._to_opengl(rows[k].reshape(4, 4)).astype(np.float32)
        assert np.array_equal(ours, d.poses[k].numpy())


def test_prefetcher_yields_the_dataset_items_in_order(tmp_path):
    room = _room(6)
    _write_replica(str(tmp_path / 'r'), room)
    data = ds.Replica(str(tmp_path / 'r'))
    loader = ds.Prefetcher(data, 'cpu', depth=2)
    try:
        for k in range(len(data)):
            a, b = loader[k], data[k]
            assert a['index'] == k and np.array_equal(a['depth'], b['depth'])
            assert np.array_equal(a['rgb'], b['rgb'])
        again = loader[1]              # random access after the fact
        assert np.array_equal(again['depth'], data[1]['depth'])
    finally:
        loader.close()


@pytest.mark.gpu
def test_prefetcher_uploads_once_to_the_device(tmp_path):
    room = _room(5)
    _write_replica(str(tmp_path / 'r'), room)
    data = ds.Replica(str(tmp_path / 'r'))
    loader = ds.Prefetcher(data, 'cuda:0', depth=2)
    try:
        for k in range(len(data)):
            it = loader[k]
            assert it['depth_dev'].is_cuda and it['depth_dev'].shape == \
                (H * W, 1)
            assert np.array_equal(it['depth_dev'].cpu().numpy().reshape(H, W),
                                  data[k]['depth'])
            assert np.array_equal(it['rgb_dev'].cpu().numpy().reshape(H, W, 3),
                                  data[k]['rgb'])
    finally:
        loader.close()
