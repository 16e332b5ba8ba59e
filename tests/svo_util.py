"""shared helpers for the Vox-Fusion native-op tests"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_lib():
    lib_path = os.path.join(ROOT, 'oracle', '_build', 'libsvo_oracle.so')
    if not os.path.exists(lib_path):
        subprocess.check_call([sys.executable,
                               os.path.join(ROOT, 'oracle', 'build_oracle.py')])
    return C.CDLL(lib_path)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def svo_intersect_oracle(ray_start, ray_dir, points, children, voxelsize,
                         n_max):
    lib = oracle_lib()
    B, M = ray_start.shape[:2]
    N = points.shape[1]
    idx = np.zeros((B, M, n_max), np.int32)
    mn = np.zeros((B, M, n_max), np.float32)
    mx = np.zeros((B, M, n_max), np.float32)
    lib.svo_intersect_ref.restype = C.c_int
    deepest = lib.svo_intersect_ref(
        C.c_int(B), C.c_int(N), C.c_int(M), C.c_float(voxelsize),
        C.c_int(n_max), P(ray_start), P(ray_dir), P(points), P(children),
        P(idx), P(mn), P(mx))
    return idx, mn, mx, deepest


def inverse_cdf_oracle(pts_idx, mn, mx, noise, probs, steps, fixed):
    lib = oracle_lib()
    G, R, Pn = mn.shape
    S = noise.shape[-1]
    sidx = -np.ones((G, R, S), np.int32)
    sdep = np.zeros((G, R, S), np.float32)
    sdis = np.zeros((G, R, S), np.float32)
    lib.inverse_cdf_sampling_ref(
        C.c_int(G), C.c_int(R), C.c_int(Pn), C.c_int(S), C.c_float(fixed),
        P(pts_idx), P(mn), P(mx), P(noise), P(probs), P(steps), P(sidx),
        P(sdep), P(sdis))
    return sidx, sdep, sdis


def make_tree(seed=0, n_vox=1500):
    """octree arrays (centres in metres, children+side) like
    SparseVoxel.get_octree builds them (slam/models/sparse_voxel.py:306-331)"""
    import torch
    from xrdslam_amd.compat import svo
    rng = np.random.default_rng(seed)
    vox = rng.integers(50, 80, size=(n_vox, 3)).astype(np.int32)
    svo.reset_id_counter()
    tree = svo.Octree()
    tree.init(256, 16, 0.2)
    tree.insert(torch.from_numpy(vox))
    voxels, children, features = tree.get_centres_and_children()
    voxel_size = 0.2
    centres = (voxels[:, :3] + voxels[:, -1:] / 2) * voxel_size
    childs = torch.cat([children, voxels[:, -1:]], -1).int()
    return centres.numpy().astype(np.float32), childs.numpy().astype(np.int32)
