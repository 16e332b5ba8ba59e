"""Python module with the function surface of the reference's pybind extension
``grid`` (third_party/sparse_voxels/src/binding.cpp:10-21), backed by the HIP
kernels.  Live functions: ``svo_intersect`` and ``inverse_cdf_sampling`` (the
only two the reference calls, slam/model_components/voxel_helpers_voxfusion.py:
248-255, 441-450); the never-called ones raise NotImplementedError.

Input checks mirror include/utils.h:10-34 (CHECK_CONTIGUOUS / CHECK_CUDA /
CHECK_IS_FLOAT / CHECK_IS_INT -> RuntimeError); launch errors raise instead of
the reference's ``exit(-1)``."""
from __future__ import annotations

import torch

from .. import _lib


def _require_device(t, name):
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA tensor')


def _check(t, name, dtype):
    _require_device(t, name)
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be a contiguous tensor')
    if t.dtype != dtype:
        kind = 'a float' if dtype == torch.float32 else 'an int'
        raise RuntimeError(f'{name} must be {kind} tensor')


def svo_intersect(ray_start, ray_dir, points, children, voxelsize, n_max):
    """-> (idx i32, min_depth f32, max_depth f32), each [B,M,n_max]"""
    for t, n in ((ray_start, 'ray_start'), (ray_dir, 'ray_dir'),
                 (points, 'points')):
        _check(t, n, torch.float32)
    _check(children, 'children', torch.int32)
    B, M = ray_start.shape[0], ray_start.shape[1]
    shared = points.dim() == 2 or points.shape[0] == 1 and B > 1
    N = points.shape[-2]
    idx = torch.zeros(B, M, n_max, dtype=torch.int32, device=ray_start.device)
    mn = torch.zeros(B, M, n_max, dtype=torch.float32,
                     device=ray_start.device)
    mx = torch.zeros_like(mn)
    _lib.check(_lib.lib().xrd_svo_intersect(
        B, N, M, float(voxelsize), int(n_max), int(shared),
        _lib.ptr(ray_start), _lib.ptr(ray_dir), _lib.ptr(points),
        _lib.ptr(children), _lib.ptr(idx), _lib.ptr(mn), _lib.ptr(mx), None,
        _lib.stream_ptr(ray_start.device)), 'xrd_svo_intersect')
    return idx, mn, mx


def inverse_cdf_sampling(pts_idx, min_depth, max_depth, uniform_noise, probs,
                         steps, fixed_step_size):
    """-> (sampled_idx i32, sampled_depth f32, sampled_dists f32) [G,R,S]"""
    _check(pts_idx, 'pts_idx', torch.int32)
    for t, n in ((min_depth, 'min_depth'), (max_depth, 'max_depth'),
                 (uniform_noise, 'uniform_noise'), (probs, 'probs'),
                 (steps, 'steps')):
        _check(t, n, torch.float32)
    G, R, P = min_depth.shape
    S = uniform_noise.shape[-1]
    dev = pts_idx.device
    sidx = -torch.ones(G, R, S, dtype=torch.int32, device=dev)
    sdep = torch.zeros(G, R, S, dtype=torch.float32, device=dev)
    sdis = torch.zeros(G, R, S, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().xrd_inverse_cdf_sampling(
        G, R, P, S, float(fixed_step_size), _lib.ptr(pts_idx),
        _lib.ptr(min_depth), _lib.ptr(max_depth), _lib.ptr(uniform_noise),
        _lib.ptr(probs), _lib.ptr(steps), _lib.ptr(sidx), _lib.ptr(sdep),
        _lib.ptr(sdis), _lib.stream_ptr(dev)), 'xrd_inverse_cdf_sampling')
    return sidx, sdep, sdis


def _dead(name):
    def fn(*a, **k):
        raise NotImplementedError(
            f'grid.{name} is never called by the reference '
            '(SURVEY.md §2.2) and is not built')
    fn.__name__ = name
    return fn


ball_intersect = _dead('ball_intersect')
aabb_intersect = _dead('aabb_intersect')
triangle_intersect = _dead('triangle_intersect')
uniform_ray_sampling = _dead('uniform_ray_sampling')
build_octree = _dead('build_octree')
