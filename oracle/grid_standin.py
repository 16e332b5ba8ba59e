"""TEST INFRASTRUCTURE ONLY.  CPU stand-in for the reference's ``grid`` pybind
module (third_party/sparse_voxels) backed by oracle/svo_oracle.c ("parity
unpinned": line-by-line restatement of the CUDA kernels, which cannot run
here; the HIP kernels are tested bit-exact against it).  Used to execute the
reference's Vox-Fusion model on the CPU for tests/golden/voxfusion_render.npz
and by the CPU test of the host mirror."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import svo_util  # noqa: E402


def svo_intersect(ray_start, ray_dir, points, children, voxelsize, n_max):
    idx, mn, mx, _ = svo_util.svo_intersect_oracle(
        np.ascontiguousarray(ray_start.detach().numpy(), np.float32),
        np.ascontiguousarray(ray_dir.detach().numpy(), np.float32),
        np.ascontiguousarray(points.detach().numpy(), np.float32),
        np.ascontiguousarray(children.detach().numpy(), np.int32),
        float(voxelsize), int(n_max))
    return torch.from_numpy(idx), torch.from_numpy(mn), torch.from_numpy(mx)


def inverse_cdf_sampling(pts_idx, min_depth, max_depth, noise, probs, steps,
                         fixed_step_size):
    a = [np.ascontiguousarray(t.detach().numpy(), d) for t, d in (
        (pts_idx, np.int32), (min_depth, np.float32), (max_depth, np.float32),
        (noise, np.float32), (probs, np.float32), (steps, np.float32))]
    sidx, sdep, sdis = svo_util.inverse_cdf_oracle(*a, float(fixed_step_size))
    return (torch.from_numpy(sidx), torch.from_numpy(sdep),
            torch.from_numpy(sdis))


def module():
    m = types.ModuleType('grid')
    m.svo_intersect = svo_intersect
    m.inverse_cdf_sampling = inverse_cdf_sampling
    return m
