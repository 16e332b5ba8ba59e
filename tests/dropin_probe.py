"""Run in a SUBPROCESS by tests/test_reference_dropin.py: the reference's own
model classes on top of xrdslam_amd.compat (CPU container: construction,
parameter groups, octree calls through torch.classes.svo; no kernels run)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle')]
import numpy as np
import torch

import ref_harness

ref_harness.install()
if not torch.cuda.is_available():
    # the reference hard-codes device='cuda' in a few factory calls
    # (sparse_voxel.py:309-313); on the CPU-only build container they land on
    # the host
    _zeros = torch.zeros

    def zeros(*a, **k):
        if str(k.get('device', '')).startswith('cuda'):
            k['device'] = 'cpu'
        return _zeros(*a, **k)
    torch.zeros = zeros
for name in ('tinycudann', 'grid', 'faiss', 'diff_gaussian_rasterization'):
    sys.modules.pop(name, None)          # the harness' mocks give way
from xrdslam_amd import compat

compat.install()
out = {}
# --- Vox-Fusion: torch.classes.svo.Octree + the `grid` module -----------------
from slam.common.camera import Camera
from slam.models.sparse_voxel import SparseVoxel, SparseVoxelConfig

cam = Camera(320., 320., 319.5, 239.5, 640, 480)
m = SparseVoxelConfig().setup(camera=cam, bounding_box=None)
assert isinstance(m, SparseVoxel)
groups = m.get_param_groups()
out['vox_groups'] = sorted(groups)
out['vox_n_decoder'] = int(sum(p.numel() for p in groups['decoder']))
pts = torch.from_numpy(np.random.default_rng(0).uniform(
    10.0, 12.0, (500, 3)).astype(np.float32))
m.insert_points(pts)
st = m.get_map_states()
out['vox_nodes'] = int(m.svo.count_nodes())
out['vox_state_keys'] = sorted(st)
out['vox_vertex_idx_shape'] = list(st['voxel_vertex_idx'].shape)
import grid as ref_grid
out['grid_module'] = ref_grid.__name__
# --- Co-SLAM: tinycudann.Encoding ------------------------------------------------
from slam.models.joint_encoding import JointEncoding, JointEncodingConfig

je = JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True).setup(
    camera=cam, bounding_box=torch.tensor([[-3., 3.], [-4., 2.5],
                                           [-2., 2.5]], dtype=torch.float64))
assert isinstance(je, JointEncoding)
g2 = je.get_param_groups()
out['co_groups'] = sorted(g2)
out['co_n_decoder'] = int(sum(p.numel() for p in g2['decoder']))
out['co_n_table'] = int(sum(p.numel() for p in g2['embed_fn']))
import tinycudann
out['tcnn_module'] = tinycudann.__name__
print('DROPIN ' + json.dumps(out))
