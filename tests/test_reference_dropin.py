"""Drop-in check of the boundary: the REFERENCE's own model
classes — imported from /root/reference, unmodified — construct and manage
their map on top of ``xrdslam_amd.compat``: ``torch.classes.svo.Octree`` (the
TorchScript seam, csrc_torch/svo_class.cpp), the ``grid`` and ``tinycudann``
import names.  Runs in a subprocess (the compiled reference octree of
tests/test_octree.py registers the same ``svo`` class name in its process).
No kernel runs here (CPU container); the GPU suites exercise the kernels
behind the same shims.

``test_reference_models_execute_on_the_shims`` goes one step further: the
reference's ``JointEncoding``, ``SparseVoxel`` and Point-SLAM ``ConvOnet2`` run
forward, losses and backward THROUGH the shims, with the C-ABI's compute entry points served by a
host backend (tests/host_abi.py: the oracles on host pointers), and reproduce
the committed goldens — the ones the HIP kernels are checked against on the
GPU.  Reference code -> shim -> boundary protocol is executed here; boundary
-> kernels there."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/slam'),
                                reason='reference tree not present')


def test_reference_models_run_on_the_shims():
    r = subprocess.run([sys.executable, os.path.join(HERE, 'dropin_probe.py')],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('DROPIN ')][-1]
    out = json.loads(line[7:])
    # Vox-Fusion: SparseVoxel built its octree through torch.classes.svo and
    # exported the map arrays update_map_states expects
    assert out['vox_groups'] == ['decoder', 'embeddings']
    assert out['vox_n_decoder'] == 54276          # 16->128->128->129, 144->128->3
    assert out['vox_nodes'] > 100
    assert out['vox_state_keys'] == ['voxel_center_xyz', 'voxel_structure',
                                     'voxel_vertex_emb', 'voxel_vertex_idx']
    assert out['vox_vertex_idx_shape'] == [out['vox_nodes'], 8]
    assert out['grid_module'] == 'xrdslam_amd.compat.grid'
    # Co-SLAM: JointEncoding on the tinycudann shim (2^16 table, 16 levels)
    assert out['co_groups'] == ['decoder', 'embed_fn']
    assert out['co_n_decoder'] == 5184
    assert out['co_n_table'] == 1640944
    assert out['tcnn_module'] == 'xrdslam_amd.compat.tinycudann'


def test_reference_models_execute_on_the_shims():
    r = subprocess.run([sys.executable,
                        os.path.join(HERE, 'dropin_exec_probe.py')],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines()
            if l.startswith('DROPIN_EXEC ')][-1]
    out = json.loads(line[12:])
    # Co-SLAM: tracking, mapping and first-frame mapping calls of the
    # reference's JointEncoding on compat.tinycudann; every output, loss term
    # and gradient equals the golden the reference produced on the oracle
    # encodings (same arithmetic behind the boundary: exact)
    assert out['coslam_calls'] == ['xrd_hashgrid_bwd', 'xrd_hashgrid_fwd',
                                   'xrd_oneblob_bwd', 'xrd_oneblob_fwd']
    for tag, (name, err) in out['coslam_worst'].items():
        assert err < 1e-6, (tag, name, err)
    # Vox-Fusion: the reference's SparseVoxel on torch.classes.svo (octree
    # arrays equal the compiled reference's) and compat.grid
    assert out['vox_calls'] == ['xrd_inverse_cdf_sampling',
                                'xrd_svo_intersect']
    assert out['vox_rays_hit'] == 280
    assert out['vox_worst'][1] < 1e-5, out['vox_worst']
    # Point-SLAM: the golden's whole generation procedure (the reference's
    # ConvOnet2 / NeuralPointCloud / decoders: two frames of point insertion,
    # renders in both stages, tracking and mapping losses, all gradients) on
    # compat.faiss; all 256 arrays of the committed file
    assert out['point_calls'] == ['xrd_knn_cell_ids', 'xrd_knn_cell_ranges',
                                  'xrd_knn_search_count']
    assert out['point_keys'] == 256 and out['point_cloud'] == [519, 858]
    assert out['point_worst'][1] < 1e-5, out['point_worst']


def test_svo_class_matches_python_shim():
    """the TorchScript class and the ctypes Octree give the same arrays"""
    code = '''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from xrdslam_amd import build_torch_ext
from xrdslam_amd.compat import svo
vox = torch.from_numpy(np.random.default_rng(1).integers(40, 90, (3000, 3)).astype(np.int32))
svo.reset_id_counter(); a = svo.Octree(); a.init(256, 16, 0.2); a.insert(vox)
ra = a.get_centres_and_children()
svo.reset_id_counter(); B = build_torch_ext.load(); b = B(); b.init(256, 16, 0.2); b.insert(vox)
rb = b.get_centres_and_children()
assert all(torch.equal(x, y) for x, y in zip(ra, rb))
assert a.count_nodes() == b.count_nodes() and a.count_leaf_nodes() == b.count_leaf_nodes()
assert abs(a.try_insert(vox[:100]) - b.try_insert(vox[:100])) < 1e-12
print("SAME", b.count_nodes())
''' % os.path.dirname(HERE)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and 'SAME' in r.stdout, r.stderr[-2000:]
