"""Bit-exact parity of the flat-array octree (csrc/octree.cpp, host code — no
GPU needed) with the COMPILED REFERENCE (third_party/sparse_octree built by
oracle/build_ref_octree.py; outputs recorded in tests/golden/octree_*.npz by
oracle/make_golden.py): node ids, centres, children and corner-feature ids
after every insert batch, counts, has_voxel, try_insert, traversal orders."""
import os
import pickle

import numpy as np
import pytest
import torch

from xrdslam_amd.compat import svo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('which', [0, 1])
def test_octree_bit_exact_vs_reference(which):
    g = np.load(os.path.join(GOLD, f'octree_{which}.npz'))
    svo.reset_id_counter()
    tree = svo.Octree()
    tree.init(256, 16, 0.2)
    for bi in range(3):
        b = torch.from_numpy(g[f'batch{bi}'])
        assert tree.try_insert(b) == float(g[f'try{bi}'])
        tree.insert(b)
        vox, ch, ft = tree.get_centres_and_children()
        assert np.array_equal(vox.numpy(), g[f'voxels{bi}'])
        assert np.array_equal(ch.numpy(), g[f'children{bi}'])
        assert np.array_equal(ft.numpy(), g[f'features{bi}'])
        assert tree.count_nodes() == int(g[f'count{bi}'])
        assert tree.count_leaf_nodes() == int(g[f'leaves{bi}'])
    has = [tree.has_voxel(torch.from_numpy(q)) for q in g['query']]
    assert np.array_equal(np.array(has), g['has'])
    assert np.array_equal(tree.get_voxels().numpy(), g['get_voxels'])
    assert np.array_equal(tree.get_leaf_voxels().numpy(),
                          g['get_leaf_voxels'])
    # pickle round trip rebuilds the identical tree (ids restart at 0)
    tree2 = pickle.loads(pickle.dumps(tree))
    v2, c2, f2 = tree2.get_centres_and_children()
    assert np.array_equal(v2.numpy(), g['voxels2'])
    assert np.array_equal(c2.numpy(), g['children2'])
    assert np.array_equal(f2.numpy(), g['features2'])


def test_octree_edge_cases():
    svo.reset_id_counter()
    tree = svo.Octree()
    tree.init(256, 16, 0.2)
    vox, ch, ft = tree.get_centres_and_children()  # empty tree: root only
    assert vox.shape == (1, 4) and float(vox[0, 3]) == 256.0
    assert (ch == -1).all() and (ft == -1).all()
    tree.insert(torch.zeros(0, 3, dtype=torch.int32))  # empty batch
    assert tree.count_nodes() == 1
    tree.insert(torch.tensor([[254, 254, 254]], dtype=torch.int32))  # top corner
    assert tree.count_leaf_nodes() == 1
    with pytest.raises(RuntimeError):
        tree.insert(torch.zeros(2, 3, dtype=torch.int64))  # wrong dtype
    n0 = tree.count_nodes()
    tree.insert(torch.tensor([[254, 254, 254]] * 5, dtype=torch.int32))
    assert tree.count_nodes() == n0  # idempotent
