"""GPU: Vox-Fusion on the HIP ray/voxel operators.  (1) SparseVoxel
(xrd_svo_intersect + xrd_inverse_cdf_sampling through compat.grid, host octree,
device-side voxel de-duplication) against the golden made from the reference's
own model: map arrays and ray mask bit-exact, outputs / losses / gradients at
1e-4; (2) a short VoxFusion tracking+mapping run on the synthetic room."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import voxfusion_golden_util as vg  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize('dedup', [True, False])
def test_sparse_voxel_vs_reference(dedup):
    g = np.load(vg.GOLDEN)
    model = vg.build_model(g, 'cuda:0')
    exact, errs = vg.run(model, g, 'cuda:0', dedup=dedup)
    assert all(exact.values()), exact
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad


def test_voxfusion_loop_tracks_synthetic_room():
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import (cadence,
                                                       voxfusion_config)
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    bound = [[-3, 3], [-4, 2.5], [-2, 2.5]]
    cam = Camera(fx=150., fy=150., cx=79.5, cy=59.5, width=160, height=120)
    cfg = voxfusion_config()
    algo = cfg.setup(camera=cam, device='cuda:0')
    data = SyntheticRoom(bound, H=120, W=160, fx=150., fy=150., cx=79.5,
                         cy=59.5, n_frames=200, device='cuda:0')
    cad = cadence['vox-fusion']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device='cuda:0',
                          use_relative_pose=cad.use_relative_pose,
                          init_pose_offset=cad.init_pose_offset)
    for k in range(8):
        slam.step(k)
    # first pose = identity + 10 m offset (tracker.py:76-89)
    first = algo.get_gt_c2w_list()[0]
    assert torch.allclose(first[:3, 3], torch.full((3, ), 10.0))
    assert algo.model.svo.count_leaf_nodes() > 50
    ate = slam.ate_rmse()
    assert ate < 0.03, ate
    with torch.no_grad():
        _, depth = algo.render_img(algo.get_estimate_c2w_list()[6].to(
            'cuda:0'), gt_depth=None)
    gt = data[6]['depth']
    gt = gt.cpu().numpy() if torch.is_tensor(gt) else np.asarray(gt)
    hit = depth > 0
    assert hit.mean() > 0.5
    assert np.abs(depth - gt)[hit & (gt > 0)].mean() < 0.15
