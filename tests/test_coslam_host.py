"""CPU: the Co-SLAM host mirror (JointEncoding: sampling, SDF weights,
rendering, losses, smoothness) against the reference-generated golden, with
the oracle encodings standing in for the HIP encodings (the same stand-in the
reference itself ran on when the golden was made), so that everything above
the encodings is checked without a GPU."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
sys.path.insert(0, os.path.dirname(__file__))
import coslam_golden_util as cg  # noqa: E402

TOL = 1e-4


@pytest.fixture()
def oracle_encodings(monkeypatch):
    import tcnn_standin
    import xrdslam_amd.slam.model_components.encodings_coslam as enc
    monkeypatch.setattr(enc, 'tcnn', tcnn_standin.module())


@pytest.mark.parametrize('tag,is_mapping,first', cg.TAGS)
def test_joint_encoding_vs_reference(oracle_encodings, tag, is_mapping, first):
    g = np.load(cg.GOLDEN)
    model = cg.build_model(g, 'cpu')
    errs = cg.run_case(model, g, tag, is_mapping, first, 'cpu')
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad


@pytest.mark.parametrize('tag,is_mapping,first,n', cg.OFFICE0_CASES)
def test_joint_encoding_at_baseline_config(oracle_encodings, tag, is_mapping,
                                           first, n):
    """host mirror == the reference's JointEncoding at its DEFAULT config
    (2^16 table, office0 bound, 1024 / 2389 rays), max-norm and element-wise"""
    import parity
    gold = np.load(cg.OFFICE0)
    model = cg.build_office0_model('cpu')
    assert model.embed_fn.params.numel() == int(gold['n_params'])
    assert model.resolution_sdf == int(gold['resolution_sdf'])
    got = cg.run_office0_case(model, tag, is_mapping, first, n, 'cpu')
    parity.assert_all(cg.office0_pairs(got, gold, tag))


@pytest.mark.parametrize('tag,is_mapping', cg.VARIANT_TAGS)
@pytest.mark.parametrize('name', list(cg.VARIANTS))
def test_non_default_model_options_vs_reference(oracle_encodings, name, tag,
                                                is_mapping):
    """oneGrid=False (second, colour-only grid + ColorSDFNet) and
    training_n_importance>0 (inverse-CDF second pass; perturbed and
    deterministic) against the reference's own model, element-wise"""
    import parity
    g = np.load(cg.VARIANT_GOLDEN)
    model, grids = cg.build_variant(g, name, 'cpu')
    parity.assert_all(cg.run_variant(model, grids, g, name, tag, is_mapping,
                                     'cpu'))


def test_fixed_shape_depth_loss_equals_compacted(oracle_encodings):
    g = np.load(cg.GOLDEN)
    model = cg.build_model(g, 'cpu')
    model.fixed_shape_losses = True
    errs = cg.run_case(model, g, 'track', False, False, 'cpu')
    assert errs['loss_depth_loss'] < TOL and errs['g_hash'] < TOL


def test_param_groups_and_config(oracle_encodings):
    from xrdslam_amd.slam.configs.input_config import (algorithm_configs,
                                                       cadence)
    cfg = algorithm_configs['co-slam']()
    assert cfg.model.tcnn_encoding and cfg.model.enc == 'HashGrid'
    assert cfg.tracking_n_iters == 10 and cfg.mapping_sample == 2048
    assert cadence['co-slam'].map_every == 5
    g = np.load(cg.GOLDEN)
    model = cg.build_model(g, 'cpu')
    groups = model.get_param_groups()
    assert set(groups) == {'decoder', 'embed_fn'}      # oneGrid default
    assert set(groups) <= set(cfg.optimizers)


def test_fused_pack_and_gradient_index_tables():
    """slot-space layout of the fused renderer (csrc/coslam_layout.h): every
    decoder weight feeds exactly one lane of the forward fragments and one
    lane of the transposed (backward) fragments, and has exactly one slot in
    the weight-gradient buffer"""
    from xrdslam_amd import _lib
    lib = _lib.lib()
    flat_len, pack_len = lib.xrd_coslam_flat_len(), lib.xrd_coslam_pack_len()
    assert flat_len == 32 * 63 + 3 * 32 + 32 * 80 + 16 * 32
    pack = np.zeros(pack_len, np.int32)
    dw = np.zeros(flat_len, np.int32)
    assert lib.xrd_coslam_index(pack.ctypes.data, dw.ctypes.data) == 0
    half = pack_len // 2
    for part in (pack[:half], pack[half:]):
        used = part[part >= 0]
        assert np.array_equal(np.sort(used), np.arange(flat_len))
    assert len(np.unique(dw)) == flat_len
    assert dw.min() >= 0 and dw.max() < lib.xrd_coslam_dw_len()


def test_batched_axis_angle_pose_equals_per_frame_module():
    """the stacked bundle-adjustment poses evaluate OptimizablePose.matrix()
    for all frames at once: same values and gradients (incl. the exact
    identity below 1e-8 rad)"""
    from xrdslam_amd.slam.utils.opt_pose import (
        OptimizablePose, axis_angle_translation_to_matrix)
    g = torch.Generator().manual_seed(0)
    rot = torch.randn(6, 3, generator=g) * 0.7
    rot[2] = 0.0
    rot[4] *= 1e-3
    trans = torch.randn(6, 3, generator=g)
    w = torch.randn(6, 4, 4, generator=g)
    rb, tb = rot.clone().requires_grad_(True), trans.clone().requires_grad_(True)
    Mb = axis_angle_translation_to_matrix(rb, tb)
    (Mb * w).sum().backward()
    for i in range(6):
        pose = OptimizablePose(torch.cat([trans[i], rot[i]]), separate_LR=True)
        M = pose.matrix()
        (M * w[i]).sum().backward()
        assert torch.allclose(M, Mb[i], atol=1e-6)
        assert torch.allclose(pose.data_r.grad, rb.grad[i], atol=1e-5)
        assert torch.allclose(pose.data_t.grad, tb.grad[i], atol=1e-6)
