"""GPU parity of the fused Vox-Fusion voxel-feature + decoder kernels
(xrd_vox_points_fwd / _bwd, engine/vox.py) against the modular torch path
(get_features + Decoder, themselves checked against the reference-made golden
in tests/test_voxfusion_hip.py / test_voxfusion_host.py): sdf, colour and every
gradient (positions, embeddings, ten decoder tensors), 1e-4 max-norm and
element-wise."""
import numpy as np
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu


def _case(P, seed, n_vox=400, n_emb=3000):
    from xrdslam_amd.slam.model_components.decoder_voxfusion import Decoder
    g = torch.Generator().manual_seed(seed)
    dev = 'cuda:0'
    voxel_size = 0.2
    centres = (torch.randint(40, 90, (n_vox, 3), generator=g).float() +
               0.5) * voxel_size
    vertex_idx = torch.randint(0, n_emb, (n_vox, 8), generator=g).int()
    emb = (torch.randn(n_emb, 16, generator=g) * 0.3)
    vox = torch.randint(0, n_vox, (P, ), generator=g).int()
    xyz = centres[vox.long()] + (torch.rand(P, 3, generator=g) - 0.5) * \
        voxel_size
    torch.manual_seed(seed)
    dec = Decoder(depth=2, width=128, in_dim=16, embedder='none').to(dev)
    ms = {'voxel_vertex_idx': vertex_idx.to(dev),
          'voxel_center_xyz': centres.to(dev),
          'voxel_vertex_emb': emb.to(dev).requires_grad_(True)}
    return dec, xyz.to(dev), vox.to(dev), ms, voxel_size


@pytest.mark.parametrize('P', [1, 17, 128, 5000, 40001])
def test_points_match_modular_path(P):
    from xrdslam_amd.engine import vox as ev
    from xrdslam_amd.slam.model_components import voxel_helpers_voxfusion as vh
    dec, xyz, vox, ms, vs = _case(P, 10 + P)
    g = torch.Generator().manual_seed(1)
    w_s = torch.randn(P, generator=g).cuda()
    w_c = torch.randn(P, 3, generator=g).cuda()

    def run(fused):
        for p in dec.parameters():
            p.grad = None
        ms['voxel_vertex_emb'].grad = None
        x = xyz.clone().requires_grad_(True)
        if fused:
            out = ev.points(dec, x, vox, ms, vs)
            assert out is not None
        else:
            out = dec(vh.get_features(
                {'sampled_point_xyz': x, 'sampled_point_voxel_idx': vox,
                 'sampled_point_distance': None}, ms, vs))
        ((out['sdf'] * w_s).sum() + (out['color'] * w_c).sum()).backward()
        res = {'sdf': out['sdf'].detach(), 'color': out['color'].detach(),
               'g_xyz': x.grad, 'g_emb': ms['voxel_vertex_emb'].grad.clone()}
        for n, p in dec.named_parameters():
            res['g_' + n] = p.grad.clone()
        return res

    ref, got = run(False), run(True)
    # A ReLU whose pre-activation is within rounding of zero may come out on
    # different sides in two float32 evaluations (rocBLAS GEMM vs the MFMA
    # chain): the outputs stay continuous, that point's gradient does not.
    # Such points are isolated (1 of 40 001 x 384 units here); they are found
    # by their position gradient, taken out of the loss in BOTH paths, and
    # everything else must then agree to 1e-4.
    sc = ref['g_xyz'].abs().max()
    dev = (got['g_xyz'] - ref['g_xyz']).abs().max(1).values / sc
    flipped = torch.nonzero(dev > 1e-4).flatten()
    assert flipped.numel() <= max(1, int(5e-4 * P)), flipped.numel()
    if flipped.numel():
        assert float(dev.max()) < 2e-2
        w_s[flipped] = 0
        w_c[flipped] = 0
        ref, got = run(False), run(True)
    parity.assert_all([(f'vox_points/P={P}/{k}', got[k], ref[k])
                       for k in ref])


def test_points_without_decoder_gradients():
    """tracking: only the positions carry a gradient; nothing is saved for
    the weight-gradient GEMMs"""
    from xrdslam_amd.engine import vox as ev
    from xrdslam_amd.slam.model_components import voxel_helpers_voxfusion as vh
    dec, xyz, vox, ms, vs = _case(3000, 5)
    for p in dec.parameters():
        p.requires_grad_(False)
    ms['voxel_vertex_emb'] = ms['voxel_vertex_emb'].detach()
    a = xyz.clone().requires_grad_(True)
    out = ev.points(dec, a, vox, ms, vs)
    (out['sdf'].sum() + out['color'].sum()).backward()
    b = xyz.clone().requires_grad_(True)
    ref = dec(vh.get_features({'sampled_point_xyz': b,
                               'sampled_point_voxel_idx': vox,
                               'sampled_point_distance': None}, ms, vs))
    (ref['sdf'].sum() + ref['color'].sum()).backward()
    parity.assert_all([('vox_points/track/sdf', out['sdf'], ref['sdf']),
                       ('vox_points/track/g_xyz', a.grad, b.grad)])


def test_unsupported_decoder_falls_back():
    from xrdslam_amd.engine import vox as ev
    from xrdslam_amd.slam.model_components.decoder_voxfusion import Decoder
    dec = Decoder(depth=3, width=128, in_dim=16, embedder='none').cuda()
    assert ev.decoder_params(dec) is None
