"""Run in a SUBPROCESS by tests/test_reference_dropin.py: the REFERENCE's own
``JointEncoding`` (Co-SLAM) and ``SparseVoxel`` (Vox-Fusion) — imported from
/root/reference, unmodified — EXECUTE forward, losses and backward on top of
``xrdslam_amd.compat`` on the CPU, with the C-ABI's compute entry points
served by the host backend of tests/host_abi.py, and must reproduce the
committed goldens (which the HIP kernels are checked against on the GPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch

import ref_harness

ref_harness.install()
_zeros = torch.zeros


def zeros(*a, **k):     # the reference hard-codes device='cuda' in places
    if str(k.get('device', '')).startswith('cuda'):
        k['device'] = 'cpu'
    return _zeros(*a, **k)


torch.zeros = zeros
for name in ('tinycudann', 'grid', 'faiss', 'diff_gaussian_rasterization'):
    sys.modules.pop(name, None)          # the harness' mocks give way
from xrdslam_amd import compat

compat.install()
import host_abi
import coslam_golden_util as cg

out = {}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


with host_abi.installed() as abi:
    # ---- Co-SLAM: slam/models/joint_encoding.py on compat.tinycudann ---------
    import tinycudann
    assert tinycudann.__name__ == 'xrdslam_amd.compat.tinycudann'
    from slam.common.camera import Camera
    from slam.models.joint_encoding import JointEncoding, JointEncodingConfig
    g = np.load(cg.GOLDEN)
    cfg = JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True,
                              hashsize=int(g['hash_cfg'][1]),
                              trainging_smooth_pts=8)
    model = JointEncoding(cfg, Camera(40., 40., 31.5, 23.5, 64, 48),
                          torch.from_numpy(g['bound']))
    assert type(model.embed_fn).__module__ == 'xrdslam_amd.compat.tinycudann'
    model.decoder.load_state_dict({k[4:]: torch.from_numpy(g[k])
                                   for k in g.files if k.startswith('dec/')})
    with torch.no_grad():
        model.embed_fn.params.copy_(torch.from_numpy(g['hash_params']))
    real_rand = torch.rand
    worst = {}
    for tag, is_mapping, first in cg.TAGS:
        draws = iter([torch.from_numpy(g[f'{tag}/rand{i}'])
                      for i in range(int(g[f'{tag}/n_rand']))])

        def fed(*shape, **kw):
            return next(draws).clone()
        torch.rand = fed
        try:
            for p in model.parameters():
                p.grad = None
            ro = torch.from_numpy(g['rays_o']).clone().requires_grad_(True)
            rd = torch.from_numpy(g['rays_d']).clone().requires_grad_(True)
            inp = {'rays_o': ro, 'rays_d': rd, 'first': first,
                   'target_s': torch.from_numpy(g['target_s']),
                   'target_d': torch.from_numpy(g['target_d'])}
            res = model.get_outputs(inp)
            ld = model.get_loss_dict(res, inp, is_mapping, 0)
            sum(ld.values()).backward()
        finally:
            torch.rand = real_rand
        errs = {k: rel(res[k].detach().numpy(), g[f'{tag}/{k}'])
                for k in ('rgb', 'depth', 'depth_var', 'acc_map', 'z_vals',
                          'raw')}
        for k, v in ld.items():
            errs[f'loss_{k}'] = rel(v.detach().numpy(), g[f'{tag}/loss_{k}'])
        errs['g_rays_o'] = rel(ro.grad.numpy(), g[f'{tag}/g_rays_o'])
        errs['g_rays_d'] = rel(rd.grad.numpy(), g[f'{tag}/g_rays_d'])
        errs['g_hash'] = rel(model.embed_fn.params.grad.numpy(),
                             g[f'{tag}/g_hash'])
        for k, p in model.decoder.named_parameters():
            errs[f'g_dec/{k}'] = rel(p.grad.numpy(), g[f'{tag}/g_dec/{k}'])
        k = max(errs, key=errs.get)
        worst[tag] = [k, errs[k]]
    out['coslam_worst'] = worst
    out['coslam_calls'] = sorted(set(abi.calls))
    abi.calls.clear()

    # ---- Vox-Fusion: slam/models/sparse_voxel.py on torch.classes.svo (the
    # TorchScript seam over the C-ABI octree) and compat.grid -------------------
    import grid as grid_mod
    assert grid_mod.__name__ == 'xrdslam_amd.compat.grid'
    import voxfusion_golden_util as vg
    from slam.models.sparse_voxel import SparseVoxel, SparseVoxelConfig
    from xrdslam_amd.compat import svo as _svo
    g = np.load(vg.GOLDEN)
    _svo.reset_id_counter()
    torch.manual_seed(0)
    vox = SparseVoxel(SparseVoxelConfig(num_embeddings=6000),
                      Camera(40., 40., 31.5, 23.5, 64, 48), None)
    assert type(vox.svo).__name__ == 'ScriptObject' or \
        'svo' in str(type(vox.svo))
    with torch.no_grad():
        vox.embeddings.copy_(torch.from_numpy(g['embeddings']))
    vox.decoder.load_state_dict({k[4:]: torch.from_numpy(g[k])
                                 for k in g.files if k.startswith('dec/')})
    vox.insert_points(torch.from_numpy(g['points']))
    ms = vox.map_states
    verr = {f'map/{k}': rel(ms[k].numpy(), g[f'map/{k}'])
            for k in ('voxel_vertex_idx', 'voxel_center_xyz',
                      'voxel_structure')}
    noise = torch.from_numpy(g['noise'])
    real_uniform = torch.Tensor.uniform_

    def fed_uniform(self, *a, **k):
        assert self.shape == noise.shape
        with torch.no_grad():
            self.copy_(noise)
        return self
    torch.Tensor.uniform_ = fed_uniform
    try:
        ro = torch.from_numpy(g['rays_o']).clone().requires_grad_(True)
        rd = torch.from_numpy(g['rays_d']).clone().requires_grad_(True)
        inp = {'rays_o': ro, 'rays_d': rd,
               'target_s': torch.from_numpy(g['target_s']),
               'target_d': torch.from_numpy(g['target_d'])}
        res = vox.get_outputs(inp)
        ld = vox.get_loss_dict(res, inp, True, 0)
        sum(ld.values()).backward()
    finally:
        torch.Tensor.uniform_ = real_uniform
    for k in ('depth', 'rgb', 'sdf', 'z_vals', 'ray_mask', 'weights', 'z_min'):
        verr[f'out/{k}'] = rel(res[k].detach().numpy().astype(np.float64),
                               g[f'out/{k}'].astype(np.float64))
    for k, v in ld.items():
        verr[f'loss/{k}'] = rel(v.detach().numpy(), g[f'loss/{k}'])
    verr['g_rays_o'] = rel(ro.grad.numpy(), g['g_rays_o'])
    verr['g_rays_d'] = rel(rd.grad.numpy(), g['g_rays_d'])
    verr['g_embeddings'] = rel(vox.embeddings.grad.numpy(), g['g_embeddings'])
    for k, p in vox.decoder.named_parameters():
        verr[f'g_dec/{k}'] = rel(p.grad.numpy(), g[f'g_dec/{k}'])
    k = max(verr, key=verr.get)
    out['vox_worst'] = [k, verr[k]]
    out['vox_rays_hit'] = int(res['ray_mask'].sum())
    out['vox_calls'] = sorted(set(abi.calls))
    abi.calls.clear()

    # ---- Point-SLAM: slam/models/conv_onet_pointslam.py + neural_point_cloud
    # on compat.faiss — the golden's own generation procedure
    # (oracle/make_golden_pointslam.py: two frames of point insertion, renders
    # in both stages, tracking and mapping losses, all gradients), with the
    # faiss stand-in swapped for the shim; everything it would have written is
    # compared with the committed file
    import faiss as faiss_mod
    assert faiss_mod.__name__ == 'xrdslam_amd.compat.faiss'
    import make_golden_pointslam as mg
    import pointslam_golden_util as pgu
    mg.faiss_standin.module = lambda: faiss_mod
    captured = {}
    real_save = np.savez_compressed
    np.savez_compressed = lambda path, **kw: captured.update(kw)
    try:
        mg.main()
    finally:
        np.savez_compressed = real_save
    g = np.load(pgu.GOLDEN)
    assert sorted(captured) == sorted(g.files)
    perr = {}
    for k in g.files:
        a, b = np.asarray(captured[k]), g[k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        if b.dtype == bool or np.issubdtype(b.dtype, np.integer):
            perr[k] = float((a != b).sum())
        else:
            perr[k] = rel(a, b)
    k = max(perr, key=perr.get)
    out['point_worst'] = [k, perr[k]]
    out['point_keys'] = len(perr)
    out['point_cloud'] = [int(captured['add0/cloud'].shape[0]),
                          int(captured['add1/cloud'].shape[0])]
    out['point_calls'] = sorted(set(abi.calls))
print('DROPIN_EXEC ' + json.dumps(out))
