"""Drives the host mirror of Vox-Fusion's SparseVoxel on the inputs of
tests/golden/voxfusion_render.npz (made by oracle/make_golden_voxfusion.py from
the reference's own model + compiled reference octree)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden',
                      'voxfusion_render.npz')


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def build_model(g, device):
    from xrdslam_amd.compat import svo
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.models.sparse_voxel import (SparseVoxel,
                                                      SparseVoxelConfig)
    svo.reset_id_counter()  # node ids come from a process-global counter
    model = SparseVoxel(SparseVoxelConfig(num_embeddings=g['embeddings']
                                          .shape[0]),
                        Camera(40., 40., 31.5, 23.5, 64, 48), None)
    model.decoder.load_state_dict(
        {k[4:]: torch.from_numpy(g[k]) for k in g.files
         if k.startswith('dec/')})
    model = model.to(device)
    with torch.no_grad():
        model.embeddings.copy_(torch.from_numpy(g['embeddings']))
    return model


def run(model, g, device, dedup):
    model.insert_points(torch.from_numpy(g['points']).to(device), dedup=dedup)
    ms = model.map_states
    exact = {k: bool(np.array_equal(ms[k].cpu().numpy(), g[f'map/{k}']))
             for k in ('voxel_vertex_idx', 'voxel_structure')}
    exact['voxel_center_xyz'] = bool(np.array_equal(
        ms['voxel_center_xyz'].cpu().numpy(), g['map/voxel_center_xyz']))
    noise = torch.from_numpy(g['noise'])

    def fed(shape, like):
        assert tuple(shape) == tuple(noise.shape), (shape, noise.shape)
        return noise.to(like)

    model.noise_fn = fed
    ro = torch.from_numpy(g['rays_o']).to(device).requires_grad_(True)
    rd = torch.from_numpy(g['rays_d']).to(device).requires_grad_(True)
    inp = {'rays_o': ro, 'rays_d': rd,
           'target_s': torch.from_numpy(g['target_s']).to(device),
           'target_d': torch.from_numpy(g['target_d']).to(device)}
    res = model.get_outputs(inp)
    ld = model.get_loss_dict(res, inp, True, 0)
    sum(ld.values()).backward()
    errs = {}
    exact['ray_mask'] = bool(np.array_equal(res['ray_mask'].cpu().numpy(),
                                            g['out/ray_mask']))
    assert tuple(res['z_vals'].shape) == g['out/z_vals'].shape
    for k in ('depth', 'rgb', 'sdf', 'z_vals', 'weights', 'z_min'):
        errs[k] = rel_err(res[k].detach().cpu().numpy(), g[f'out/{k}'])
    for k, v in ld.items():
        errs[f'loss_{k}'] = rel_err(v.detach().cpu().numpy(), g[f'loss/{k}'])
    errs['g_rays_o'] = rel_err(ro.grad.cpu().numpy(), g['g_rays_o'])
    errs['g_rays_d'] = rel_err(rd.grad.cpu().numpy(), g['g_rays_d'])
    errs['g_embeddings'] = rel_err(model.embeddings.grad.cpu().numpy(),
                                   g['g_embeddings'])
    for k, p in model.decoder.named_parameters():
        errs[f'g_dec/{k}'] = rel_err(p.grad.cpu().numpy(), g[f'g_dec/{k}'])
    return exact, errs
