"""TEST INFRASTRUCTURE ONLY.  Vox-Fusion golden vectors: executes the
REFERENCE's own ``SparseVoxel`` (slam/models/sparse_voxel.py, imported from
/root/reference) on the CPU — octree = the reference's sparse_octree sources
compiled by oracle/build_ref_octree.py, ``grid`` = oracle/grid_standin.py (C
oracle of the two CUDA kernels) — on a small synthetic scene and stores
inputs, the sampler's recorded noise, outputs, loss terms and gradients in
tests/golden/voxfusion_render.npz.

    python oracle/make_golden_voxfusion.py            # the small case
    python oracle/make_golden_voxfusion.py office0    # BASELINE configs[2]

``office0``: the shapes of the reference's vox-fusion configuration
(slam/configs/input_config.py:158-197: 1024 mapping rays, default
SparseVoxelConfig = 0.2 m voxels, 20000 embeddings) on a 640x480 camera inside
a room whose visible surfaces give > 800 leaf voxels and ragged rows of up to
~100 samples.  Inputs and the sampler's noise are regenerated from seeds on
both sides (tests/voxfusion_golden_util.office0_inputs); the file holds the
reference's outputs -> tests/golden/voxfusion_office0.npz.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import build_ref_octree  # noqa: E402
import grid_standin  # noqa: E402
import ref_harness  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def scene_points(seed=0):
    """a wall + floor patch around (10.6, 10.3, 9.0) m, i.e. inside the
    octree's positive octant like a relative-pose Vox-Fusion run"""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(6000, 2, generator=g)
    wall = torch.stack([10.0 + 2.0 * u[:3000, 0], 9.4 + 1.8 * u[:3000, 1],
                        torch.full((3000, ), 8.05)], -1)
    floor = torch.stack([10.0 + 2.0 * u[3000:, 0], torch.full((3000, ), 9.35),
                         8.05 + 1.5 * u[3000:, 1]], -1)
    return torch.cat([wall, floor]).float()


def main():
    ref_harness.install()
    sys.modules['grid'] = grid_standin.module()
    build_ref_octree.load()
    # the reference allocates its embedding table with device='cuda'
    real_zeros = torch.zeros

    def zeros_cpu(*a, **k):
        if k.get('device') == 'cuda':
            k['device'] = 'cpu'
        return real_zeros(*a, **k)

    torch.zeros = zeros_cpu
    import slam.model_components.voxel_helpers_voxfusion as vh
    vh._ext = sys.modules['grid']
    from slam.common.camera import Camera
    from slam.models.sparse_voxel import SparseVoxel, SparseVoxelConfig

    torch.manual_seed(0)
    model = SparseVoxel(SparseVoxelConfig(num_embeddings=6000),
                        Camera(40., 40., 31.5, 23.5, 64, 48), None)
    torch.zeros = real_zeros
    with torch.no_grad():
        model.embeddings.normal_(0, 0.3)
    pts = scene_points()
    model.insert_points(pts)
    out = {'points': pts.numpy(), 'embeddings': model.embeddings.detach()
           .numpy().copy()}
    for k, v in model.decoder.state_dict().items():
        out[f'dec/{k}'] = v.numpy().copy()
    ms = model.map_states
    out['map/voxel_vertex_idx'] = ms['voxel_vertex_idx'].numpy()
    out['map/voxel_center_xyz'] = ms['voxel_center_xyz'].numpy()
    out['map/voxel_structure'] = ms['voxel_structure'].numpy()

    # rays from a camera 1.2 m in front of the wall, looking at it (-z)
    g = torch.Generator().manual_seed(5)
    n = 300
    o = torch.tensor([11.0, 10.2, 9.3]) + 0.02 * torch.randn(n, 3, generator=g)
    d = torch.stack([(torch.rand(n, generator=g) - 0.5) * 1.6,
                     (torch.rand(n, generator=g) - 0.5) * 1.2,
                     -torch.ones(n)], -1)
    d[:20, 2] = 1.0  # some rays miss everything
    depth = (o[:, 2] - 8.05).clamp(min=0.2)[:, None] * \
        (1 + 0.02 * torch.randn(n, 1, generator=g))
    depth[40:60] = 0.0  # invalid sensor depth
    color = torch.rand(n, 3, generator=g)
    out.update(rays_o=o.numpy(), rays_d=d.numpy(), target_d=depth.numpy(),
               target_s=color.numpy())

    draws = []
    gen = torch.Generator().manual_seed(11)
    real_uniform = torch.Tensor.uniform_

    def rec_uniform(self, *a, **k):
        real_uniform(self, *a, generator=gen, **k)
        draws.append(self.clone())
        return self

    torch.Tensor.uniform_ = rec_uniform
    try:
        ro = o.clone().requires_grad_(True)
        rd = d.clone().requires_grad_(True)
        inp = {'rays_o': ro, 'rays_d': rd, 'target_s': color,
               'target_d': depth}
        res = model.get_outputs(inp)
        ld = model.get_loss_dict(res, inp, True, 0)
        sum(ld.values()).backward()
    finally:
        torch.Tensor.uniform_ = real_uniform
    assert len(draws) == 1
    out['noise'] = draws[0].numpy()
    for k in ('depth', 'rgb', 'sdf', 'z_vals', 'ray_mask', 'weights', 'z_min'):
        out[f'out/{k}'] = res[k].detach().numpy()
    for k, v in ld.items():
        out[f'loss/{k}'] = v.detach().numpy()
    out['g_rays_o'] = ro.grad.numpy()
    out['g_rays_d'] = rd.grad.numpy()
    out['g_embeddings'] = model.embeddings.grad.numpy().copy()
    for k, p in model.decoder.named_parameters():
        out[f'g_dec/{k}'] = p.grad.numpy().copy()
    path = os.path.join(GOLD, 'voxfusion_render.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB;',
          int(res['ray_mask'].sum()), 'of', n, 'rays hit;',
          tuple(res['z_vals'].shape), 'samples;',
          {k: float(v) for k, v in ld.items()})


def office0_inputs():
    """seeded inputs of the office0-shaped case (shared with the tests through
    tests/voxfusion_golden_util.py)"""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import voxfusion_golden_util as vg
    return vg.office0_inputs()


def main_office0():
    ref_harness.install()
    sys.modules['grid'] = grid_standin.module()
    build_ref_octree.load()
    real_zeros = torch.zeros

    def zeros_cpu(*a, **k):
        if k.get('device') == 'cuda':
            k['device'] = 'cpu'
        return real_zeros(*a, **k)

    torch.zeros = zeros_cpu
    import slam.model_components.voxel_helpers_voxfusion as vh
    vh._ext = sys.modules['grid']
    from slam.common.camera import Camera
    from slam.models.sparse_voxel import SparseVoxel, SparseVoxelConfig
    inp = office0_inputs()
    torch.manual_seed(0)
    model = SparseVoxel(SparseVoxelConfig(), Camera(*inp['cam']), None)
    torch.zeros = real_zeros
    with torch.no_grad():
        model.embeddings.copy_(inp['embeddings'])
    model.insert_points(inp['points'])
    ms = model.map_states
    out = {f'dec/{k}': v.numpy().copy()
           for k, v in model.decoder.state_dict().items()}
    out.update({'map/voxel_vertex_idx': ms['voxel_vertex_idx'].numpy(),
           'map/voxel_center_xyz': ms['voxel_center_xyz'].numpy(),
           'map/voxel_structure': ms['voxel_structure'].numpy()})
    leaves = int(ms['voxel_vertex_idx'].shape[0])
    draws = []
    gen = torch.Generator().manual_seed(inp['noise_seed'])
    real_uniform = torch.Tensor.uniform_

    def rec_uniform(self, *a, **k):
        real_uniform(self, *a, generator=gen, **k)
        draws.append(self.clone())
        return self

    torch.Tensor.uniform_ = rec_uniform
    try:
        ro = inp['rays_o'].clone().requires_grad_(True)
        rd = inp['rays_d'].clone().requires_grad_(True)
        minp = {'rays_o': ro, 'rays_d': rd, 'target_s': inp['target_s'],
                'target_d': inp['target_d']}
        res = model.get_outputs(minp)
        ld = model.get_loss_dict(res, minp, True, 0)
        sum(ld.values()).backward()
    finally:
        torch.Tensor.uniform_ = real_uniform
    assert len(draws) == 1
    out['noise_shape'] = np.array(draws[0].shape)
    out['noise_sum'] = np.float64(draws[0].double().sum())
    for k in ('depth', 'rgb', 'z_vals', 'ray_mask', 'weights', 'z_min'):
        out[f'out/{k}'] = res[k].detach().numpy()
    for k, v in ld.items():
        out[f'loss/{k}'] = v.detach().numpy()
    out['g_rays_o'] = ro.grad.numpy()
    out['g_rays_d'] = rd.grad.numpy()
    ge = model.embeddings.grad
    rows = ge.abs().sum(1).nonzero().reshape(-1)
    out['g_embeddings/rows'] = rows.numpy()
    out['g_embeddings/vals'] = ge[rows].numpy().copy()
    for k, p in model.decoder.named_parameters():
        out[f'g_dec/{k}'] = p.grad.numpy().copy()
    path = os.path.join(GOLD, 'voxfusion_office0.npz')
    np.savez_compressed(path, **out)
    zs = res['z_vals']
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB;', leaves,
          'leaf voxels;', int(res['ray_mask'].sum()), 'of',
          ro.shape[0], 'rays hit;', tuple(zs.shape), 'samples (padded);',
          {k: float(v) for k, v in ld.items()})


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'office0':
        main_office0()
    else:
        main()
