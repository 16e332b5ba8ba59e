"""Drives the host mirror of Vox-Fusion's SparseVoxel on the inputs of
tests/golden/voxfusion_render.npz (made by oracle/make_golden_voxfusion.py from
the reference's own model + compiled reference octree)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden',
                      'voxfusion_render.npz')


GOLDEN_OFFICE0 = os.path.join(os.path.dirname(__file__), 'golden',
                              'voxfusion_office0.npz')
OFFICE0_CAM = (320., 320., 319.5, 239.5, 640, 480)


def office0_inputs():
    """seeded inputs of the office0-shaped golden (BASELINE configs[2]: 1024
    rays of a 640x480 camera, default SparseVoxelConfig): a room corner seen
    from inside — two walls, floor and a table-height box top, 60 000 surface
    points ~ 1300 leaf voxels of 0.2 m — in the octree's positive octant
    (init_pose_offset 10 m, input_config.py:165).  Used by
    oracle/make_golden_voxfusion.py AND by the tests: the npz stores outputs
    only."""
    g = torch.Generator().manual_seed(100)
    n = 15000
    u = torch.rand(4, n, 2, generator=g)
    wall_a = torch.stack([8.0 + 5.0 * u[0, :, 0], 8.6 + 2.8 * u[0, :, 1],
                          torch.full((n, ), 7.03)], -1)          # z = 7.03
    wall_b = torch.stack([torch.full((n, ), 7.97), 8.6 + 2.8 * u[1, :, 1],
                          7.0 + 4.5 * u[1, :, 0]], -1)           # x = 7.97
    floor = torch.stack([8.0 + 5.0 * u[2, :, 0], torch.full((n, ), 8.57),
                         7.0 + 4.5 * u[2, :, 1]], -1)            # y = 8.57
    box = torch.stack([9.4 + 1.5 * u[3, :, 0], torch.full((n, ), 9.33),
                       8.0 + 1.2 * u[3, :, 1]], -1)              # table top
    points = torch.cat([wall_a, wall_b, floor, box]).float()
    fx, fy, cx, cy, W, H = OFFICE0_CAM
    n_rays = 1024
    pix = torch.randint(0, W * H, (n_rays, ), generator=g)
    col, row = (pix % W).float(), (pix // W).float()
    d_cam = torch.stack([(col - cx) / fx, -(row - cy) / fy,
                         -torch.ones(n_rays)], -1)
    # camera at (11.2, 10.1, 10.3) looking towards the corner (-x, -z), a bit
    # downwards: rotation about y by 40 degrees, then about x by -12 degrees
    a, b = np.deg2rad(40.0), np.deg2rad(-12.0)
    Ry = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0],
                       [-np.sin(a), 0, np.cos(a)]], dtype=torch.float32)
    Rx = torch.tensor([[1, 0, 0], [0, np.cos(b), -np.sin(b)],
                       [0, np.sin(b), np.cos(b)]], dtype=torch.float32)
    R = Ry @ Rx
    rays_d = d_cam @ R.T
    rays_o = torch.tensor([11.2, 10.1, 10.3]).expand(n_rays, 3).contiguous()
    # sensor depth = distance along -z_cam to the nearest of the four planes
    t_all = []
    for axis, val in ((2, 7.03), (0, 7.97), (1, 8.57)):
        t = (val - rays_o[:, axis]) / rays_d[:, axis]
        t_all.append(torch.where(t > 0, t, torch.full_like(t, 1e9)))
    t_box = (9.33 - rays_o[:, 1]) / rays_d[:, 1]
    hit = rays_o + rays_d * t_box[:, None]
    on_box = (t_box > 0) & (hit[:, 0] > 9.4) & (hit[:, 0] < 10.9) & \
        (hit[:, 2] > 8.0) & (hit[:, 2] < 9.2)
    t_all.append(torch.where(on_box, t_box, torch.full_like(t_box, 1e9)))
    t = torch.stack(t_all).min(0).values
    depth = (t * (1 + 0.01 * torch.randn(n_rays, generator=g)))[:, None]
    depth[torch.rand(n_rays, generator=g) < 0.05] = 0.0   # invalid sensor px
    depth = depth.clamp(max=9.5).float()
    color = torch.rand(n_rays, 3, generator=g)
    emb = torch.randn(20000, 16, generator=g) * 0.3
    return {'cam': OFFICE0_CAM, 'points': points, 'rays_o': rays_o.float(),
            'rays_d': rays_d.float(), 'target_d': depth, 'target_s': color,
            'embeddings': emb, 'noise_seed': 11}


def office0_noise(shape, seed=11):
    """the sampler's uniform draws of the office0 golden, regenerated (CPU
    generator: the same stream as when the golden was made)"""
    gen = torch.Generator().manual_seed(seed)
    return torch.empty(tuple(int(x) for x in shape)).uniform_(generator=gen)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


class Golden:
    """uniform view of the two goldens: the small one stores its inputs, the
    office0-shaped one regenerates them from seeds (office0_inputs)"""

    def __init__(self, path=GOLDEN):
        self.z = np.load(path)
        self.office0 = 'noise_shape' in self.z.files
        self.inp = office0_inputs() if self.office0 else None

    def __getitem__(self, k):
        if self.office0 and k in ('points', 'rays_o', 'rays_d', 'target_d',
                                  'target_s', 'embeddings'):
            return self.inp[k].numpy()
        if self.office0 and k == 'noise':
            n = office0_noise(self.z['noise_shape'], self.inp['noise_seed'])
            assert abs(float(n.double().sum()) - float(self.z['noise_sum'])) \
                < 1e-6, 'CPU generator stream differs from the golden run'
            return n.numpy()
        if self.office0 and k == 'g_embeddings':
            full = np.zeros((20000, 16), np.float32)
            full[self.z['g_embeddings/rows']] = self.z['g_embeddings/vals']
            return full
        return self.z[k]

    @property
    def files(self):
        return self.z.files

    @property
    def cam(self):
        return OFFICE0_CAM if self.office0 else (40., 40., 31.5, 23.5, 64, 48)


def as_golden(g):
    return g if isinstance(g, Golden) else _Wrapped(g)


class _Wrapped(Golden):
    def __init__(self, z):
        self.z, self.office0, self.inp = z, False, None


def build_model(g, device):
    from xrdslam_amd.compat import svo
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.models.sparse_voxel import (SparseVoxel,
                                                      SparseVoxelConfig)
    g = as_golden(g)
    svo.reset_id_counter()  # node ids come from a process-global counter
    emb = g['embeddings']
    model = SparseVoxel(SparseVoxelConfig(num_embeddings=emb.shape[0]),
                        Camera(*g.cam), None)
    model.decoder.load_state_dict(
        {k[4:]: torch.from_numpy(g[k]) for k in g.files
         if k.startswith('dec/')})
    model = model.to(device)
    with torch.no_grad():
        model.embeddings.copy_(torch.from_numpy(emb))
    return model


def run(model, g, device, dedup):
    g = as_golden(g)
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
        if f'out/{k}' in g.files:
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
