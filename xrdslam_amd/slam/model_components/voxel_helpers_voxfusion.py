"""Voxel helpers of Vox-Fusion (reference:
slam/model_components/voxel_helpers_voxfusion.py): ray / sparse-voxel-octree
intersection, inverse-CDF sampling inside the hit voxels, trilinear voxel
features.  The two native operators come from ``xrdslam_amd.compat.grid``
(HIP: ``xrd_svo_intersect``, ``xrd_inverse_cdf_sampling``; indices bit-exact
with the reference's CUDA kernels via the C oracle).

Differences from the reference's Python wrappers that do not change results:
the intersection is one launch over all rays with the octree shared (the
reference replicates the octree G<=256 times and reshapes the rays to
[G, N/G] to bound memory, :233-278), and the sampler runs in one call instead
of chunks of 800 rows (:441-452)."""
import numpy as np
import torch
import torch.nn.functional as F

from ...compat import grid as _ext

MAX_DEPTH = 10.0


def ray(ray_start, ray_dir, depths):
    return ray_start + ray_dir * depths


def masked_scatter(mask, x):
    """values of the valid samples back into the padded [rays, samples] grid,
    zeros elsewhere"""
    B, K = mask.shape
    if x.dim() == 1:
        return x.new_zeros(B, K).masked_scatter(mask, x)
    return x.new_zeros(B, K, x.size(-1)).masked_scatter(
        mask.unsqueeze(-1).expand(B, K, x.size(-1)), x)


def masked_scatter_ones(mask, x):
    """same, padded with ONES (an sdf of 1 = free space)"""
    B, K = mask.shape
    if x.dim() == 1:
        return x.new_ones(B, K).masked_scatter(mask, x)
    return x.new_ones(B, K, x.size(-1)).masked_scatter(
        mask.unsqueeze(-1).expand(B, K, x.size(-1)), x)


def offset_points(point_xyz, quarter_voxel=1, offset_only=False, bits=2):
    """the 8 corner offsets (+-1 per axis for bits=2), x slowest"""
    c = torch.arange(1, 2 * bits, 2, device=point_xyz.device)
    ox, oy, oz = torch.meshgrid([c, c, c], indexing='ij')
    offset = (torch.stack([ox.reshape(-1), oy.reshape(-1), oz.reshape(-1)], 1)
              .type_as(point_xyz) - bits) / float(bits - 1)
    if offset_only:
        return offset * quarter_voxel
    return point_xyz.unsqueeze(1) + offset.unsqueeze(0) * quarter_voxel


@torch.enable_grad()
def trilinear_interp(p, q, point_feats):
    """p [N,1,3] local coordinate in [0,1], q [1,8,3] corner selectors"""
    w = (p * q + (1 - p) * (1 - q)).prod(dim=-1, keepdim=True)
    if point_feats.dim() == 2:
        point_feats = point_feats.view(point_feats.size(0), 8, -1)
    return (w * point_feats).sum(1)


@torch.enable_grad()
def get_embeddings(sampled_xyz, point_xyz, point_feats, voxel_size):
    p = ((sampled_xyz - point_xyz) / voxel_size + 0.5).unsqueeze(1)
    q = offset_points(p, 0.5, offset_only=True).unsqueeze(0) + 0.5
    return trilinear_interp(p, q, point_feats).float()


@torch.enable_grad()
def get_features(samples, map_states, voxel_size):
    """trilinear feature of every valid sample: voxel id -> 8 vertex ids ->
    embedding rows (:109-123)"""
    dev = map_states['voxel_vertex_emb'].device
    vertex_idx = map_states['voxel_vertex_idx'].to(dev)
    centres = map_states['voxel_center_xyz'].to(dev)
    values = map_states['voxel_vertex_emb']
    sampled_idx = samples['sampled_point_voxel_idx'].long()
    sampled_xyz = samples['sampled_point_xyz'].requires_grad_(True)
    point_xyz = F.embedding(sampled_idx, centres)
    point_feats = F.embedding(F.embedding(sampled_idx, vertex_idx),
                              values).view(point_xyz.size(0), -1)
    feats = get_embeddings(sampled_xyz, point_xyz, point_feats, voxel_size)
    return {'dists': samples['sampled_point_distance'], 'emb': feats}


@torch.no_grad()
def svo_ray_intersect(voxel_size, n_max, points, children, ray_start,
                      ray_dir):
    """points [1,Nv,3] voxel centres, children [1,Nv,9], rays [1,N,3] ->
    (idx i32, min_depth, max_depth) [1,N,n_max]"""
    idx, mn, mx = _ext.svo_intersect(ray_start.float().contiguous(),
                                     ray_dir.float().contiguous(),
                                     points.float().contiguous(),
                                     children.int().contiguous(),
                                     voxel_size, n_max)
    return idx, mn.type_as(ray_start), mx.type_as(ray_start)


@torch.no_grad()
def ray_intersect(ray_start, ray_dir, flatten_centers, flatten_children,
                  voxel_size, max_hits, max_distance=10.0):
    """hits sorted by entry depth, beyond ``max_distance`` dropped, trimmed to
    the largest hit count (:647-687)"""
    pts_idx, min_depth, max_depth = svo_ray_intersect(
        voxel_size, 50, flatten_centers, flatten_children, ray_start, ray_dir)
    miss = pts_idx.eq(-1)
    min_depth.masked_fill_(miss, max_distance)
    max_depth.masked_fill_(miss, max_distance)
    min_depth, order = min_depth.sort(dim=-1)
    max_depth = max_depth.gather(-1, order)
    pts_idx = pts_idx.gather(-1, order)
    pts_idx[min_depth > max_distance] = -1
    miss = pts_idx.eq(-1)
    min_depth.masked_fill_(miss, max_distance)
    max_depth.masked_fill_(miss, max_distance)
    n_hit = int(torch.max(pts_idx.ne(-1).sum(-1)))
    out = {'min_depth': min_depth[..., :n_hit],
           'max_depth': max_depth[..., :n_hit],
           'intersected_voxel_idx': pts_idx[..., :n_hit]}
    return out, out['intersected_voxel_idx'].ne(-1).any(-1)


def _uniform_noise(shape, like):
    """the sampler's pre-drawn noise (``noise.uniform_()``, :432-437);
    replaceable for parity tests"""
    return like.new_zeros(shape).uniform_()


@torch.no_grad()
def inverse_cdf_sampling(pts_idx, min_depth, max_depth, probs, steps,
                         fixed_step_size=-1, deterministic=False,
                         noise_fn=_uniform_noise):
    """[N,P] hits -> padded samples [N,S] (idx -1 = padding).  The noise
    tensor keeps the reference's [200, ceil(N/200), max_steps] shape so that a
    seeded run draws the same numbers for the same rays (:407-437)."""
    G, N, P = 200, pts_idx.size(0), pts_idx.size(1)
    H = int(np.ceil(N / G)) * G
    if H > N:
        def pad(t):
            return torch.cat([t, t[:1].expand(H - N, *t.shape[1:])], 0)
        pts_idx, min_depth, max_depth, probs, steps = (
            pad(t) for t in (pts_idx, min_depth, max_depth, probs, steps))
    max_steps = int(steps.ceil().long().max()) + P
    if deterministic:
        noise = min_depth.new_full((G, H // G, max_steps), 0.5)
    else:
        noise = noise_fn((G, H // G, max_steps), min_depth).clamp(
            min=0.001, max=0.999)
    sidx, sdepth, sdist = _ext.inverse_cdf_sampling(
        pts_idx.reshape(G, -1, P).int().contiguous(),
        min_depth.reshape(G, -1, P).float().contiguous(),
        max_depth.reshape(G, -1, P).float().contiguous(),
        noise.float().contiguous(),
        probs.reshape(G, -1, P).float().contiguous(),
        steps.reshape(G, -1).float().contiguous(), fixed_step_size)
    sidx = sidx.reshape(H, -1)[:N]
    sdepth = sdepth.reshape(H, -1)[:N].type_as(min_depth)
    sdist = sdist.reshape(H, -1)[:N].type_as(min_depth)
    max_len = int(sidx.ne(-1).sum(-1).max())
    return sidx[:, :max_len], sdepth[:, :max_len], sdist[:, :max_len]


@torch.no_grad()
def ray_sample(intersection_outputs, step_size=0.01, fixed=False,
               noise_fn=_uniform_noise):
    """samples proportional to the chord length inside each hit voxel
    (:690-714)"""
    io = intersection_outputs
    dists = (io['max_depth'] - io['min_depth']).masked_fill(
        io['intersected_voxel_idx'].eq(-1), 0)
    io['probs'] = dists / dists.sum(dim=-1, keepdim=True)
    io['steps'] = dists.sum(-1) / step_size
    sidx, sdepth, sdist = inverse_cdf_sampling(
        io['intersected_voxel_idx'], io['min_depth'], io['max_depth'],
        io['probs'], io['steps'], -1, fixed, noise_fn=noise_fn)
    sdist = sdist.clamp(min=0.0)
    pad = sidx.eq(-1)
    sdepth.masked_fill_(pad, MAX_DEPTH)
    sdist.masked_fill_(pad, 0.0)
    return {'sampled_point_depth': sdepth, 'sampled_point_distance': sdist,
            'sampled_point_voxel_idx': sidx}
