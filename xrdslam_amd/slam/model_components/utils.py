"""Loss helpers shared by Co-SLAM / Vox-Fusion (reference:
slam/model_components/utils.py:10-28,100-186)."""
import torch
import torch.nn.functional as F


def coordinates(voxel_dim, device, flatten=True):
    if isinstance(voxel_dim, int):
        nx = ny = nz = voxel_dim
    else:
        nx, ny, nz = voxel_dim
    ax = [torch.arange(0, n, dtype=torch.long, device=device)
          for n in (nx, ny, nz)]
    x, y, z = torch.meshgrid(*ax, indexing='ij')
    if not flatten:
        return torch.stack([x, y, z], dim=-1)
    return torch.stack((x.flatten(), y.flatten(), z.flatten()))


def compute_loss(prediction, target, loss_type='l2'):
    if loss_type == 'l2':
        return F.mse_loss(prediction, target)
    if loss_type == 'l1':
        return F.l1_loss(prediction, target)
    raise Exception('Unsupported loss type')


def get_masks(z_vals, target_d, truncation):
    """free-space (in front of the truncation band) and SDF (inside the band,
    valid depth) sample masks with batch-global balancing weights"""
    front = (z_vals < (target_d - truncation)).to(z_vals.dtype)
    back = (z_vals > (target_d + truncation)).to(z_vals.dtype)
    valid = (target_d > 0.0).to(target_d.dtype)
    sdf_mask = (1.0 - front) * (1.0 - back) * valid
    n_fs = torch.count_nonzero(front)
    n_sdf = torch.count_nonzero(sdf_mask)
    total = n_sdf + n_fs
    return front, sdf_mask, 1.0 - n_fs / total, 1.0 - n_sdf / total


def get_sdf_loss(z_vals, target_d, predicted_sdf, truncation, loss_type=None,
                 grad=None):
    front, sdf_mask, fs_w, sdf_w = get_masks(z_vals, target_d, truncation)
    fs_loss = compute_loss(predicted_sdf * front,
                           torch.ones_like(predicted_sdf) * front,
                           loss_type) * fs_w
    sdf_loss = compute_loss((z_vals + predicted_sdf * truncation) * sdf_mask,
                            target_d * sdf_mask, loss_type) * sdf_w
    if grad is not None:
        eik = (((grad.norm(2, dim=-1) - 1)**2) * sdf_mask /
               sdf_mask.sum()).sum()
        return fs_loss, sdf_loss, eik
    return fs_loss, sdf_loss


def _exclusive_cumprod(x, ones):
    """[1, x0, x0 x1, ...] along the last axis (= cumprod(cat([1, x]))[:, :-1]).
    For the few samples per ray of Point-SLAM the running product is written
    out: torch.cumprod's backward reads a flag back to the host (zero check),
    which a captured hipGraph cannot do; same products in the same order."""
    S = x.shape[-1]
    if S > 16:
        return torch.cumprod(torch.cat([ones, x], -1), dim=-1)[:, :-1]
    cols = [ones[:, 0]]
    for k in range(S - 1):
        cols.append(cols[-1] * x[:, k])
    return torch.stack(cols, -1)


def raw2outputs_nerf_color2(raw, z_vals, rays_d, device='cuda:0', coef=0.1):
    """Point-SLAM compositing (reference: slam/model_components/utils.py:
    247-294): occupancy alpha = sigmoid(coef * logit) (written back into
    ``raw`` like the reference does), transmittance weights, colour and depth
    NORMALISED by the weight sum, depth variance around the rendered depth."""
    rgb = raw[..., :-1]
    raw[..., -1] = torch.sigmoid(coef * raw[..., -1])
    alpha = raw[..., -1]
    ones = torch.ones((alpha.shape[0], 1), device=alpha.device).float()
    weights = alpha.float() * _exclusive_cumprod(
        (1. - alpha + 1e-10).float(), ones)
    wsum = torch.sum(weights, dim=-1).unsqueeze(-1) + 1e-10
    rgb_map = torch.sum(weights[..., None] * rgb, -2) / wsum
    depth_map = torch.sum(weights * z_vals, -1) / wsum.squeeze(-1)
    tmp = z_vals - depth_map.unsqueeze(-1)
    depth_var = torch.sum(weights * tmp * tmp, dim=1)
    return depth_map, depth_var, rgb_map, weights
