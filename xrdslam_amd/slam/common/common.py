"""Pixel -> ray sampling and keyframe overlap selection (the A1/A2 rows of
SURVEY.md §8a; reference: slam/common/common.py:39-122,188-227,288-310,342-426).

Same functions and argument meaning as the reference; differences are
MI355X-side plumbing only: frames keep a device-resident copy of depth/colour
(the reference re-uploads the whole image per call, common.py:67-68), the pixel
meshgrid is computed arithmetically from the drawn index instead of being
materialised, and the overlap test for all keyframes runs as one batched device
computation.  Random draws use the global torch / numpy RNGs like the
reference, or an explicit ``generator`` when the caller wants reproducibility.
"""
from __future__ import annotations

import numpy as np
import torch


def get_rays_from_uv(i, j, c2w, fx, fy, cx, cy, device):
    """rays for pixel columns ``i`` and rows ``j`` (common.py:39-53): OpenGL
    camera, direction NOT normalised, gradient flows to ``c2w``."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    c2w = c2w.to(device)
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)],
                       -1).to(device)
    if c2w.is_cuda and c2w.requires_grad and c2w.dtype == torch.float32 and \
            dirs.dim() == 2:
        # one launch each way (xrd_pose_rays_*) instead of the broadcast
        # multiply / sum / expand chain and its backward (whose pose-gradient
        # tail also does not survive hipGraph capture on ROCm 7.2)
        from ...engine import slam_ops
        ids = torch.zeros(dirs.shape[0], dtype=torch.int64, device=dirs.device)
        return slam_ops.PoseRaysFn.apply(c2w.unsqueeze(0),
                                         dirs.float().contiguous(), ids)
    rays_d = (dirs.reshape(-1, 1, 3) * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def _device_images(depth, color, device, frame=None):
    if frame is not None:
        return frame.device_images(device)
    d = torch.as_tensor(depth, dtype=torch.float32).to(device).reshape(-1, 1)
    c = torch.as_tensor(color, dtype=torch.float32).to(device).reshape(-1, 3)
    return d, c


def get_sample_uv(H0, H1, W0, W1, n, depth, color, device='cuda:0',
                  full_width=None, frame=None, generator=None):
    """draw n pixels (with replacement) from rows [H0,H1) x cols [W0,W1)
    (common.py:109-122 + select_uv :56-71).  Index k of the cropped row-major
    meshgrid is pixel (row H0 + k // (W1-W0), col W0 + k % (W1-W0))."""
    w = W1 - W0
    cnt = (H1 - H0) * w
    idx = torch.randint(cnt, (n, ), device=device, generator=generator)
    rows = H0 + torch.div(idx, w, rounding_mode='floor')
    cols = W0 + idx % w
    W = full_width if full_width is not None else (
        depth.shape[1] if hasattr(depth, 'shape') else None)
    d, c = _device_images(depth, color, device, frame)
    flat = rows * W + cols
    return cols.float(), rows.float(), d[flat], c[flat]


def masked_lower_median(x, valid):
    """torch.median (the LOWER median) of x[valid] without compaction: sort
    with the masked-out entries sent to +inf and read entry (count - 1) // 2
    through a device-side index (no size read-back; NaN for an empty set)"""
    flat = torch.where(valid, x, torch.full_like(x, float('inf'))).reshape(-1)
    vals = torch.sort(flat).values
    cnt = valid.sum()
    k = torch.clamp((cnt - 1) // 2, min=0)
    med = vals.gather(0, k.reshape(1)).reshape(())
    return torch.where(cnt > 0, med, torch.full_like(med, float('nan')))


def get_samples(camera, n, c2w, depth, color, device, Hedge=0, Wedge=0,
                depth_filter=False, return_index=False, depth_limit=None,
                frame=None, generator=None):
    """n rays of one frame (common.py:188-227); ``frame`` (optional) supplies
    the cached device images."""
    i, j, sd, sc = get_sample_uv(Hedge, camera.height - Hedge, Wedge,
                                 camera.width - Wedge, n, depth, color,
                                 device=device, full_width=camera.width,
                                 frame=frame, generator=generator)
    rays_o, rays_d = get_rays_from_uv(i, j, c2w, camera.fx, camera.fy,
                                      camera.cx, camera.cy, device)
    if depth_filter:
        sd = sd.reshape(-1)
        mask = sd > 0
        if depth_limit is not None:
            mask = mask & (sd < depth_limit)
        # one compaction index (one size read-back) for all six tensors
        keep = torch.nonzero(mask).reshape(-1)
        if keep.numel() != mask.numel():
            rays_o, rays_d, sd, sc, i, j = (t.index_select(0, keep) for t in (
                rays_o, rays_d, sd, sc, i, j))
    if return_index:
        return rays_o, rays_d, sd, sc, i.to(torch.int64), j.to(torch.int64)
    return rays_o, rays_d, sd, sc


def rgb2gray(image):
    """luma of an RGB image (skimage.color.rgb2gray weights)"""
    return image[..., 0] * 0.2125 + image[..., 1] * 0.7154 + \
        image[..., 2] * 0.0721


def _sobel(intensity, axis):
    """skimage.filters.sobel_h (axis 0) / sobel_v (axis 1): difference
    [1,0,-1] along ``axis``, smoothing [1,2,1]/4 across it, reflected borders.
    Restated from the published definition (skimage is not installed here, so
    parity with skimage itself is unpinned; the magnitude is checked against
    scipy.ndimage.convolve with the published kernel in
    tests/test_reference_host_parity.py).  It only steers WHICH pixels are
    sampled and the per-pixel radii, not the render arithmetic."""
    a = np.asarray(intensity, dtype=np.float64)
    p = np.pad(a, 1, mode='symmetric')
    if axis == 0:
        d = p[:-2, :] - p[2:, :]
        return (d[:, :-2] + 2 * d[:, 1:-1] + d[:, 2:]) / 4.0
    d = p[:, :-2] - p[:, 2:]
    return (d[:-2, :] + 2 * d[1:-1, :] + d[2:, :]) / 4.0


def color_gradient_magnitude(image):
    gray = rgb2gray(np.asarray(image))
    return np.sqrt(_sobel(gray, 1)**2 + _sobel(gray, 0)**2)


def get_sample_uv_with_grad(H0, H1, W0, W1, n, image, ratio=15):
    """n flat pixel indices drawn (without replacement) from the ratio*n
    pixels with the largest colour gradient that fall inside the crop
    (common.py:74-106)"""
    image = np.asarray(image.cpu() if torch.is_tensor(image) else image)
    grad_mag = color_gradient_magnitude(image)
    size = (image.shape[0], image.shape[1])
    top = np.argpartition(grad_mag, -ratio * n, axis=None)[-ratio * n:]
    h, w = np.unravel_index(top, size)
    m = (h >= H0) & (h < H1) & (w >= W0) & (w < W1)
    h, w = h[m], w[m]
    flat = np.ravel_multi_index(np.array((h, w)), size)
    return flat[np.random.choice(range(0, h.shape[0]), size=n, replace=False)]


def get_samples_with_pixel_grad(camera, n_color, c2w, depth, color, device,
                                Hedge=0, Wedge=0, depth_filter=True,
                                return_index=True, depth_limit=None):
    """rays through the pixels with the strongest colour gradients
    (common.py:230-285)"""
    assert n_color > 0, 'invalid number of rays to sample.'
    H, W = camera.height, camera.width
    idx = np.union1d(get_sample_uv_with_grad(Hedge, H - Hedge, Wedge,
                                             W - Wedge, n_color, color), [])
    rows, cols = np.unravel_index(idx.astype(int), (H, W))
    i = torch.from_numpy(cols).to(device).float()
    j = torch.from_numpy(rows).to(device).float()
    rays_o, rays_d = get_rays_from_uv(i, j, c2w.to(device), camera.fx,
                                      camera.fy, camera.cx, camera.cy, device)
    i, j = i.long(), j.long()
    depth = torch.as_tensor(depth).to(device)
    color = torch.as_tensor(color).to(device)
    sd, sc = depth[j, i].reshape(-1), color[j, i].reshape(-1, 3)
    if depth_filter:
        m = sd > 0
        if depth_limit is not None:
            m = m & (sd < depth_limit)
        rays_o, rays_d, sd, sc, i, j = rays_o[m], rays_d[m], sd[m], sc[m], \
            i[m], j[m]
    if return_index:
        return rays_o, rays_d, sd, sc, i.to(torch.int64), j.to(torch.int64)
    return rays_o, rays_d, sd, sc


def get_rays(camera, c2w, device):
    """rays of the full image, [H,W,3] each (common.py:288-310)"""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    c2w = c2w.to(device)
    cols = torch.arange(camera.width, device=device, dtype=torch.float32)
    rows = torch.arange(camera.height, device=device, dtype=torch.float32)
    j, i = torch.meshgrid(rows, cols, indexing='ij')
    dirs = torch.stack([(i - camera.cx) / camera.fx,
                        -(j - camera.cy) / camera.fy, -torch.ones_like(i)], -1)
    rays_d = (dirs[..., None, :] * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def _pointcloud_rows(depth, camera, c2w, sampled_indices):
    """-> (world points [n,3] of the pixels ``sampled_indices`` [n,2] =
    (row, col), camera looking down +z; keep [n]: False for the rows the
    reference drops).  The reference appends the origin to
    |round(points, 4)|, takes ``unique(dim=0)`` with counts and drops every row
    whose rounded value occurs more than once (common.py:313-339): i.e. rows
    that coincide with the camera origin or with ANOTHER sampled row.  The same
    set comes out of an all-pairs comparison of the n <= 1600 rows — three
    elementwise launches; ``unique(dim=0)`` (a lexicographic device sort),
    ``round(decimals=)`` and ``isin`` cost ~20 ms EACH per call on the GPU
    (measured: 44 ms of a 95 ms SplaTAM frame went into this selection)."""
    xx = (sampled_indices[:, 1] - camera.cx) / camera.fx
    yy = (sampled_indices[:, 0] - camera.cy) / camera.fy
    z = depth[sampled_indices[:, 0], sampled_indices[:, 1]]
    pts_cam = torch.stack((xx * z, yy * z, z), -1)
    pts4 = torch.cat([pts_cam, torch.ones_like(pts_cam[:, :1])], 1)
    pts = (c2w.to(pts4) @ pts4.T).T[:, :3]
    # torch.round(pts, decimals=4) = nearbyint(x * 1e4) / 1e4
    rounded = torch.abs(torch.round(pts * 1e4) / 1e4)
    same = (rounded[:, None, :] == rounded[None, :, :]).all(-1)
    dup = (same.sum(1) > 1) | (rounded == 0).all(1)
    return pts, ~dup


def get_pointcloud(depth, camera, c2w, sampled_indices):
    """world points of the pixels ``sampled_indices`` [n,2] = (row, col),
    camera looking down +z; points that coincide with the camera origin or
    with another sampled point (rounded to 1e-4) are dropped
    (common.py:313-339)"""
    pts, keep = _pointcloud_rows(depth, camera, c2w, sampled_indices)
    return pts[keep]


def setup_camera(camera, w2c, near=0.01, far=100, device='cuda'):
    """rasteriser settings of SplaTAM for the FIRST frame's w2c
    (common.py:592-619): transposed view matrix and full projection"""
    from ...compat.diff_gaussian_rasterization import \
        GaussianRasterizationSettings
    w, h = camera.width, camera.height
    fx, fy, cx, cy = camera.fx, camera.fy, camera.cx, camera.cy
    w2c = torch.as_tensor(np.asarray(w2c), dtype=torch.float32).to(device)
    cam_center = torch.inverse(w2c)[:3, 3]
    w2c_t = w2c.unsqueeze(0).transpose(1, 2)
    proj = torch.tensor(
        [[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
         [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
         [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
         [0.0, 0.0, 1.0, 0.0]], dtype=torch.float32,
        device=device).unsqueeze(0).transpose(1, 2)
    return GaussianRasterizationSettings(
        image_height=h, image_width=w, tanfovx=w / (2 * fx),
        tanfovy=h / (2 * fy),
        bg=torch.zeros(3, dtype=torch.float32, device=device),
        scale_modifier=1.0, viewmatrix=w2c_t, projmatrix=w2c_t.bmm(proj),
        sh_degree=0, campos=cam_center, prefiltered=False)


@torch.no_grad()
def keyframe_selection_overlap(camera, cur_frame, keyframes_graph, k,
                               N_samples=16, pixs_per_image=100,
                               use_ray_sample=True, device='cuda:0'):
    """keyframes whose view contains part of the current frustum, then a random
    k of them (common.py:342-426, ray-sample branch).  Points: 16 samples in
    [0.8 d, d+0.5] along 100 valid-depth rays; a keyframe counts a point when
    it projects >20 px inside the image and lies in front of the camera."""
    if len(keyframes_graph) == 0:
        return []
    H, W = camera.height, camera.width
    keep = None
    if use_ray_sample:
        rays_o, rays_d, gd, _ = get_samples(camera, pixs_per_image,
                                            cur_frame.get_pose(),
                                            cur_frame.depth, cur_frame.rgb,
                                            device, depth_filter=True,
                                            frame=cur_frame)
        gd = gd.reshape(-1, 1).repeat(1, N_samples)
        t = torch.linspace(0., 1., steps=N_samples, device=device)
        z = gd * 0.8 * (1. - t) + (gd + 0.5) * t
        pts = (rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]) \
            .reshape(-1, 3)
    else:
        # SplaTAM branch (:373-387): back-projected valid-depth pixels, camera
        # looking down +z, no x flip
        # (the frame's device-resident depth; the dropped rows travel as a
        # mask: no compaction, no size read-back)
        if torch.device(device).type == 'cuda':
            depth = cur_frame.device_images(device)[0].reshape(H, W)
        else:
            depth = torch.as_tensor(cur_frame.depth).to(device)
        valid = torch.stack(torch.where(depth > 0), 1)
        pick = valid[torch.randint(valid.shape[0],
                                   (pixs_per_image * N_samples, )).to(device)]
        pts, keep = _pointcloud_rows(depth, camera,
                                     cur_frame.get_pose().to(device), pick)
    c2ws = torch.stack([kf.get_pose().detach().to(device)
                        for kf in keyframes_graph])
    if c2ws.is_cuda:
        # rigid inverse [R^T | -R^T t] in f64 (three elementwise launches): the
        # poses come out of OptimizablePose (unit quaternion / axis-angle), so
        # it equals the LU inverse to ~1e-7 — far inside the 20 px margin of
        # the test below — while torch.linalg.inv on the device is a batched
        # LU with a status read-back (several ms per mapping call)
        c64 = c2ws.double()
        Rt = c64[:, :3, :3].transpose(1, 2)
        w2c = torch.zeros_like(c64)
        w2c[:, :3, :3] = Rt
        w2c[:, :3, 3] = -(Rt @ c64[:, :3, 3:4]).squeeze(-1)
        w2c[:, 3, 3] = 1.0
    else:
        w2c = torch.linalg.inv(c2ws.double())
    homo = torch.cat([pts.double(), torch.ones_like(pts[:, :1]).double()], 1)
    cam = torch.einsum('kij,nj->kni', w2c, homo)[..., :3]
    if use_ray_sample:
        cam[..., 0] *= -1  # x flip: pixel u grows to the right
    u = camera.fx * cam[..., 0] + camera.cx * cam[..., 2]
    v = camera.fy * cam[..., 1] + camera.cy * cam[..., 2]
    zc = cam[..., 2] + 1e-5
    u, v = (u / zc).float(), (v / zc).float()
    edge = 20
    inside = (u < W - edge) & (u > edge) & (v < H - edge) & (v > edge) & \
        ((zc < 0) if use_ray_sample else (zc > 0))
    if keep is None:
        percent = inside.float().mean(1)
    else:   # the mean over the kept rows
        percent = (inside & keep).float().sum(1) / keep.float().sum()
    percent = percent.cpu().numpy()
    order = np.argsort(-percent, kind='stable')
    selected = [keyframes_graph[a] for a in order if percent[a] > 0.0]
    perm = np.random.permutation(len(selected))[:k]
    return [selected[a] for a in perm]
