"""Synthetic Replica-shaped RGB-D sequences: an analytic
axis-aligned box room plus three spheres, exact ray-cast z-depth in metres,
smooth procedural colour, 2 % invalid depth pixels, and a smooth Lissajous
trajectory looking at the room centre.  Poses are OpenGL-convention
camera-to-world matrices like slam/common/datasets.py:163-164 produces.
Replaces the file-based datasets (out of scope) for tests and benchmarks."""
from __future__ import annotations

import numpy as np
import torch


def look_at(eye, target, up=(0.0, 0.0, 1.0)):
    """OpenGL c2w: camera looks along -z, +y up, +x right"""
    eye, target, up = (np.asarray(a, dtype=np.float64)
                       for a in (eye, target, up))
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    true_up = np.cross(right, fwd)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, true_up, -fwd, eye
    return c2w


class SyntheticRoom:
    def __init__(self, bound, H=480, W=640, fx=320.0, fy=320.0, cx=319.5,
                 cy=239.5, n_frames=200, seed=0, shrink=0.5,
                 invalid_frac=0.02, device='cpu'):
        self.device = torch.device(device)
        self.H, self.W = H, W
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.n_frames, self.seed, self.invalid_frac = n_frames, seed, \
            invalid_frac
        b = np.asarray(bound, dtype=np.float64)
        self.lo, self.hi = b[:, 0] + shrink, b[:, 1] - shrink
        c = 0.5 * (self.lo + self.hi)
        ext = self.hi - self.lo
        self.center = c
        self.spheres = [(c + ext * np.array([0.25, 0.2, -0.3]), 0.12 * ext.min()),
                        (c + ext * np.array([-0.3, 0.1, -0.25]), 0.10 * ext.min()),
                        (c + ext * np.array([0.1, -0.3, -0.35]), 0.08 * ext.min())]
        self.poses = [self._pose(k) for k in range(n_frames)]
        self._cache = {}

    def _pose(self, k):
        """<= ~2 cm / ~1 deg per frame at 200 frames"""
        s = np.pi * k / max(self.n_frames, 200)
        ext = self.hi - self.lo
        eye = self.center + ext * np.array([0.04 * np.sin(s),
                                            0.03 * np.sin(2 * s + 0.5),
                                            0.02 * np.sin(3 * s)])
        tgt = self.center + ext * np.array([0.15 * np.cos(s), 0.15 * np.sin(s),
                                            -0.1])
        return look_at(eye, tgt).astype(np.float32)

    def __len__(self):
        return self.n_frames

    def _raycast(self, c2w):
        """exact z-depth (ray parameter of the un-normalised OpenGL ray) and
        hit points, float64 torch on ``self.device``"""
        dev, H, W = self.device, self.H, self.W
        f64 = torch.float64
        j, i = torch.meshgrid(torch.arange(H, dtype=f64, device=dev),
                              torch.arange(W, dtype=f64, device=dev),
                              indexing='ij')
        dirs = torch.stack([(i - self.cx) / self.fx, -(j - self.cy) / self.fy,
                            -torch.ones_like(i)], -1)
        c2w = torch.as_tensor(c2w, dtype=f64, device=dev)
        R, o = c2w[:3, :3], c2w[:3, 3]
        d = dirs @ R.T
        hi = torch.as_tensor(self.hi, dtype=f64, device=dev)
        lo = torch.as_tensor(self.lo, dtype=f64, device=dev)
        t_wall = torch.where(d > 0, (hi - o) / d, (lo - o) / d)
        t_wall = torch.where(torch.isfinite(t_wall) & (t_wall > 0), t_wall,
                             torch.full_like(t_wall, float('inf')))
        t = t_wall.min(-1)[0]
        a = (d * d).sum(-1)
        for c, r in self.spheres:
            oc = o - torch.as_tensor(c, dtype=f64, device=dev)
            bq = (d * oc).sum(-1)
            cq = (oc * oc).sum() - r * r
            disc = bq * bq - a * cq
            ts = (-bq - torch.sqrt(disc.clamp(min=0))) / a
            t = torch.where((disc > 0) & (ts > 0) & (ts < t), ts, t)
        pts = o + d * t[..., None]
        return t, pts

    def preload(self, indices, device_images=True):
        """generate frames ahead of time (what a prefetching dataset loader
        does) and, on a GPU, keep their float32 images resident in HBM: items
        then also carry 'depth_dev' [H*W,1] / 'rgb_dev' [H*W,3], which
        SequentialSLAM hands to the Frame as its device image cache"""
        for k in indices:
            if k in self._cache or not 0 <= k < self.n_frames:
                continue
            item = self._make(k)
            if device_images and self.device.type == 'cuda':
                item['depth_dev'] = torch.from_numpy(item['depth']).to(
                    self.device).reshape(-1, 1)
                item['rgb_dev'] = torch.from_numpy(item['rgb']).to(
                    self.device).reshape(-1, 3)
            self._cache[k] = item
        return self

    def __getitem__(self, k):
        hit = self._cache.get(k)
        return dict(hit) if hit is not None else self._make(k)

    def _make(self, k):
        c2w = self.poses[k]
        depth, pts = self._raycast(c2w)
        x, y, z = pts[..., 0], pts[..., 1], pts[..., 2]
        rgb = torch.stack([0.5 + 0.45 * torch.sin(1.3 * x + 0.7 * y),
                           0.5 + 0.45 * torch.sin(1.1 * y - 0.9 * z + 1.0),
                           0.5 + 0.45 * torch.sin(0.8 * z + 1.7 * x + 2.0)],
                          -1)
        rng = np.random.default_rng(self.seed * 100003 + k)
        depth = depth.float().cpu().numpy()
        depth[rng.random(depth.shape) < self.invalid_frac] = 0.0
        return {'index': k, 'rgb': rgb.float().cpu().numpy(), 'depth': depth,
                'c2w': c2w.copy()}
