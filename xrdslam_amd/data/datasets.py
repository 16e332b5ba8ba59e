"""File datasets + a prefetching device loader.

Reference: slam/common/datasets.py:60-134 (``BaseDataset``: devices.yaml,
image decoding, crop, down-sampling, intrinsics after crop/down-sampling),
:140-167 (``Replica``), :461-552 (``TUM_RGBD``) and the ingest side of
slam/pipeline/tracker.py:54-102 (torch DataLoader with one worker; every frame
crosses host -> device once per ITERATION in the reference, common.py:67-68).

Here a frame is decoded on the host (PIL instead of OpenCV, which is not in
this image: own bilinear / nearest / undistortion resamplers with OpenCV's
pixel-centre conventions), staged in pinned memory and copied to HBM ONCE on a
side stream while the previous frame is being tracked (``Prefetcher``); items
carry ``depth_dev`` / ``rgb_dev`` which ``SequentialSLAM`` hands to the Frame
as its device image cache.

Items: ``{'index', 'rgb' f32 [H,W,3] in [0,1], 'depth' f32 [H,W] metres,
'c2w' f64 [4,4] OpenGL convention}`` — what ``SyntheticRoom`` yields."""
from __future__ import annotations

import glob
import os
import threading
from typing import Dict, List, Optional

import numpy as np
import torch
import yaml

from ..slam.common.camera import Camera


# -- resamplers with OpenCV's conventions --------------------------------------
def resize_bilinear(img: np.ndarray, W: int, H: int) -> np.ndarray:
    """cv2.resize(img, (W, H), INTER_LINEAR): src = (dst + 0.5) * scale - 0.5,
    border replicated, no anti-aliasing"""
    h, w = img.shape[:2]
    if (h, w) == (H, W):
        return img
    x = np.clip((np.arange(W) + 0.5) * (w / W) - 0.5, 0, w - 1)
    y = np.clip((np.arange(H) + 0.5) * (h / H) - 0.5, 0, h - 1)
    return _bilinear(img, x[None, :].repeat(H, 0), y[:, None].repeat(W, 1))


def resize_nearest(img: np.ndarray, W: int, H: int) -> np.ndarray:
    """cv2.resize(..., INTER_NEAREST): src = floor(dst * scale)"""
    h, w = img.shape[:2]
    if (h, w) == (H, W):
        return img
    x = np.minimum((np.arange(W) * (w / W)).astype(np.int64), w - 1)
    y = np.minimum((np.arange(H) * (h / H)).astype(np.int64), h - 1)
    return img[y[:, None], x[None, :]]


def _bilinear(img, x, y):
    h, w = img.shape[:2]
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    fx, fy = (x - x0), (y - y0)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    if img.ndim == 3:
        fx, fy = fx[..., None], fy[..., None]
    top = img[y0c, x0c] * (1 - fx) + img[y0c, x1c] * fx
    bot = img[y1c, x0c] * (1 - fx) + img[y1c, x1c] * fx
    return top * (1 - fy) + bot * fy


def undistort(img: np.ndarray, fx, fy, cx, cy, dist) -> np.ndarray:
    """cv2.undistort(img, K, dist) with the new camera matrix = K: every
    output pixel samples the input at its distorted position (radial k1,k2,k3
    + tangential p1,p2; pixels mapped outside read 0 like BORDER_CONSTANT)"""
    d = list(np.asarray(dist, dtype=np.float64).reshape(-1)) + [0.0] * 5
    k1, k2, p1, p2, k3 = d[:5]
    h, w = img.shape[:2]
    u, v = np.meshgrid(np.arange(w, dtype=np.float64),
                       np.arange(h, dtype=np.float64))
    x, y = (u - cx) / fx, (v - cy) / fy
    r2 = x * x + y * y
    rad = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    us, vs = xd * fx + cx, yd * fy + cy
    out = _bilinear(img.astype(np.float64), us, vs)
    inside = (us >= 0) & (us <= w - 1) & (vs >= 0) & (vs <= h - 1)
    if img.ndim == 3:
        inside = inside[..., None]
    return np.where(inside, out, 0.0)


def _imread_rgb(path) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGB'), dtype=np.uint8)


def _imread_depth(path) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:     # 16-bit PNG ('I;16'): raw sensor units
        return np.asarray(im)


class FileDataset:
    """mirror of ``BaseDataset`` (datasets.py:60-134)"""

    def __init__(self, data_path: str, device='cpu'):
        self.input_folder = data_path
        with open(os.path.join(data_path, 'devices.yaml')) as f:
            cam = yaml.safe_load(f)['cam']
        self.cfg = cam
        self.device = torch.device(device)
        self.png_depth_scale = float(cam['png_depth_scale'])
        self.H, self.W = int(cam['H']), int(cam['W'])
        self.fx, self.fy = float(cam['fx']), float(cam['fy'])
        self.cx, self.cy = float(cam['cx']), float(cam['cy'])
        self.distortion = (np.asarray(cam['distortion'], dtype=np.float64)
                           if 'distortion' in cam else None)
        self.crop_edge = int(cam.get('crop_edge', 0))
        self.downsample_factor = int(cam.get('downsample_factor', 1))
        e, ds = self.crop_edge, self.downsample_factor
        # intrinsics after cropping and down-sampling (datasets.py:84-91)
        self.camera = Camera(fx=self.fx / ds, fy=self.fy / ds,
                             cx=(self.cx - e) / ds, cy=(self.cy - e) / ds,
                             height=int((self.H - 2 * e) / ds),
                             width=int((self.W - 2 * e) / ds))
        self.color_paths: List[str] = []
        self.depth_paths: List[str] = []
        self.poses: List[np.ndarray] = []

    def __len__(self):
        return len(self.color_paths)

    @property
    def n_frames(self):
        return len(self)

    def get_camera(self):
        return self.camera

    def __getitem__(self, k) -> Dict:
        color = _imread_rgb(self.color_paths[k]).astype(np.float64)
        depth = _imread_depth(self.depth_paths[k])
        if self.distortion is not None:
            # colour only, like the reference (datasets.py:104-107)
            color = undistort(color, self.fx, self.fy, self.cx, self.cy,
                              self.distortion)
        color = color / 255.0
        depth = depth.astype(np.float32) / np.float32(self.png_depth_scale)
        H, W = depth.shape
        color = resize_bilinear(color, W, H)
        e = self.crop_edge
        if e > 0:
            color, depth = color[e:-e, e:-e], depth[e:-e, e:-e]
        if self.downsample_factor > 1:
            H = (H - 2 * e) // self.downsample_factor
            W = (W - 2 * e) // self.downsample_factor
            color = resize_bilinear(color, W, H)
            depth = resize_nearest(depth, W, H)
        return {'index': k, 'rgb': np.ascontiguousarray(color, np.float32),
                'depth': np.ascontiguousarray(depth, np.float32),
                'c2w': self.poses[k].copy()}


def _to_opengl(c2w: np.ndarray) -> np.ndarray:
    """camera axes of the datasets (x right, y down, z forward) -> the
    codebase's (x right, y up, z backward): datasets.py:156-164"""
    c2w = np.array(c2w, dtype=np.float64)
    c2w[:3, 1] *= -1
    c2w[:3, 2] *= -1
    return c2w


class Replica(FileDataset):
    """results/frame*.jpg, results/depth*.png, traj.txt (datasets.py:140-167)"""

    def __init__(self, data_path, device='cpu'):
        super().__init__(data_path, device)
        self.color_paths = sorted(glob.glob(f'{data_path}/results/frame*.jpg'))
        self.depth_paths = sorted(glob.glob(f'{data_path}/results/depth*.png'))
        with open(f'{data_path}/traj.txt') as f:
            lines = f.readlines()
        self.poses = [
            # the reference stores float32 poses (torch .float())
            _to_opengl(np.array(list(map(float, lines[i].split())))
                       .reshape(4, 4)).astype(np.float32).astype(np.float64)
            for i in range(len(self.color_paths))]


def associate_frames(t_image, t_depth, t_pose, max_dt=0.08):
    """nearest depth / pose per colour time stamp (datasets.py:476-497)"""
    out = []
    for i, t in enumerate(t_image):
        j = int(np.argmin(np.abs(t_depth - t)))
        if t_pose is None:
            if abs(t_depth[j] - t) < max_dt:
                out.append((i, j))
        else:
            k = int(np.argmin(np.abs(t_pose - t)))
            if abs(t_depth[j] - t) < max_dt and abs(t_pose[k] - t) < max_dt:
                out.append((i, j, k))
    return out


class TUM_RGBD(FileDataset):
    """rgb.txt / depth.txt / groundtruth.txt association, sub-sampled to at
    most ``frame_rate`` frames a second (datasets.py:461-552)"""

    def __init__(self, data_path, device='cpu', frame_rate=32):
        super().__init__(data_path, device)
        from scipy.spatial.transform import Rotation
        pose_list = os.path.join(data_path, 'groundtruth.txt')
        if not os.path.isfile(pose_list):
            pose_list = os.path.join(data_path, 'pose.txt')

        def parse(path, skiprows=0):
            return np.loadtxt(path, delimiter=' ', dtype=np.str_,
                              skiprows=skiprows)
        image_data = parse(os.path.join(data_path, 'rgb.txt'))
        depth_data = parse(os.path.join(data_path, 'depth.txt'))
        pose_data = parse(pose_list, skiprows=1)
        pose_vecs = pose_data[:, 1:].astype(np.float64)
        t_img = image_data[:, 0].astype(np.float64)
        t_dep = depth_data[:, 0].astype(np.float64)
        t_pose = pose_data[:, 0].astype(np.float64)
        assoc = associate_frames(t_img, t_dep, t_pose)
        keep = [0]
        for i in range(1, len(assoc)):
            if t_img[assoc[i][0]] - t_img[assoc[keep[-1]][0]] > \
                    1.0 / frame_rate:
                keep.append(i)
        for ix in keep:
            i, j, k = assoc[ix]
            self.color_paths.append(os.path.join(data_path, image_data[i, 1]))
            self.depth_paths.append(os.path.join(data_path, depth_data[j, 1]))
            pose = np.eye(4)
            pose[:3, :3] = Rotation.from_quat(pose_vecs[k, 3:]).as_matrix()
            pose[:3, 3] = pose_vecs[k, :3]
            self.poses.append(_to_opengl(pose).astype(np.float32)
                              .astype(np.float64))


dataset_dict = {'replica': Replica, 'tumrgbd': TUM_RGBD}


def get_dataset(data_path, data_type, device='cpu'):
    return dataset_dict[data_type](data_path, device=device)


class Prefetcher:
    """Decodes frames ahead on a worker thread and uploads them ONCE: pinned
    staging buffers, ``non_blocking`` copies on a side stream, an event per
    frame that the consumer's stream waits on.  ``loader[k]`` returns the
    dataset item plus ``depth_dev`` [H*W,1] / ``rgb_dev`` [H*W,3] on the
    device; frames are expected in increasing order (SLAM), ``depth`` frames
    are kept in flight."""

    def __init__(self, dataset, device, depth: int = 2):
        self.dataset = dataset
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self.n_frames = len(dataset)
        self._cuda = self.device.type == 'cuda'
        self._stream = torch.cuda.Stream(self.device) if self._cuda else None
        self._ready: Dict[int, Dict] = {}
        self._cv = threading.Condition()
        self._want = 0            # frames < _want may be loaded
        self._stop = False
        self._err: Optional[BaseException] = None
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._next = 0
        self._advance(0)
        self._thread.start()

    def __len__(self):
        return self.n_frames

    def _advance(self, k):
        with self._cv:
            self._want = max(self._want, min(self.n_frames,
                                             k + 1 + self.depth))
            self._cv.notify_all()

    def _work(self):
        try:
            while True:
                with self._cv:
                    while not self._stop and self._next >= self._want:
                        self._cv.wait()
                    if self._stop:
                        return
                    k = self._next
                item = dict(self.dataset[k])
                if self._cuda:
                    d = torch.from_numpy(item['depth']).reshape(-1, 1)
                    c = torch.from_numpy(item['rgb']).reshape(-1, 3)
                    d, c = d.pin_memory(), c.pin_memory()
                    with torch.cuda.stream(self._stream):
                        item['depth_dev'] = d.to(self.device,
                                                 non_blocking=True)
                        item['rgb_dev'] = c.to(self.device, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self._stream)
                    item['_event'], item['_pinned'] = ev, (d, c)
                with self._cv:
                    self._ready[k] = item
                    self._next = k + 1
                    self._cv.notify_all()
        except BaseException as e:  # surfaced to the consumer
            with self._cv:
                self._err = e
                self._cv.notify_all()

    def __getitem__(self, k):
        if not 0 <= k < self.n_frames:
            raise IndexError(k)
        self._advance(k)
        with self._cv:
            while k not in self._ready and self._err is None:
                if k < self._next - 0 and k not in self._ready:
                    # already consumed: decode again synchronously
                    break
                self._cv.wait()
            if self._err is not None:
                raise self._err
            item = self._ready.pop(k, None)
            for old in [j for j in self._ready if j < k]:
                del self._ready[old]
        if item is None:
            item = dict(self.dataset[k])
            if self._cuda:
                item['depth_dev'] = torch.from_numpy(item['depth']).reshape(
                    -1, 1).to(self.device)
                item['rgb_dev'] = torch.from_numpy(item['rgb']).reshape(
                    -1, 3).to(self.device)
            return item
        ev = item.pop('_event', None)
        item.pop('_pinned', None)
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            # allocated on the side stream, consumed on this one: without
            # record_stream the caching allocator hands the block back to the
            # side stream's pool when the Frame frees it, and the next
            # prefetch copy could overwrite memory queued kernels still read
            for k2 in ('depth_dev', 'rgb_dev'):
                item[k2].record_stream(cur)
        return item

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join(timeout=5)
