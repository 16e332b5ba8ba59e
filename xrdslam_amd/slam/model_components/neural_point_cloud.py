"""``NeuralPointCloud`` — Point-SLAM's map (reference:
slam/model_components/neural_point_cloud.py): neural points are added along
the sensor rays (``N_add`` per pixel around the measured depth) wherever no
existing point lies within the add radius; every point carries a geometry and
a colour feature vector (the optimised parameters).

MI355X-side: positions live in ONE device tensor (the reference keeps a Python
list and rebuilds a tensor from it in every decoder call,
decoder_pointslam.py:165,420) and the neighbour index is the exact uniform-grid
kNN (``xrd_knn_*``, engine/knn.py) instead of FAISS IVF(400 lists, 4 probes):
exact search returns the true 8 nearest neighbours, which the approximate index
only does with high probability."""
import threading

import numpy as np
import torch
import torch.nn as nn


def _default_knn(device):
    from ...engine.knn import GridKNN
    return GridKNN(0.16, device)


def _feature_init(n, c_dim):
    """N(0, 0.1) on the CPU generator like the reference (:184-189)"""
    return torch.zeros([n, c_dim]).normal_(mean=0, std=0.1)


class NeuralPointCloud(nn.Module):
    def __init__(self, c_dim, nn_num, radius_add, cuda_id, radius_min,
                 radius_query, fix_interval_when_add_along_ray,
                 use_dynamic_radius, N_surface, N_add, near_end_surface,
                 far_end_surface, device, knn_factory=_default_knn):
        super().__init__()
        self.device = device
        self.c_dim, self.nn_num = c_dim, nn_num
        self.use_dynamic_radius = use_dynamic_radius
        self.radius_add, self.radius_min = radius_add, radius_min
        self.radius_query = radius_query
        self.fix_interval_when_add_along_ray = fix_interval_when_add_along_ray
        self.N_surface, self.N_add = N_surface, N_add
        self.near_end_surface = near_end_surface
        self.far_end_surface = far_end_surface
        self._cloud = torch.zeros(0, 3, device=device)   # all neural points
        self._input_pos = torch.zeros(0, 3, device=device)
        self._input_rgb = torch.zeros(0, 3, device=device)
        self._pts_num = 0
        self.col_feats = None
        self.geo_feats = None
        self.frustum_mask = None
        self.cuda_id = cuda_id
        self.lock = threading.Lock()
        self.index = knn_factory(device)
        self.feature_init_fn = _feature_init  # replaceable for parity tests
        self.device_insert = True   # selection + placement as kernels (GPU)

    # -- accessors (the reference returns Python lists) ---------------------------
    def cloud_tensor(self, device=None):
        return self._cloud if device is None else self._cloud.to(device)

    def cloud_pos(self, index=None):
        c = self._cloud.cpu().tolist()
        return c if index is None else c[index]

    def input_pos(self):
        return self._input_pos.cpu().tolist()

    def input_rgb(self):
        return self._input_rgb.cpu().tolist()

    def pts_num(self):
        return self._pts_num

    def index_ntotal(self):
        return self.index.ntotal

    def get_radius_query(self):
        return self.radius_query

    def set_mask(self, new_mask):
        self.frustum_mask.data.copy_(new_mask.unsqueeze(1))

    def get_geo_feats(self):
        return self.geo_feats * self.frustum_mask

    def get_col_feats(self):
        return self.col_feats * self.frustum_mask

    def update_geo_feats(self, feats, indices=None):
        if indices is not None:
            self.geo_feats[indices] = feats.detach().clone()
        else:
            assert feats.shape[0] == self.geo_feats.shape[0]
            self.geo_feats = feats.detach().clone()

    def update_col_feats(self, feats, indices=None):
        if indices is not None:
            self.col_feats[indices] = feats.detach().clone()
        else:
            assert feats.shape[0] == self.col_feats.shape[0]
            self.col_feats = feats.detach().clone()

    # -- growth ---------------------------------------------------------------------
    def add_neural_points(self, batch_rays_o, batch_rays_d, batch_gt_depth,
                          batch_gt_color, train=False, is_pts_grad=False,
                          dynamic_radius=None):
        """-> number of sensor points that received neural points (:109-221)"""
        if not batch_rays_o.shape[0]:
            return 0
        dev = self.device
        if torch.device(dev).type == 'cuda' and self.device_insert:
            new_pos, new_rgb, pts, n_kept = self._select_and_place_kernels(
                batch_rays_o, batch_rays_d, batch_gt_depth, batch_gt_color,
                is_pts_grad, dynamic_radius)
        else:
            new_pos, new_rgb, pts, n_kept = self._select_and_place_torch(
                batch_rays_o, batch_rays_d, batch_gt_depth, batch_gt_color,
                is_pts_grad, dynamic_radius)
        self._input_pos = torch.cat([self._input_pos, new_pos], 0)
        self._input_rgb = torch.cat([self._input_rgb, new_rgb], 0)
        self._cloud = torch.cat([self._cloud, pts.detach().float()], 0)
        self._pts_num += pts.shape[0]
        n = pts.shape[0]
        geo = self.feature_init_fn(n, self.c_dim).to(dev)
        col = self.feature_init_fn(n, self.c_dim).to(dev)
        if self.geo_feats is None:
            self.geo_feats = nn.Parameter(geo.clone().detach()
                                          .requires_grad_(True))
            self.col_feats = nn.Parameter(col.clone().detach()
                                          .requires_grad_(True))
        else:
            self.geo_feats = nn.Parameter(
                torch.cat([self.geo_feats.detach(), geo], 0)
                .requires_grad_(True))
            self.col_feats = nn.Parameter(
                torch.cat([self.col_feats.detach(), col], 0)
                .requires_grad_(True))
        self.frustum_mask = nn.Parameter(
            torch.ones(self._pts_num, 1, dtype=torch.bool, device=dev),
            requires_grad=False)
        with self.lock:
            self.index.add(pts.detach())
        return n_kept

    def _linspace_table(self):
        """the N_add offsets (fixed interval) or interpolation weights along
        the ray, from torch.linspace on the device like the reference"""
        key = (self.fix_interval_when_add_along_ray, self.N_add)
        tab = getattr(self, '_lin_tab', None)
        if tab is None or tab[0] != key:
            lo, hi = (-0.04, 0.04) if self.fix_interval_when_add_along_ray \
                else (0.0, 1.0)
            tab = (key, torch.linspace(lo, hi, steps=self.N_add,
                                       device=self.device))
            self._lin_tab = tab
        return tab[1]

    def _select_and_place_kernels(self, rays_o, rays_d, gt_depth, gt_color,
                                  is_pts_grad, dynamic_radius):
        """selection (depth > 0, no neural point within the add radius) and
        placement as two launches around the neighbour count
        (xrd_point_sensor_points, xrd_knn_search_count, xrd_point_insert); one
        size read-back.  Rays without depth take part in the search and are
        dropped by the insertion kernel, which keeps ray order: the same rows
        as compacting by ``valid`` first."""
        from ...engine import map_ops
        dev = self.device
        o, d = rays_o.to(dev), rays_d.to(dev)
        depth, color = gt_depth.to(dev), gt_color.to(dev)
        pts_gt = map_ops.point_sensor_points(o, d, depth)
        n_nb = None
        if self.index.ntotal > 0:
            _, _, n_nb = self.find_neighbors_faiss(
                pts_gt, step='add', is_pts_grad=is_pts_grad,
                dynamic_radius=dynamic_radius)
        return map_ops.point_insert(
            o, d, depth, color, pts_gt, n_nb, self._linspace_table(),
            self.fix_interval_when_add_along_ray, self.near_end_surface,
            self.far_end_surface)

    def _select_and_place_torch(self, batch_rays_o, batch_rays_d,
                                batch_gt_depth, batch_gt_color, is_pts_grad,
                                dynamic_radius):
        """the same in torch ops, statement by statement like the reference
        (:113-176)"""
        dev = self.device
        valid = batch_gt_depth > 0
        color = (batch_gt_color * 255)[valid]
        rays_o, rays_d = batch_rays_o[valid].to(dev), batch_rays_d[valid].to(dev)
        depth = batch_gt_depth[valid].to(dev)
        pts_gt = (rays_o[..., None, :] + rays_d[..., None, :] *
                  depth[..., None, None]).reshape(-1, 3)
        keep = torch.ones(pts_gt.shape[0], device=dev).bool()
        if self.index.ntotal > 0:
            _, _, n_nb = self.find_neighbors_faiss(
                pts_gt, step='add', is_pts_grad=is_pts_grad,
                dynamic_radius=dynamic_radius)
            keep = n_nb == 0
        d = depth.unsqueeze(-1).repeat(1, self.N_add)
        if self.fix_interval_when_add_along_ray:
            z_vals = d + torch.linspace(-0.04, 0.04, steps=self.N_add,
                                        device=dev).unsqueeze(0)
        else:
            t = torch.linspace(0.0, 1.0, steps=self.N_add, device=dev)
            z_vals = self.near_end_surface * d * (1. - t) + \
                self.far_end_surface * d * t
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        pts = pts[keep].reshape(-1, 3)
        return pts_gt[keep], color[keep].to(dev), pts.detach().float(), \
            int(torch.sum(keep))

    # -- queries ------------------------------------------------------------------
    def find_neighbors_faiss(self, pos, step='add', retrain=False,
                             is_pts_grad=False, dynamic_radius=None):
        """-> (squared distances [n,8] ascending, ids [n,8] (-1 = none),
        number of neighbours inside the radius [n]) (:223-282)"""
        assert step in ('add', 'query')
        if step == 'query':
            radius = self.radius_query
        else:
            radius = self.radius_min if is_pts_grad else self.radius_add
        counted = getattr(self.index, 'search_count', None)
        if counted is not None and pos.is_cuda:
            # the HIP grid kNN counts the neighbours inside the (per-query)
            # radius in the search launch
            if dynamic_radius is not None:
                assert pos.shape[0] == dynamic_radius.shape[0]
            with self.lock:
                D, ids, n_nb = counted(
                    pos.float().detach(), self.nn_num,
                    dynamic_radius if dynamic_radius is not None else radius)
            return D, ids, n_nb
        with self.lock:
            D, ids = self.index.search(pos.float().detach(), self.nn_num)
        D, ids = D.to(self.device), ids.to(self.device)
        if dynamic_radius is not None:
            dynamic_radius = dynamic_radius.to(self.device)
            assert pos.shape[0] == dynamic_radius.shape[0]
            n_nb = (D < dynamic_radius.reshape(-1, 1)**2).sum(-1).int()
        else:
            n_nb = (D < radius**2).sum(-1).int()
        return D, ids, n_nb

    def sample_near_pcl(self, rays_o, rays_d, near, far, num):
        """z samples for rays without sensor depth: march 25 points from
        ``near`` to ``far`` and keep the span between the first and second
        marched point that have neighbours (:284-350)"""
        rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        n_rays, intervals = rays_d.shape[0], 25
        z = torch.linspace(near, far, steps=intervals, device=self.device)
        pts = (rays_o[..., None, :] + rays_d[..., None, :] *
               z[..., :, None]).reshape(-1, 3)
        if torch.is_tensor(far):
            far = far.item()
        section = np.linspace(near, far, intervals)
        total = np.tile(np.linspace(near, far, num), (n_rays, 1))
        _, _, n_nb = self.find_neighbors_faiss(pts, step='query')
        has = n_nb.cpu().numpy().reshape(n_rays, -1).astype(bool)
        invalid = has.sum(-1) < 2
        if invalid.sum() < n_rays:
            r, c = np.where(has[~invalid])
            idx = np.concatenate(([0], np.flatnonzero(r[1:] != r[:-1]) + 1,
                                  [r.size]))
            spans = [c[idx[i]:idx[i + 1]] for i in range(len(idx) - 1)]
            total[~invalid] = np.asarray(
                [np.linspace(section[s[0]], section[s[1]], num=num)
                 for s in spans])
        return torch.from_numpy(total).float().to(self.device), \
            torch.from_numpy(invalid).to(self.device)
