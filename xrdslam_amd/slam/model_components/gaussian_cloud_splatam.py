"""``GaussianCloud`` — SplaTAM's map (reference:
slam/model_components/gaussian_cloud_splatam.py): isotropic 3-D Gaussians
(means, colours, unnormalised rotations, logit opacities, log scales) seeded
from back-projected depth pixels, rendered twice per iteration by the tile
rasteriser (colour pass; depth / silhouette / depth^2 pass), grown where the
silhouette is empty or the depth disagrees, pruned by opacity / size.

The rasteriser is ``xrdslam_amd.compat.diff_gaussian_rasterization`` (HIP,
``xrd_gs_*``).  Parameters live on the device the first frame arrives on."""
import torch
import torch.nn as nn

from ...compat import diff_gaussian_rasterization as _dgr
from ..common.common import setup_camera
from .slam_helpers_splatam import (accumulate_mean2d_gradient, build_rotation,
                                   transform_to_frame,
                                   transformed_params2depthplussilhouette,
                                   transformed_params2rendervar)


def _renderer(settings):
    # resolved at call time so that tests can stand in an oracle rasteriser
    return _dgr.GaussianRasterizer(raster_settings=settings)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianCloud(nn.Module):
    def __init__(self, init_rgb, init_depth, w2c, camera, prune_dict,
                 densify_dict):
        super().__init__()
        self.camera = camera
        self.device = init_depth.device
        self.gaussian_cam = setup_camera(camera, w2c.detach().cpu().numpy(),
                                         device=self.device)
        self.first_frame_w2c = w2c.detach()
        self.prune_dict, self.densify_dict = prune_dict, densify_dict
        self.fused_passes = True   # one raster pass for rgb + depth/silhouette
        mask = (init_depth > 0).reshape(-1)
        pts, sq = self.get_pointcloud(init_rgb, init_depth, w2c, mask=mask)
        self.params, self.variables = self.initialize_params(pts, sq)
        self.variables['scene_radius'] = init_depth.max() / 3.0

    # -- rendering ---------------------------------------------------------------
    def render(self, w2c, gaussians_grad=False, camera_grad=False,
               retain_grad=True, c2w=None):
        """``w2c`` [4,4], or ``c2w`` (then the rigid inverse is taken inside
        the preparation kernel instead of a torch.inverse per iteration)"""
        if self.fused_passes and self.params['means3D'].is_cuda:
            return self._render_fused(w2c, c2w, gaussians_grad, camera_grad)
        if w2c is None:
            w2c = torch.inverse(c2w)
        pts = transform_to_frame(self.params['means3D'], w2c,
                                 gaussians_grad=gaussians_grad,
                                 camera_grad=camera_grad)
        rv = transformed_params2rendervar(self.params, pts)
        ds_rv = transformed_params2depthplussilhouette(
            self.params, self.first_frame_w2c, pts)
        if retain_grad:
            rv['means2D'].retain_grad()
        im, radius, depth = _renderer(self.gaussian_cam)(**rv)
        depth_sil, _, _ = _renderer(self.gaussian_cam)(**ds_rv)
        if retain_grad:
            self.variables['means2D'] = rv['means2D']
        seen = radius > 0
        self.variables['max_2D_radius'][seen] = torch.max(
            radius[seen], self.variables['max_2D_radius'][seen])
        self.variables['seen'] = seen
        return {'rgb': im, 'depth_sil': depth_sil, 'depth': depth}

    def _render_fused(self, w2c, c2w, gaussians_grad, camera_grad):
        """MI355X path: one preparation launch (csrc/gs_prepare.hip) + one
        raster pass with two colour sets (compat.rasterize_dual): both
        renders share every Gaussian's geometry and opacity.  means2D
        carries no gradient here — it only feeds the 3D-GS densification
        statistics, which switch this path off (fused_passes = not
        use_gaussian_splatting_densification)."""
        from ...engine.gs import GsPrepareFn
        p = self.params
        pose, is_c2w = (c2w, True) if c2w is not None else (w2c, False)
        pts, rot, opac, scales, dscol = GsPrepareFn.apply(
            p['means3D'], p['unnorm_rotations'], p['logit_opacities'],
            p['log_scales'], pose, self.first_frame_w2c, is_c2w,
            gaussians_grad, camera_grad)
        colors = p['rgb_colors'] if gaussians_grad else \
            p['rgb_colors'].detach()
        im, radius, depth, depth_sil = _dgr.rasterize_dual(
            self.gaussian_cam, pts, None, opac, colors, dscol, scales, rot)
        # max over the seen Gaussians == max over all (unseen: radius 0)
        mr = self.variables['max_2D_radius']
        torch.maximum(mr, radius, out=mr)
        self.variables['radius'] = radius
        self.variables.pop('seen', None)
        return {'rgb': im, 'depth_sil': depth_sil, 'depth': depth}

    def pair_count(self, c2w):
        """(Gaussian, tile) pairs a render from ``c2w`` would bin (0-d device
        tensor; preparation + preprocess launches only)"""
        from ...engine.gs import GsPrepareFn
        p = self.params
        with torch.no_grad():
            pts, rot, opac, scales, _ = GsPrepareFn.apply(
                p['means3D'], p['unnorm_rotations'], p['logit_opacities'],
                p['log_scales'], c2w, self.first_frame_w2c, True, False, False)
            return _dgr.count_pairs(self.gaussian_cam, pts, scales, rot, opac)

    # -- optimiser surgery -------------------------------------------------------
    def _reset_stats(self, n):
        for k in ('means2D_gradient_accum', 'denom', 'max_2D_radius'):
            self.variables[k] = torch.zeros(n, device=self.device).float()

    def remove_points(self, to_remove, optimizer):
        """drop rows from every non-pose parameter and the per-Gaussian
        statistics (:84-111).  Reference behaviour kept on purpose: the sliced
        Adam moments are re-keyed to the OLD parameter object before the new
        one is swapped in, so a pruned parameter continues with a fresh Adam
        state."""
        # ONE compaction for the five parameters, their Adam moments and the
        # statistics (boolean-mask indexing would compact — and read a size
        # back to the host — once per tensor: 19 times).  On the GPU:
        # xrd_compact_rows, three launches and one size read-back for all.
        keep = ~to_remove.reshape(-1)
        slots = []          # (setter, tensor)
        for name, opt in optimizer.items():
            if 'pose' in name:
                continue
            group = opt.param_groups[0]
            old = group['params'][0]
            state = opt.state.get(old, None)
            if state is not None:
                slots.append(((state, 'exp_avg'), state['exp_avg']))
                slots.append(((state, 'exp_avg_sq'), state['exp_avg_sq']))
            slots.append(((opt, name), old.detach()))
        for k in ('means2D_gradient_accum', 'denom', 'max_2D_radius',
                  'timestep'):
            if k in self.variables:
                slots.append(((self.variables, k), self.variables[k]))
        tensors = [t for _, t in slots]
        if keep.is_cuda and all(t.element_size() == 4 for t in tensors):
            from ...engine.map_ops import compact_rows
            rows, _ = compact_rows(keep, tensors)
        else:
            idx = torch.nonzero(keep).reshape(-1)
            rows = [t.index_select(0, idx) for t in tensors]
        for ((owner, key), _), new in zip(slots, rows):
            if isinstance(owner, dict):
                owner[key] = new
                continue
            # the parameter itself: moments re-keyed to the OLD object first
            group = owner.param_groups[0]
            old = group['params'][0]
            state = owner.state.get(old, None)
            if state is not None:
                del owner.state[old]
                owner.state[old] = state
            group['params'][0] = nn.Parameter(new.requires_grad_(True))
            self.params[key] = group['params'][0]

    def update_params_and_optimizer(self, new_params, optimizer):
        """replace whole parameters, moments reset to zero (:113-123)"""
        for k, v in new_params.items():
            opt = optimizer[k]
            group = opt.param_groups[0]
            state = opt.state.get(group['params'][0], None)
            state['exp_avg'] = torch.zeros_like(v)
            state['exp_avg_sq'] = torch.zeros_like(v)
            del opt.state[group['params'][0]]
            group['params'][0] = nn.Parameter(v.requires_grad_(True))
            opt.state[group['params'][0]] = state
            self.params[k] = group['params'][0]

    def cat_params_to_optimizer(self, new_params, optimizer):
        """append rows, zero moments for them (:156-178)"""
        for k, v in new_params.items():
            opt = optimizer[k]
            group = opt.param_groups[0]
            old = group['params'][0]
            state = opt.state.get(old, None)
            grown = nn.Parameter(torch.cat((old, v), 0).requires_grad_(True))
            if state is not None:
                state['exp_avg'] = torch.cat(
                    (state['exp_avg'], torch.zeros_like(v)), 0)
                state['exp_avg_sq'] = torch.cat(
                    (state['exp_avg_sq'], torch.zeros_like(v)), 0)
                del opt.state[old]
                opt.state[grown] = state
            group['params'][0] = grown
            self.params[k] = grown

    def _opacity_size_mask(self, d, it):
        thr = d['final_removal_opacity_threshold'] \
            if it == d['stop_after'] else d['removal_opacity_threshold']
        rm = (torch.sigmoid(self.params['logit_opacities']) < thr).squeeze()
        if it >= d['remove_big_after']:
            big = torch.exp(self.params['log_scales']).max(dim=1).values > \
                0.1 * self.variables['scene_radius']
            rm = torch.logical_or(rm, big)
        return rm

    def prune_gaussians(self, it, optimizer):
        d = self.prune_dict
        if it > d['stop_after']:
            return
        if it >= d['start_after'] and it % d['prune_every'] == 0:
            self.remove_points(self._opacity_size_mask(d, it), optimizer)
        if it > 0 and it % d['reset_opacities_every'] == 0 and \
                d['reset_opacities']:
            self.update_params_and_optimizer({'logit_opacities': inverse_sigmoid(
                torch.ones_like(self.params['logit_opacities']) * 0.01)},
                optimizer)

    def densify(self, it, optimizer):
        """3D-GS clone/split densification (:180-267); off in the reference's
        default SplaTAM configuration"""
        d = self.densify_dict
        if it > d['stop_after']:
            return
        self.variables = accumulate_mean2d_gradient(self.variables)
        thr = d['grad_thresh']
        if it >= d['start_after'] and it % d['densify_every'] == 0:
            grads = self.variables['means2D_gradient_accum'] / \
                self.variables['denom']
            grads[grads.isnan()] = 0.0
            size = torch.exp(self.params['log_scales']).max(dim=1).values
            small = size <= 0.01 * self.variables['scene_radius']
            clone = torch.logical_and(grads >= thr, small)
            self.cat_params_to_optimizer(
                {k: v[clone] for k, v in self.params.items()
                 if 'pose' not in k}, optimizer)
            n_pts = self.params['means3D'].shape[0]
            padded = torch.zeros(n_pts, device=self.device)
            padded[:grads.shape[0]] = grads
            size = torch.exp(self.params['log_scales']).max(dim=1).values
            split = torch.logical_and(
                padded >= thr, size > 0.01 * self.variables['scene_radius'])
            n = d['num_to_split_into']
            new = {k: v[split].repeat(n, 1) for k, v in self.params.items()
                   if 'pose' not in k}
            stds = torch.exp(self.params['log_scales'])[split].repeat(n, 3)
            samples = torch.normal(mean=torch.zeros_like(stds), std=stds)
            rots = build_rotation(
                self.params['unnorm_rotations'][split]).repeat(n, 1, 1)
            new['means3D'] += torch.bmm(rots, samples.unsqueeze(-1)) \
                .squeeze(-1)
            new['log_scales'] = torch.log(torch.exp(new['log_scales']) /
                                          (0.8 * n))
            self.cat_params_to_optimizer(new, optimizer)
            self._reset_stats(self.params['means3D'].shape[0])
            self.remove_points(torch.cat((split, torch.zeros(
                n * int(split.sum()), dtype=torch.bool, device=self.device))),
                optimizer)
            self.remove_points(self._opacity_size_mask(d, it), optimizer)
        if it > 0 and it % d['reset_opacities_every'] == 0 and \
                d.get('reset_opacities', False):
            self.update_params_and_optimizer({'logit_opacities': inverse_sigmoid(
                torch.ones_like(self.params['logit_opacities']) * 0.01)},
                optimizer)

    # -- growth -------------------------------------------------------------------
    def add_new_gaussians(self, gt_rgb, gt_depth, curr_w2c, sil_thres,
                          mean_sq_dist_method):
        """new Gaussians where the current map is absent (silhouette below
        ``sil_thres``) or hides something nearer than it renders (:269-316)"""
        pts = transform_to_frame(self.params['means3D'], curr_w2c,
                                 gaussians_grad=False, camera_grad=False)
        ds_rv = transformed_params2depthplussilhouette(
            self.params, self.first_frame_w2c, pts)
        depth_sil, _, _ = _renderer(self.gaussian_cam)(**ds_rv)
        absent = depth_sil[1] < sil_thres
        render_depth = depth_sil[0]
        err = torch.abs(gt_depth - render_depth) * (gt_depth > 0)
        in_front = (render_depth > gt_depth) * (err > 50 * err.median())
        mask = (absent | in_front).reshape(-1)
        if torch.sum(mask) == 0:
            return
        mask = mask & (gt_depth > 0).reshape(-1)
        new_pts, sq = self.get_pointcloud(gt_rgb, gt_depth, curr_w2c.detach(),
                                          mask=mask,
                                          mean_sq_dist_method=mean_sq_dist_method)
        new_params, _ = self.initialize_params(new_pts, sq)
        for k, v in new_params.items():
            self.params[k] = nn.Parameter(
                torch.cat((self.params[k], v), 0).requires_grad_(True))
        self._reset_stats(self.params['means3D'].shape[0])

    def initialize_params(self, pt_cld, mean3_sq_dist):
        n = pt_cld.shape[0]
        dev = self.device
        raw = {
            'means3D': pt_cld[:, :3],
            'rgb_colors': pt_cld[:, 3:6],
            'unnorm_rotations': torch.tensor(
                [1., 0., 0., 0.], device=dev).repeat(n, 1),
            'logit_opacities': torch.zeros((n, 1), dtype=torch.float,
                                           device=dev),
            'log_scales': torch.log(torch.sqrt(mean3_sq_dist))[..., None],
        }
        params = {k: nn.Parameter(v.to(dev).float().contiguous()
                                  .requires_grad_(True))
                  for k, v in raw.items()}
        variables = {k: torch.zeros(n, device=dev).float()
                     for k in ('max_2D_radius', 'means2D_gradient_accum',
                               'denom')}
        return params, variables

    def get_pointcloud(self, color, depth, w2c, mask=None,
                       compute_mean_sq_dist=True,
                       mean_sq_dist_method='projective'):
        """[x,y,z,r,g,b] per pixel (camera looking down +z) in the world of
        ``w2c`` + squared initial radius depth / mean focal (:355-399)"""
        cam, dev = self.camera, depth.device
        xg, yg = torch.meshgrid(
            torch.arange(cam.width, device=dev).float(),
            torch.arange(cam.height, device=dev).float(), indexing='xy')
        xx = ((xg - cam.cx) / cam.fx).reshape(-1)
        yy = ((yg - cam.cy) / cam.fy).reshape(-1)
        z = depth.reshape(-1)
        pts4 = torch.stack((xx * z, yy * z, z, torch.ones_like(z)), -1)
        pts = (torch.inverse(w2c) @ pts4.T).T[:, :3]
        sq = None
        if compute_mean_sq_dist:
            if mean_sq_dist_method != 'projective':
                raise ValueError(
                    f'Unknown mean_sq_dist_method {mean_sq_dist_method}')
            sq = (z / ((cam.fx + cam.fy) / 2))**2
        cloud = torch.cat((pts, color.reshape(-1, 3)), -1)
        if mask is not None:
            if cloud.is_cuda and sq is not None:
                from ...engine.map_ops import compact_rows
                (cloud, sq), _ = compact_rows(mask.reshape(-1),
                                              [cloud, sq.contiguous()])
            else:
                idx = torch.nonzero(mask.reshape(-1)).reshape(-1)
                cloud = cloud.index_select(0, idx)
                if sq is not None:
                    sq = sq.index_select(0, idx)
        return cloud, sq
