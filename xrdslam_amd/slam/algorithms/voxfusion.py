"""``VoxFusion`` algorithm plugin (reference: slam/algorithms/voxfusion.py):
every frame is a map frame; mapping first allocates voxels for the new depth
image (back-projected with the current pose), then optimises embeddings,
decoder and the poses of a random keyframe window + the current frame."""
from __future__ import annotations

import functools
from dataclasses import dataclass, field
from typing import Type

import numpy as np
import torch

from ...engine import dist as _dist
from ..common.common import get_rays, get_samples
from ..models.sparse_voxel import SparseVoxelConfig
from .base_algorithm import Algorithm, AlgorithmConfig


@dataclass
class VoxFusionConfig(AlgorithmConfig):
    _target: Type = field(default_factory=lambda: VoxFusion)
    model: SparseVoxelConfig = field(default_factory=SparseVoxelConfig)
    mapping_sample: int = 2048
    min_sample_pixels: int = 100
    tracking_sample: int = 1024
    ray_batch_size: int = 3000


class VoxFusion(Algorithm):
    config: VoxFusionConfig

    def __init__(self, config: VoxFusionConfig, camera, device: str) -> None:
        super().__init__(config, camera, device)
        self.model = config.model.setup(camera=camera, bounding_box=None)
        self.model.to(device)
        self.bundle_adjust = True
        self._rays_cam = None
        # one launch sequence per iteration without host syncs
        # (SparseVoxel.fused_loss); off = the plugin hooks get_outputs /
        # get_loss_dict on the modular operators
        self.fused_iteration = True

    def _camera_rays(self, device):
        """[H*W,3] camera-frame directions, OpenGL (precompute(), :37-52)"""
        if self._rays_cam is None or self._rays_cam.device != \
                torch.device(device):
            cam = self.camera
            ix, iy = torch.meshgrid(
                torch.arange(cam.width, device=device),
                torch.arange(cam.height, device=device), indexing='xy')
            self._rays_cam = torch.stack(
                [(ix - cam.cx) / cam.fx, -(iy - cam.cy) / cam.fy,
                 -torch.ones_like(ix)], -1).float().reshape(-1, 3)
        return self._rays_cam

    # the iteration's shapes depend on the window only through its length:
    # mapping graphs are kept from call to call (static frame slots, map
    # arrays updated in place, capacities keyed by capacity_version)
    persistent_map_graph = True

    def map_slot_key(self, n_iters, optimize_frames, coarse):
        if not self.fused_iteration or _dist.state.enabled:
            return None     # sharded mapping: collectives inside the iteration
        f = optimize_frames[-1]
        return (len(optimize_frames), n_iters, f.h, f.w, f.separate_LR,
                f.rot_rep, self.config.mapping_sample,
                self.model.capacity_version)

    def track_slot_key(self):
        return (self.model.capacity_version, self.fused_iteration)

    def _graphs_ok(self, optimizers, is_mapping):
        # the modular path syncs the host (hit counts, ragged lengths);
        # sharded (multi-GPU) mapping exchanges its batch-global loss
        # normalisers in the middle of the iteration: eager
        if not self.fused_iteration:
            return False
        if is_mapping and _dist.state.enabled and \
                not _dist.state.deterministic:
            # independent draws per rank: the loss normalisers are exchanged
            # in the middle of the iteration (deterministic sharding has no
            # collective before the gradient all-reduce: [gradient graph]
            # all-reduce [step graph], base class)
            return False
        return super()._graphs_ok(optimizers, is_mapping)

    def optimize_update(self, n_iters, optimize_frames, is_mapping,
                        coarse=False):
        out = super().optimize_update(n_iters, optimize_frames, is_mapping,
                                      coarse=coarse)
        if self.fused_iteration:
            # one read of the last batch's size record per call: grows the
            # static capacities (and retires the graphs) if it did not fit
            if getattr(self, 'device_track_result', False):
                # the frame loop keeps its pose chain on the device
                # (SequentialSLAM device_poses): do not drain the queue here
                # either — the record is read one call late
                rec = self.model.check_capacity_deferred()
                if rec is not None:
                    self.last_batch_sizes = rec
            else:
                self.last_batch_sizes = self.model.check_capacity()
        return out

    # -- hooks ---------------------------------------------------------------------
    def get_model_input(self, optimize_frames, is_mapping):
        cfg, dev = self.config, self.model.device
        n = cfg.mapping_sample if is_mapping else cfg.tracking_sample
        # multi-GPU mapping: every rank draws 1/world of the rays from its own
        # RNG stream; losses use batch-global normalisers, gradients are
        # summed in Optimizers.optimizer_step_all
        sharded = is_mapping and _dist.state.enabled
        det = sharded and _dist.state.deterministic
        gen = _dist.state.shard_generator if sharded and not det else None
        if sharded and not det:
            n = _dist.state.shard_count(n)
        if torch.device(dev).type == 'cuda' and self.fused_iteration:
            return self._model_input_kernels(optimize_frames, n, gen, sharded,
                                             track=not is_mapping)
        ro, rd, gd, gc = [], [], [], []
        for f in optimize_frames:
            o, d, dep, col = get_samples(self.camera, n, f.get_pose(), f.depth,
                                         f.rgb, device=dev, frame=f,
                                         generator=gen)
            if det:   # same draws on every rank, this rank's slice
                lo, hi = _dist.state.shard_slice(n)
                o, d, dep, col = o[lo:hi], d[lo:hi], dep[lo:hi], col[lo:hi]
            ro.append(o.float())
            rd.append(d.float())
            gd.append(dep.float())
            gc.append(col.float())
        return {'rays_o': torch.cat(ro), 'rays_d': torch.cat(rd),
                'target_s': torch.cat(gc), 'target_d': torch.cat(gd),
                'sharded': sharded}

    def _model_input_kernels(self, frames, n, gen, sharded, track=False):
        """get_samples of every window frame (common.py:188-227: pixels drawn
        with replacement over the whole image, OpenGL rays through the frame's
        pose) as one index draw + one launch per frame, differentiable w.r.t.
        the poses (engine/slam_ops.SampleRaysFn)"""
        from ...engine.slam_ops import SampleRaysFn
        cam, dev = self.camera, self.model.device
        idx = torch.randint(cam.height * cam.width, (len(frames), n),
                            device=dev, generator=gen)
        poses = [f.get_pose().to(dev) for f in frames]
        # one frame (tracking, the first mapping calls): a view, no copy
        c2ws = poses[0].unsqueeze(0) if len(poses) == 1 else torch.stack(poses)
        if track:
            # the iteration's best-pose bookkeeping reads this pose: one
            # quaternion -> matrix launch an iteration instead of two
            self._iter_c2w = poses[-1].detach()
        imgs = [f.device_images(dev) for f in frames]
        big = 1e30
        ro, rd, td, tc, _, _ = SampleRaysFn.apply(
            c2ws, idx, [i[0] for i in imgs], [i[1] for i in imgs], cam,
            (0, 0, cam.width), (-big, big, -big, big, -big, big), False)
        out = {'rays_o': ro, 'rays_d': rd, 'target_s': tc, 'target_d': td,
               'sharded': sharded}
        if sharded and _dist.state.deterministic:
            # every rank drew the SAME batch (shared RNG stream) and passes
            # all of it: the ray pipeline (intersection, the sampler's
            # [200, R, P] regrouping of the hit rays, the size record) runs on
            # the whole batch everywhere, this rank evaluates the points of a
            # contiguous slice of every frame's rays (ray_keep)
            lo, hi = _dist.state.shard_slice(n)
            keep = torch.zeros(len(frames), n, dtype=torch.uint8, device=dev)
            keep[:, lo:hi] = 1
            out['ray_keep'] = keep.reshape(-1)
        return out

    def create_voxels(self, frame):
        """allocate the voxels seen by this frame (:96-107); the back
        projection runs on the device-resident depth image"""
        dev = self.model.device
        depth, _ = frame.device_images(dev)
        pts = self._camera_rays(dev) * depth
        pts = pts[depth.reshape(-1) > 0]
        pose = frame.get_pose().detach().to(dev)
        pts = pts @ pose[:3, :3].transpose(-1, -2) + pose[:3, 3]
        self.model.insert_points(pts)

    def pre_precessing(self, cur_frame, is_mapping):
        if is_mapping:
            self.create_voxels(cur_frame)

    def post_processing(self, step, is_mapping, optimizer=None, coarse=False):
        pass

    def optimizer_config_update(self, max_iters, coarse=False):
        pass

    def get_loss(self, optimize_frames, is_mapping, step=None, n_iters=None,
                 coarse=False):
        inp = self.get_model_input(optimize_frames, is_mapping)
        if self.fused_iteration:
            # (a shard of the mapping rays included: the size record's loss
            # normalisers are made batch-global inside, engine/vox.py)
            fused = self.model.fused_loss(inp, is_mapping)
            if fused is not None:
                self.last_loss_terms = fused[1]
                return fused[0]
        keep = inp.pop('ray_keep', None)
        if keep is not None:
            # the modular path renders a shard of the rays (normalisers
            # exchanged in the loss): this rank's slice of the batch
            sel = keep.bool()
            inp = {k: (v[sel] if torch.is_tensor(v) and v.dim() > 0 and
                       v.shape[0] == sel.shape[0] else v)
                   for k, v in inp.items()}
        out = self.model(inp)
        losses = self.model.get_loss_dict(out, inp, is_mapping, step)
        return functools.reduce(torch.add, losses.values())

    def render_img(self, c2w, gt_depth=None, idx=None):
        with torch.no_grad():
            dev = self.model.device
            rays_o, rays_d = get_rays(self.camera, c2w, device=dev)
            rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
            if gt_depth is not None:
                gt_depth = torch.as_tensor(gt_depth).to(dev).reshape(-1, 1)
            H, W = self.camera.height, self.camera.width
            depths, colors = [], []
            bs = self.config.ray_batch_size
            from ...engine import vox as _vox
            fused = self.fused_iteration and rays_o.is_cuda and \
                _vox.decoder_params(self.model.decoder) is not None
            for i in range(0, rays_d.shape[0], bs):
                td = None if gt_depth is None else gt_depth[i:i + bs]
                n = rays_d[i:i + bs].shape[0]
                if fused:
                    # the same chunks through the fused ray pipeline (hits,
                    # samples, decoder, compositing: 9 launches a chunk); a
                    # chunk that outgrows the static capacities is redone
                    m = self.model
                    while True:
                        ws = m.ray_workspace(n, False)
                        d, c = _vox.render(
                            m.decoder, ws, m.map_states, m.config,
                            rays_o[i:i + bs], rays_d[i:i + bs],
                            m.draw_noise(ws))
                        if not m.check_capacity()['grown']:
                            break
                    depths.append(d.double())
                    colors.append(c.clone())
                    continue
                out = self.model.render_rays(
                    rays_o[None, i:i + bs], rays_d[None, i:i + bs],
                    target_d=td)
                if out is None:  # no voxel hit in this chunk
                    depths.append(torch.zeros(n, dtype=torch.float64,
                                              device=dev))
                    colors.append(torch.zeros(n, 3, device=dev))
                    continue
                depths.append(out['depth'].double())
                colors.append(out['rgb'])
            return torch.cat(colors).reshape(H, W, 3).cpu().numpy(), \
                torch.cat(depths).reshape(H, W).cpu().numpy()

    def update_mesh(self):
        pass

    def get_cloud(self, c2w_np, gt_depth_np):
        return None

    def get_mesh(self):
        """per-voxel 8^3 lattices of the decoder's sdf, zero level set,
        vertex colours from the colour head (voxfusion.py:173-278)"""
        with self.lock, torch.no_grad():
            return self.extract_mesh(clean_mesh=False, require_color=True,
                                     res=8)

    @torch.no_grad()
    def extract_mesh(self, res=8, clean_mesh=False, require_color=False):
        from ...engine import vox as _vox
        from ..common.mesher import Mesh, marching_tetrahedra
        m = self.model
        ms, vs = m.map_states, m.config.voxel_size
        dev = ms['voxel_center_xyz'].device
        # leaf voxels: all eight vertices present (voxfusion.py:181-184)
        ids = torch.nonzero(~ms['voxel_vertex_idx'].eq(-1).any(-1)).flatten()
        n = int(ids.numel())
        if n == 0:
            return None
        lin = torch.linspace(-0.5, 0.5, res, device=dev)
        xx, yy, zz = torch.meshgrid(lin, lin, lin, indexing='ij')
        offs = torch.stack([xx, yy, zz], -1).reshape(1, -1, 3) * vs
        centres = ms['voxel_center_xyz'][ids]
        xyz = (centres[:, None, :] + offs).reshape(-1, 3).contiguous()
        vox = ids[:, None].expand(n, res**3).reshape(-1).int().contiguous()
        out = _vox.points(m.decoder, xyz, vox, ms, vs)
        if out is None:
            raise NotImplementedError('mesh extraction needs the fused '
                                      'decoder kernels (CUDA, default decoder)')
        sdf = out['sdf'].reshape(n, res, res, res)
        # every voxel is its own little volume (the reference runs marching
        # cubes voxel by voxel, :254-277): stacked along x with a separator
        # plane of inf, which the extraction skips
        vol = torch.full((n, res + 1, res, res), float('inf'), device=dev)
        vol[:, :res] = sdf
        h = 1.0 / (res - 1)
        verts, faces = marching_tetrahedra(vol.reshape(-1, res, res), 0.0,
                                           (h, h, h))
        if verts.shape[0] == 0:
            return None
        v = torch.from_numpy(verts).to(dev)
        gx = v[:, 0] / h
        # lattice x of voxel k spans [k (res+1), k (res+1) + res - 1]; the
        # half-cell shift keeps rounding at the ends on the right voxel
        k = torch.div(gx + 0.5, res + 1, rounding_mode='floor').long() \
            .clamp(0, n - 1)
        local = torch.stack([(gx - k * (res + 1)) * h, v[:, 1], v[:, 2]], -1)
        world = ((local - 0.5) * vs + centres[k].double()).float()
        colors = None
        if require_color:
            col = _vox.points(m.decoder, world.contiguous(),
                              ids[k].int().contiguous(), ms, vs)['color']
            colors = (col.clamp(0, 1) * 255).byte().cpu().numpy()
        return Mesh(world.cpu().numpy().astype(np.float64), faces, colors)
