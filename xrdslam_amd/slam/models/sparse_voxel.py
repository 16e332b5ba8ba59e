"""``SparseVoxel`` — the Vox-Fusion model (reference:
slam/models/sparse_voxel.py:39-357): a sparse voxel octree whose leaf-voxel
vertices index rows of one embedding table; rays are intersected with the
octree, sampled inside the hit voxels, decoded by a small MLP and composited
with SDF bell weights.

Native operators: the octree (``compat.svo.Octree``, host C++, node ids
bit-exact with the reference's pybind class), the two ray/voxel kernels
(``compat.grid``, HIP) and the fused voxel-feature + decoder kernels
(``engine/vox.py``: vertex gather, trilinear feature, the 16-128-128-129 /
144-128-3 decoder and their backward as one launch each way).  Compositing
and losses are torch ops on the padded [rays, samples] arrays."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Type, Union

import torch
from torch.nn import Parameter

from ...compat import svo as _svo
from ..model_components import voxel_helpers_voxfusion as vh
from ..model_components.decoder_voxfusion import Decoder
from ..model_components.utils import compute_loss, get_sdf_loss
from .base_model import Model, ModelConfig


@dataclass
class SparseVoxelConfig(ModelConfig):
    _target: Type = field(default_factory=lambda: SparseVoxel)
    # octree
    voxels_each_dim: int = 256
    voxel_size: float = 0.2
    num_embeddings: int = 20000
    embed_dim: int = 16
    max_distance: int = 10
    max_dpeth: float = 10
    # loss weights
    training_trunc: float = 0.05
    trainging_rgb_weight: float = .5
    trainging_depth_weight: float = 1.0
    trainging_sdf_weight: float = 5000
    trainging_fs_weight: float = 10.0
    # decoder
    depth: int = 2
    width: int = 128
    in_dim: int = 16
    embedder: str = 'none'
    # tracking / mapping
    step_size: float = 0.05
    max_voxel_hit: int = 20
    num_iterations: int = 30
    overlap_th: float = 0.7
    keyframe_th: int = 30
    keyframe_selection: str = 'random'
    data_sc_factor: int = 1


class SparseVoxel(Model):
    config: SparseVoxelConfig

    def __init__(self, config, camera, bounding_box=None, **kwargs) -> None:
        super().__init__(config=config, camera=camera,
                         bounding_box=bounding_box, **kwargs)
        self.map_lock = torch.multiprocessing.RLock()
        # like the reference (:89-92) the step is stored in metres on the
        # config object itself
        self.config.step_size = self.config.voxel_size * self.config.step_size
        self.pose_offset = int(self.config.voxels_each_dim / 2.0 *
                               self.config.voxel_size)
        self.map_states = None
        self.noise_fn = vh._uniform_noise  # replaceable for parity tests
        # static capacities of the fused ray pipeline: sample slots per ray
        # and mean points per ray (grown by check_capacity when a batch does
        # not fit); capacity_version keys captured graphs
        self.meta_sync = None   # replaceable (tests play the ranks in turn)
        self.s_cap = 256
        self.pts_per_ray = 96
        self.capacity_version = 0
        self._workspaces = {}

    def populate_modules(self):
        super().populate_modules()
        cfg = self.config
        # node ids come from a process-global counter (the reference's static
        # Octant::next_index_) and double as row indices of the exported
        # arrays: one live octree per process, like the reference — a new
        # model starts the counter again
        _svo.reset_id_counter()
        self.svo = _svo.Octree()
        self.svo.init(256, cfg.embed_dim, cfg.voxel_size)
        self.embeddings = torch.nn.Parameter(
            torch.zeros(cfg.num_embeddings, cfg.embed_dim))
        torch.nn.init.normal_(self.embeddings, std=0.01)
        self.decoder = Decoder(depth=cfg.depth, width=cfg.width,
                               in_dim=cfg.embed_dim, embedder=cfg.embedder)

    # -- plugin surface -------------------------------------------------------
    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        return {'decoder': list(self.decoder.parameters()),
                'embeddings': [self.embeddings]}

    def get_outputs(self, input) -> Dict[str, Union[torch.Tensor, List]]:
        return self.render_rays(input['rays_o'].unsqueeze(0),
                                input['rays_d'].unsqueeze(0),
                                target_d=input['target_d'].unsqueeze(0))

    def get_loss_dict(self, outputs, inputs, is_mapping,
                      stage=None) -> Dict[str, torch.Tensor]:
        """L1 colour/depth on the rays that hit the octree, SDF + free-space
        terms on their samples (:103-143)"""
        cfg = self.config
        from ...engine import dist as _dist
        if inputs.get('sharded', False) and _dist.state.enabled:
            return self._sharded_losses(outputs, inputs)
        ray_mask = outputs['ray_mask']
        target_d = inputs['target_d'][ray_mask]
        target_rgb = inputs['target_s'][ray_mask]
        td = target_d.squeeze()
        valid = (td > 0.01) * (td < cfg.max_dpeth)
        w = valid.clone().unsqueeze(-1)
        rgb_loss = compute_loss(outputs['rgb'][ray_mask] * w, target_rgb * w,
                                loss_type='l1')
        depth_loss = compute_loss(outputs['depth'][ray_mask].squeeze()[valid],
                                  td[valid], loss_type='l1')
        fs_loss, sdf_loss = get_sdf_loss(
            outputs['z_vals'], target_d, outputs['sdf'],
            cfg.training_trunc * cfg.data_sc_factor, 'l2')
        return {'rgb_loss': rgb_loss * cfg.trainging_rgb_weight,
                'depth_loss': depth_loss * cfg.trainging_depth_weight,
                'sdf_loss': sdf_loss * cfg.trainging_sdf_weight,
                'fs_loss': fs_loss * cfg.trainging_fs_weight}

    def _sharded_losses(self, outputs, inputs):
        """the same four terms for a SHARD of the mapping rays (multi-GPU):
        local sums divided by batch-global normalisers.  The reference's
        SDF / free-space terms are means over the PADDED [rays, samples] array,
        so the global denominator is (hit rays over all ranks) x (longest
        sample row over all ranks); the balancing weights use global counts."""
        import torch.distributed as dist
        cfg = self.config
        ray_mask = outputs['ray_mask']
        target_d = inputs['target_d'][ray_mask]
        target_rgb = inputs['target_s'][ray_mask]
        td = target_d.squeeze(-1)
        valid = (td > 0.01) & (td < cfg.max_dpeth)
        w = valid.unsqueeze(-1).to(target_rgb.dtype)
        rgb_sum = torch.abs(outputs['rgb'][ray_mask] * w - target_rgb * w).sum()
        depth_sum = torch.where(
            valid, torch.abs(outputs['depth'][ray_mask] - td),
            torch.zeros_like(td)).sum()
        z, sdf = outputs['z_vals'], outputs['sdf']
        trunc = cfg.training_trunc * cfg.data_sc_factor
        front = (z < (target_d - trunc)).to(z.dtype)
        back = (z > (target_d + trunc)).to(z.dtype)
        m = (1.0 - front) * (1.0 - back) * (target_d > 0.0).to(z.dtype)
        fs_sum = ((sdf * front - front)**2).sum()
        sdf_sum = (((z + sdf * trunc) * m - target_d * m)**2).sum()
        counts = torch.stack([
            front.sum(), m.sum(), valid.sum().to(z.dtype),
            torch.tensor(float(td.shape[0]), device=z.device, dtype=z.dtype)
        ]).double()
        s_max = torch.tensor(float(z.shape[1]), device=z.device,
                             dtype=torch.float64)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(s_max, op=dist.ReduceOp.MAX)
        n_fs, n_sdf, n_valid, n_hit = [float(c) for c in counts]
        fs_w = 1.0 - n_fs / (n_fs + n_sdf)
        sdf_w = 1.0 - n_sdf / (n_fs + n_sdf)
        denom = n_hit * float(s_max)
        return {
            'rgb_loss': rgb_sum / (3.0 * n_hit) * cfg.trainging_rgb_weight,
            'depth_loss': depth_sum / max(n_valid, 1.0) *
            cfg.trainging_depth_weight,
            'sdf_loss': sdf_sum / denom * sdf_w * cfg.trainging_sdf_weight,
            'fs_loss': fs_sum / denom * fs_w * cfg.trainging_fs_weight}

    # -- rendering --------------------------------------------------------------
    def render_rays(self, rays_o, rays_d, target_d=None, chunk_size=-1):
        """rays [1,N,3] -> dict (None when nothing is hit), :160-275"""
        cfg = self.config
        ms = self.map_states
        intersections, hits = vh.ray_intersect(
            rays_o, rays_d, ms['voxel_center_xyz'], ms['voxel_structure'],
            cfg.voxel_size, cfg.max_voxel_hit, cfg.max_distance)
        if hits.sum() == 0:
            print('\n\n', '!' * 20, 'render_rays. no hit', '!' * 20, '\n\n')
            return None
        ray_mask = hits.view(1, -1)
        intersections = {k: v[ray_mask].reshape(-1, v.size(-1))
                         for k, v in intersections.items()}
        rays_o = rays_o[ray_mask].reshape(-1, 3)
        rays_d = rays_d[ray_mask].reshape(-1, 3)
        samples = vh.ray_sample(intersections, step_size=cfg.step_size,
                                noise_fn=self.noise_fn)
        z_vals = samples['sampled_point_depth']
        sample_mask = samples['sampled_point_voxel_idx'].long().ne(-1)
        if sample_mask.sum() == 0:
            return None
        xyz = vh.ray(rays_o.unsqueeze(1), rays_d.unsqueeze(1),
                     z_vals.unsqueeze(2))
        dirs = rays_d.unsqueeze(1).expand(*z_vals.size(), rays_d.size(-1))
        dirs = dirs / (torch.norm(dirs, 2, -1, keepdim=True) + 1e-8)
        samples['sampled_point_xyz'] = xyz
        samples['sampled_point_ray_direction'] = dirs
        valid = {k: v[sample_mask] for k, v in samples.items()}
        n_pts = valid['sampled_point_depth'].shape[0]
        step = n_pts if chunk_size < 0 else chunk_size
        outs = []
        for i in range(0, n_pts, step):
            chunk = {k: v[i:i + step] for k, v in valid.items()}
            fused = None
            if getattr(self, 'use_fused', True):
                # voxel features + decoder in one kernel each way
                # (engine/vox.py: xrd_vox_points_fwd / _bwd)
                from ...engine import vox as _vox
                fused = _vox.points(self.decoder, chunk['sampled_point_xyz'],
                                    chunk['sampled_point_voxel_idx'], ms,
                                    cfg.voxel_size)
            outs.append(fused if fused is not None else self.decoder(
                vh.get_features(chunk, ms, cfg.voxel_size)))
        field_out = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
        # padded samples read sdf = 1 (free space) and colour 0
        sdf = vh.masked_scatter_ones(sample_mask, field_out['sdf']).squeeze(-1)
        colour = vh.masked_scatter(sample_mask, field_out['color'])
        valid_mask = torch.where(sample_mask, torch.ones_like(sample_mask),
                                 torch.zeros_like(sample_mask))
        weights, z_min = self.sdf2weights(sdf, z_vals, valid_mask)
        rgb = torch.sum(weights[..., None] * colour, dim=-2)
        depth = torch.sum(weights * z_vals, dim=-1)
        ray_mask = ray_mask.squeeze()
        n = ray_mask.shape[0]
        depth = depth.new_zeros(n).masked_scatter(ray_mask, depth)
        rgb = rgb.new_zeros(n, 3).masked_scatter(
            ray_mask.unsqueeze(-1).expand(n, 3), rgb)
        return {'depth': depth, 'rgb': rgb, 'sdf': sdf, 'z_vals': z_vals,
                'ray_mask': ray_mask, 'weights': weights, 'z_min': z_min}

    def sdf2weights(self, sdf, z_vals, valid_mask):
        """sigma(s/tr) sigma(-s/tr), cut ``tr`` behind the first sign change,
        padded samples removed, normalised (:277-304)"""
        tr = self.config.training_trunc
        w = torch.sigmoid(sdf / tr) * torch.sigmoid(-sdf / tr)
        signs = sdf[:, 1:] * sdf[:, :-1]
        crossing = torch.where(signs < 0.0, torch.ones_like(signs),
                               torch.zeros_like(signs))
        inds = torch.argmax(crossing, axis=1)[..., None]
        z_min = torch.gather(z_vals, 1, inds)
        mask = torch.where(z_vals < z_min + self.config.data_sc_factor * tr,
                           torch.ones_like(z_vals), torch.zeros_like(z_vals))
        w = w * mask * valid_mask
        return w / (torch.sum(w, axis=-1, keepdims=True) + 1e-8), z_min

    # -- map maintenance ----------------------------------------------------------
    @staticmethod
    def distinct_voxels_torch(voxels):
        """distinct rows in first-occurrence order with torch ops"""
        uniq, inv = torch.unique(voxels, dim=0, return_inverse=True)
        first = torch.full((uniq.shape[0], ), voxels.shape[0],
                           dtype=torch.int64, device=voxels.device)
        first.scatter_reduce_(0, inv, torch.arange(
            voxels.shape[0], device=voxels.device), reduce='amin')
        return voxels[torch.sort(first).values]

    def insert_points(self, points, dedup=True):
        """world points -> voxel coordinates -> octree (:333-340).  With
        ``dedup`` the distinct voxels are extracted on the device in
        first-occurrence order before the (host-side) insertion: the octree
        only ever creates a node for the first occurrence, so node and vertex
        ids are identical while the device->host copy shrinks from one row per
        depth pixel to one per voxel.  (``distinct_voxels_torch`` is the same
        selection in torch ops: the parity check of the kernels.)"""
        voxels = torch.div(points, self.config.voxel_size,
                           rounding_mode='floor').int()
        if dedup and voxels.is_cuda and voxels.shape[0] > 0:
            # first occurrences through a device hash table + one compaction
            # (xrd_voxel_first_flags / xrd_compact_rows) in place of
            # torch.unique(dim=0)'s lexicographic sort of one row per pixel
            from ...engine.map_ops import distinct_voxels
            voxels = distinct_voxels(voxels)
        self.svo.insert(voxels.cpu().int())
        self.update_map_states()

    @torch.enable_grad()
    def update_map_states(self):
        """flat octree arrays on the device (:342-357).  They live in
        capacity buffers that are updated IN PLACE (the octree only grows), so
        that captured graphs keep valid addresses from frame to frame; the
        buffers are re-allocated (and ``capacity_version`` bumped, which
        retires the graphs) only when the node count outgrows them."""
        dev = self.embeddings.device
        voxels, children, features = self.svo.get_centres_and_children()
        centres = (voxels[:, :3] + voxels[:, -1:] / 2) * self.config.voxel_size
        children = torch.cat([children, voxels[:, -1:]], -1)
        T = voxels.shape[0]
        buf = getattr(self, '_map_buf', None)
        if buf is None or buf['cap'] < T or buf['device'] != dev:
            cap = max(4096, 2 * T)
            buf = {'cap': cap, 'device': dev,
                   'vertex': torch.full((cap, 8), -1, dtype=torch.int32,
                                        device=dev),
                   'centres': torch.zeros(cap, 3, device=dev),
                   'structure': torch.full((cap, 9), -1, dtype=torch.int32,
                                           device=dev)}
            self._map_buf = buf
            self.capacity_version += 1
        # ONE upload of the three host arrays (each pageable host->device copy
        # is a host wait: three a frame): [T, 8 + 3 + 9] int32 with the
        # centres' float bits, split on the device
        packed = torch.cat([features.int(),
                            centres.float().contiguous().view(torch.int32),
                            children.int()], 1).to(dev)
        buf['vertex'][:T].copy_(packed[:, :8])
        buf['centres'][:T].copy_(packed[:, 8:11].contiguous()
                                 .view(torch.float32))
        buf['structure'][:T].copy_(packed[:, 11:20])
        state = {'voxel_vertex_idx': buf['vertex'][:T],
                 'voxel_center_xyz': buf['centres'][:T],
                 'voxel_structure': buf['structure'][:T],
                 'voxel_vertex_emb': self.embeddings}
        with self.map_lock:
            self.map_states = state

    # -- fused iteration (engine/vox.py, csrc/vox_rays.hip) ---------------------
    def ray_workspace(self, n_rays, need_w):
        """static buffers of one ray-batch shape at the current capacities"""
        from ...engine import vox as _vox
        key = (n_rays, self.s_cap, self.pts_per_ray, bool(need_w))
        ws = self._workspaces.get(key)
        if ws is None:
            ws = _vox.RayWorkspace(n_rays, self.s_cap,
                                   n_rays * self.pts_per_ray, need_w,
                                   self.embeddings.device)
            self._workspaces[key] = ws
        self._last_ws = ws
        return ws

    def draw_noise(self, ws):
        """the sampler's uniform draws, one row per ray (None = fixed 0.5)"""
        if self.noise_fn is None:
            return None
        if self.noise_fn is vh._uniform_noise:
            return torch.empty_like(ws.s_depth).uniform_()
        return self.noise_fn((ws.n, ws.s_cap), ws.s_depth)

    def fused_loss(self, inputs, is_mapping):
        """get_outputs + get_loss_dict of one ray batch as a fixed sequence of
        launches without a host sync -> (summed loss, [rgb, depth, sdf, fs,
        sum]) or None when the kernels do not cover the configuration"""
        from ...engine import vox as _vox
        if not inputs['rays_o'].is_cuda or \
                _vox.decoder_params(self.decoder) is None:
            return None
        need_w = is_mapping and torch.is_grad_enabled()
        ws = self.ray_workspace(inputs['rays_o'].shape[0], need_w)
        # multi-GPU mapping.  Deterministic sharding: the WHOLE batch arrives
        # with a mask of this rank's rays (ray_keep) — every rank samples all
        # rays, so the regrouping and the loss normalisers are the batch's own
        # and the ranks' losses / gradients add up exactly; only the points of
        # the kept rays are evaluated.  Independent draws per rank: a shard of
        # rays, the normalisers are exchanged (meta_sync).
        keep = inputs.get('ray_keep')
        ws.keep = None if keep is None else \
            keep.to(torch.uint8).contiguous()
        ws.meta_sync = (self.meta_sync or _vox.allreduce_meta) \
            if inputs.get('sharded', False) and keep is None else None
        out = _vox.render_loss(
            self.decoder, ws, self.map_states, self.config, inputs['rays_o'],
            inputs['rays_d'], inputs['target_d'], inputs['target_s'],
            self.draw_noise(ws), map_grads=need_w)
        return out[0], out[1]

    def check_capacity(self):
        """one device->host read of the last batch's size record; grows the
        static capacities when a batch did not fit (its result was computed
        on the truncated samples) and retires graphs captured on the old
        shapes.  Returns the record."""
        ws = getattr(self, '_last_ws', None)
        if ws is None:
            return None
        bits, meta = ws.overflow()
        return self._apply_capacity_record(ws, bits, meta)

    def check_capacity_deferred(self):
        """check_capacity without draining the queue: the size record of THIS
        call is copied to pinned host memory behind the call's launches, and
        the record of the PREVIOUS call (long complete) is evaluated now.  An
        overflow is therefore reported — and the capacities grown — one
        optimisation call late; the iterations in between run on truncated
        rows exactly like the ones the immediate check reports after the fact.
        Returns the previous call's record (None for the first call)."""
        out = None
        pend = self.__dict__.pop('_cap_pending', None)
        if pend is not None:
            ws, host, ev = pend
            ev.synchronize()
            m = host.tolist()
            bits = m[5] | (8 if m[10] else 0) | m[11]
            m[2] = max(m[2], m[12])
            m[14] = max(m[14], m[13])
            out = self._apply_capacity_record(ws, bits, m)
        ws = getattr(self, '_last_ws', None)
        if ws is not None:
            bufs = self.__dict__.setdefault('_cap_host', [])
            if len(bufs) < 2:
                bufs.append(torch.empty(ws.meta.shape, dtype=ws.meta.dtype,
                                        pin_memory=True))
            host = bufs[0]
            bufs.reverse()           # alternate: the other one may be pending
            host.copy_(ws.meta, non_blocking=True)
            ws.meta[11:14].zero_()   # the sticky slots restart behind the copy
            ev = torch.cuda.Event()
            ev.record()
            self._cap_pending = (ws, host, ev)
        return out

    def _apply_capacity_record(self, ws, bits, meta):
        if bits & 4:
            raise RuntimeError('vox sampler produced a sample row with a '
                               'hole: the fused compaction assumes prefixes')
        if bits & 8:
            raise RuntimeError('octree traversal stack overflow')
        grown = False
        if bits & 1:
            self.s_cap = int(-(-meta[2] * 5 // 4 // 64) * 64)
            grown = True
        if bits & 2:
            per_ray = -(-int(meta[14]) // ws.n)
            self.pts_per_ray = int(-(-per_ray * 5 // 4 // 16) * 16)
            grown = True
        if grown:
            import warnings
            warnings.warn(
                'Vox-Fusion fused ray pipeline: a batch since the last check '
                f'did not fit the static capacities (bits {bits}); those '
                'iterations ran on truncated sample rows.  Capacities grown '
                f'to s_cap={self.s_cap}, pts_per_ray={self.pts_per_ray}.')
            self.capacity_version += 1
            self._workspaces.clear()
            self._last_ws = None
        return {'n_hit_rays': meta[1], 'max_hits': meta[0],
                'row_len': meta[2], 's_max': meta[3], 'n_pts': meta[4],
                'grown': grown, 's_cap': self.s_cap,
                'pts_per_ray': self.pts_per_ray}

    def get_map_states(self):
        with self.map_lock:
            return self.map_states.copy()
