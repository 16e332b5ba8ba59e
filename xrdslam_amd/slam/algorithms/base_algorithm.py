"""``Algorithm`` — the optimiser loop shared by tracking and mapping, with the
reference's hooks and semantics (slam/algorithms/base_algorithm.py:44-302):

* plugin hooks: get_model_input / get_loss / pre_precessing / post_processing /
  render_img / get_mesh / get_cloud / optimizer_config_update;
* ``optimize_update``: pre_precessing(last frame) -> fresh ``Optimizers`` ->
  n_iters x (zero_grad, loss, backward, post_processing, step, scheduler);
  tracking returns the pose that had the LOWEST loss evaluated BEFORE its Adam
  step (:262-265);
* ``setup_optimizers``: tracking optimises the (first) frame's pose group(s);
  mapping the model groups, plus the poses of all window frames but the oldest
  when bundle adjustment is on (:160-209);
* pose/keyframe bookkeeping under one re-entrant lock (:51,106-158).

MI355X-side differences: the best-loss pose is tracked on the device and read
back once per call, instead of one ``loss.cpu().item()`` host sync per tracking
iteration — same result, no per-iteration stall; and with
``use_graphs`` the iterations of one stage segment are captured once into a
hipGraph and replayed (launch-bound inner loop, ~100 small launches per
iteration otherwise).
"""
from __future__ import annotations

import random
from abc import abstractmethod
from dataclasses import dataclass, field
from typing import Any, Dict, Type

import torch

from ...engine import dist as _dist
from ..common.common import keyframe_selection_overlap
from ..configs.base_config import InstantiateConfig
from ..engine.optimizers import OptimizerConfig, Optimizers
from ..models.base_model import ModelConfig


class _capture(torch.cuda.graph):
    """hipGraph capture context of the frame loops: ``torch.cuda.graph`` with
    three changes (the third: no cyclic garbage collection inside a capture,
    see __enter__).

    * capture mode 'thread_local': the frame prefetcher
      (data/datasets.Prefetcher) allocates pinned memory and issues copies on
      its own stream while the tracker / mapper captures — legal, but in the
      default 'global' mode any such call from ANY thread ends the capture
      with hipErrorStreamCaptureInvalidated (seen in the --ingest files leg of
      bench.py).
    * no ``torch.cuda.empty_cache()`` in front of the capture.  The stock
      context returns every cached block to the driver "to free memory for the
      graph"; SplaTAM and Point-SLAM capture three times per FRAME, and every
      tensor of the next phase then went through hipMalloc again (measured:
      the 1600-point keyframe selection right after the tracking capture
      took 10-27 ms, of which the arithmetic is < 1 ms).  The capture's own
      allocations come from the graph's private pool either way."""

    # the attributes of torch.cuda.graph this override touches (checked once
    # per process; a torch that renamed them gets the stock context, which is
    # slower here but correct)
    _PRIVATE = ('stream_ctx', 'pool', 'cuda_graph', 'capture_error_mode')
    _checked = None

    def __init__(self, graph):
        super().__init__(graph, capture_error_mode='thread_local')
        if _capture._checked is None:
            _capture._checked = all(hasattr(self, a) for a in self._PRIVATE)
            if not _capture._checked:
                import warnings
                warnings.warn('torch.cuda.graph internals changed: graph '
                              'captures use the stock context (empties the '
                              'allocator cache before every capture)')
            else:
                import gc
                gc.collect()    # once, like the stock context does per entry

    def __enter__(self):
        # no cyclic-GC pass while the stream captures: a collection that frees
        # a graph / event of an earlier frame inside the capture aborts the
        # process (the stock context runs gc.collect() in front of every
        # capture instead — tens of ms with the frame objects alive here)
        import gc
        self._gc_was = gc.isenabled()
        gc.disable()
        try:
            if not _capture._checked:
                return super().__enter__()
            torch.cuda.synchronize()
            self.stream_ctx.__enter__()
            self.cuda_graph.capture_begin(
                *self.pool, capture_error_mode=self.capture_error_mode)
        except BaseException:
            # __exit__ is not called when __enter__ raises: the collector
            # must not stay off for the rest of the process
            if self._gc_was:
                gc.enable()
            raise

    def __exit__(self, *exc):
        try:
            return super().__exit__(*exc)
        finally:
            if self._gc_was:
                import gc
                gc.enable()


@dataclass
class AlgorithmConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: Algorithm)
    model: ModelConfig = field(default_factory=ModelConfig)
    keyframe_selection_method: str = 'overlap'
    keyframe_use_ray_sample: bool = True
    tracking_n_iters: int = 10
    mapping_n_iters: int = 60
    mapping_first_n_iters: int = 200
    coarse: bool = False
    mapping_window_size: int = 5
    separate_LR: bool = False
    rot_rep: str = 'quat'
    retain_graph: bool = False
    optimizers: Dict[str, Any] = field(default_factory=lambda: {
        'model': {'optimizer': OptimizerConfig(lr=1e-2)},
        'tracking_pose': {'optimizer': OptimizerConfig(lr=1e-2)},
        'mapping_pose': {'optimizer': OptimizerConfig(lr=1e-3)},
    })


class Algorithm:
    def __init__(self, config: AlgorithmConfig, camera, device: str) -> None:
        self.config, self.camera = config, camera
        self.initialized = False
        self.finished = False
        self.lock = torch.multiprocessing.RLock()
        self.gt_c2w_list = []
        self.gt_c2w_list_ori = []
        self.estimate_c2w_list = []
        self.keyframe_graph = []
        self.bundle_adjust = False

    # ---- hooks ----------------------------------------------------------
    @abstractmethod
    def get_model_input(self, optimize_frames, is_mapping):
        pass

    @abstractmethod
    def get_loss(self, optimize_frames, is_mapping, step=None, n_iters=None,
                 coarse=False):
        pass

    @abstractmethod
    def pre_precessing(self, cur_frame, is_mapping):
        pass

    @abstractmethod
    def post_processing(self, step, is_mapping, optimizer=None, coarse=False):
        pass

    @abstractmethod
    def render_img(self, c2w, gt_depth=None, idx=None):
        return None, None

    @abstractmethod
    def update_mesh(self):
        pass

    @abstractmethod
    def get_mesh(self):
        return None

    @abstractmethod
    def get_cloud(self, c2w_np, gt_depth_np):
        return None

    @abstractmethod
    def optimizer_config_update(self, max_iters, coarse=False):
        pass

    @property
    def device(self):
        return self.model.device

    # ---- shared state (manager-process RPC surface) -----------------------
    def add_framepose(self, c2w, gt_c2w, gt_c2w_ori):
        with self.lock:
            self.estimate_c2w_list.append(c2w)
            self.gt_c2w_list.append(gt_c2w)
            self.gt_c2w_list_ori.append(gt_c2w_ori)

    def update_framepose(self, idx, c2w):
        with self.lock:
            self.estimate_c2w_list[idx] = c2w

    def get_estimate_c2w_list(self):
        with self.lock:
            return self.estimate_c2w_list

    def get_gt_c2w_list(self):
        with self.lock:
            return self.gt_c2w_list

    def get_gt_c2w_list_ori(self):
        with self.lock:
            return self.gt_c2w_list_ori

    def get_keyframes(self):
        with self.lock:
            return self.keyframe_graph

    def add_keyframe(self, keyframe):
        with self.lock:
            self.keyframe_graph.append(keyframe)

    def is_separate_LR(self):
        with self.lock:
            return self.config.separate_LR

    def get_rot_rep(self):
        with self.lock:
            return self.config.rot_rep

    def is_initialized(self):
        with self.lock:
            return self.initialized

    def set_initialized(self):
        with self.lock:
            self.initialized = True

    def is_finished(self):
        with self.lock:
            return self.finished

    def set_finished(self):
        with self.lock:
            self.finished = True

    # ---- optimiser construction -------------------------------------------
    def _pose_groups(self, frames, prefix):
        if self.config.separate_LR:
            groups = {f'{prefix}_r': [], f'{prefix}_t': []}
            for f in frames:
                r, t = f.get_params()
                groups[f'{prefix}_r'].append(r)
                groups[f'{prefix}_t'].append(t)
            return groups
        groups = {prefix: []}
        for f in frames:
            groups[prefix].extend(f.get_params())
        return groups

    def setup_optimizers(self, n_iters, optimize_frames, is_mapping=True,
                         coarse=False) -> Optimizers:
        self.optimizer_config_update(n_iters, coarse)
        cfg = dict(self.config.optimizers)
        if not is_mapping:
            # the reference returns inside its loop: only the first frame's
            # pose is optimised (base_algorithm.py:176,181)
            return Optimizers(cfg, self._pose_groups(optimize_frames[:1],
                                                     'tracking_pose'))
        model_groups = self.model.get_param_groups()
        if not self.bundle_adjust or len(optimize_frames) == 1:
            return Optimizers(cfg, {**model_groups})
        oldest = min(f.fid for f in optimize_frames)
        free = [f for f in optimize_frames if f.fid != oldest]
        return Optimizers(cfg, {**self._pose_groups(free, 'mapping_pose'),
                                **model_groups})

    # ---- tracking / mapping -------------------------------------------------
    def do_tracking(self, cur_frame):
        if self.is_initialized():
            return self.optimize_update(self.config.tracking_n_iters,
                                        [cur_frame], is_mapping=False)

    def do_mapping(self, cur_frame):
        n_iters = self.config.mapping_n_iters if self.is_initialized() \
            else self.config.mapping_first_n_iters
        with torch.no_grad():
            frames = self.select_optimize_frames(
                cur_frame, self.config.keyframe_selection_method)
        self.optimize_update(n_iters, frames, is_mapping=True, coarse=False)
        if not self.is_initialized():
            self.set_initialized()

    # ---- the optimiser loop ------------------------------------------------
    use_graphs = False  # capture each stage segment into a hipGraph (MI355X)
    persistent_track_graph = True  # one tracking graph reused across frames
    # the persistent tracking graph hands its best pose back as a device
    # tensor instead of numpy (set by a driver that keeps the pose chain on
    # the device, slam/pipeline.py: SequentialSLAM(device_poses=True))
    device_track_result = False
    # mapping graphs kept from one mapping call to the next (algorithms whose
    # mapping work has call-independent shapes opt in, see _map_slot_run)
    persistent_map_graph = False
    # ... also when the mapping rays are sharded over ranks (two graphs around
    # the eager gradient all-reduce, job list refreshed per call)
    persistent_map_graph_sharded = True

    def map_slot_key(self, n_iters, optimize_frames, coarse):
        """everything a captured mapping iteration depends on besides the
        data that _map_slot_run copies into the slot; None = do not persist"""
        return None

    def after_mapping_update(self):
        """end of a mapping optimize_update (replayed graphs included)"""

    def track_slot_key(self):
        """whatever else the captured tracking iteration depends on (static
        capacities, kernel-path switches)"""
        return None

    def graph_segment_key(self, is_mapping, step, n_iters, coarse=False):
        """iterations with equal keys run identical device work (same stage,
        same learning rates) and may share one captured graph"""
        return 0

    def host_pre_iteration(self, optimize_frames, is_mapping, step):
        """host-side work of an iteration that must not be frozen into a
        captured graph (e.g. SplaTAM's random choice of the frame an
        iteration renders): called before EVERY iteration — eager, captured
        or replayed — outside the capture"""

    def _graphs_ok(self, optimizers, is_mapping):
        if not self.use_graphs or not torch.cuda.is_available():
            return False
        if torch.device(self.device).type != 'cuda':
            return False
        # sharded mapping issues a collective per iteration: it is kept OUT of
        # the graphs (two graphs around an eager all-reduce, optimize_update)
        for name in optimizers.optimizers:
            if self.config.optimizers[name]['optimizer'].accum_step is not None \
                    or self.config.optimizers[name]['optimizer'].max_norm \
                    is not None:
                return False
        for params in optimizers.parameters.values():
            if any(not p.is_cuda for p in params):
                return False
        return not (self.config.retain_graph and is_mapping)

    def _iteration(self, optimizers, optimize_frames, is_mapping, step,
                   n_iters, coarse, track, part=None):
        """one optimisation iteration.  ``part``: None = all of it; 'grad' =
        up to and including backward / post_processing; 'step' = the optimiser
        steps without the gradient exchange (multi-GPU split-graph mode)"""
        if part == 'step':
            optimizers.optimizer_step_all(step=step, exchange=False)
            return
        optimizers.zero_grad_all()
        # the fused steps' hand-over fields live for ONE iteration: a
        # get_loss call outside this method (tests, evaluation, an exception)
        # must not make the next iteration skip its backward
        self._grads_assigned = False
        self.__dict__.pop('_iter_c2w', None)
        loss = self.get_loss(optimize_frames, is_mapping, step, n_iters,
                             coarse=coarse)
        if not is_mapping:
            # keep the pose that produced the lowest loss (evaluated before
            # its Adam step, base_algorithm.py:262-265), on the device
            cur = self.__dict__.pop('_iter_c2w', None)
            if cur is None:
                cur = optimize_frames[-1].get_pose().detach()
            if cur.is_cuda and loss.is_cuda and \
                    track['loss'].device == loss.device:
                from ...engine import slam_ops
                slam_ops.track_best(loss.detach().double(), cur, track)
            else:
                lval = loss.detach().to(cur.device, torch.float64)
                better = lval < track['loss']
                track['c2w'].copy_(torch.where(better, cur, track['c2w']))
                track['loss'].copy_(torch.where(better, lval, track['loss']))
                track['valid'].logical_or_(better)
        if getattr(self, '_grads_assigned', False):
            # a fused iteration computed the loss AND assigned every .grad
            # (NiceSLAM._fused_map_step / _fused_track_step say so
            # explicitly): nothing to back-propagate
            self._grads_assigned = False
        else:
            # a loss that arrives detached by accident raises here, as in the
            # reference loop
            loss.backward(retain_graph=(self.config.retain_graph and
                                        is_mapping))
        self.post_processing(step, is_mapping, optimizers.optimizers,
                             coarse=coarse)
        if part == 'grad':
            return
        optimizers.optimizer_step_all(step=step)

    # ---- persistent tracking graph ---------------------------------------
    def _track_slot_run(self, n_iters, frame):
        """Tracking through ONE hipGraph kept across frames.  The graph is
        captured over an internal slot frame with static device buffers; each
        new frame is copied into the slot (images, initial pose), the Adam
        state is re-zeroed (the reference builds a fresh Adam per frame,
        base_algorithm.py:160-181) and the graph is replayed n_iters times.
        The caller replaces the frame's pose by the returned best pose
        (tracker.py:107-112), so optimising a copy is equivalent."""
        from ..common.frame import Frame
        dev = self.device
        slot = getattr(self, '_track_slot', None)
        # the captured launches bake in the learning rates and the sampling
        # configuration: they are part of the key
        lrs = tuple(sorted(
            (k, float(v['optimizer'].lr)) for k, v in
            self.config.optimizers.items() if k.startswith('tracking_pose')))
        shape_key = (frame.h, frame.w, frame.separate_LR, frame.rot_rep,
                     n_iters, lrs, getattr(self.config, 'tracking_sample',
                                           None), self.track_slot_key())
        if slot is not None and slot['key'] != shape_key:
            slot = None
        init = frame.get_pose().detach()
        if slot is None:
            sf = Frame(-1, frame.rgb, frame.depth,
                       init_pose=init.cpu().numpy(), gt_pose=None,
                       separate_LR=frame.separate_LR, rot_rep=frame.rot_rep,
                       device=str(dev))
            sf.device_images(dev)  # static image buffers
            slot = {'key': shape_key, 'frame': sf, 'graph': None,
                    'opt': None, 'track': None}
            self._track_slot = slot
        sf = slot['frame']
        d_dev, c_dev = sf.device_images(dev)
        # through the frame's own device cache: uploaded once per frame (or
        # already resident), reused when the frame becomes a mapping frame
        f_d, f_c = frame.device_images(dev)
        d_dev.copy_(f_d, non_blocking=True)
        c_dev.copy_(f_c, non_blocking=True)
        with torch.no_grad():
            for ps, pf in zip(sf.get_params(), frame.get_params()):
                ps.copy_(pf.detach().to(ps.device))
        self.pre_precessing(sf, False)
        if slot['opt'] is None:
            slot['opt'] = self.setup_optimizers(n_iters, [sf],
                                                is_mapping=False)
            slot['opt'].allreduce = False
            slot['track'] = {
                'loss': torch.full((), 10000000000., dtype=torch.float64,
                                   device=dev),
                'c2w': torch.zeros(4, 4, device=dev),
                'valid': torch.zeros((), dtype=torch.bool, device=dev)}
        else:
            self.optimizer_config_update(n_iters, False)
            for opt in slot['opt'].optimizers.values():
                for st in opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
            slot['track']['loss'].fill_(10000000000.)
            slot['track']['c2w'].zero_()
            slot['track']['valid'].fill_(False)
        opt, track = slot['opt'], slot['track']
        self.fixed_shape_batches = True
        for step in range(n_iters):
            if slot['graph'] is None and step == 0:
                self._iteration(opt, [sf], False, step, n_iters, False, track)
            elif slot['graph'] is None:
                g = torch.cuda.CUDAGraph()
                with _capture(g):
                    self._iteration(opt, [sf], False, step, n_iters, False,
                                    track)
                slot['graph'] = g
                g.replay()
            else:
                slot['graph'].replay()
        self.fixed_shape_batches = False
        if self.device_track_result:
            # no host sync: the caller keeps the pose on the device (a frame
            # whose loss never improved keeps its start, like `return None`)
            return torch.where(track['valid'], track['c2w'], init)
        if not bool(track['valid'].item()):
            return None
        return track['c2w'].cpu().numpy()

    # ---- persistent mapping graphs -----------------------------------------
    def _map_slot_run(self, key, n_iters, frames, coarse):
        """One mapping call through hipGraphs that outlive it.  A slot (per
        ``map_slot_key``) owns stand-in frames with static image buffers and
        pose parameters, ONE Optimizers object and one captured graph per stage
        segment.  A call copies the window's images and poses into the slot,
        lets the model refresh its cell selection in place, zeroes the Adam
        state (the reference builds fresh optimisers per call,
        base_algorithm.py:160-181) and replays; bundle-adjusted poses are
        copied back.  The first call of a slot runs the ordinary segment loop
        (first iteration eager, second captured) and keeps the graphs.
        Returns False when the slot cannot be used (caller falls back)."""
        slots = self.__dict__.setdefault('_map_slots', {})
        slot = slots.get(key)
        if slot is not None and slot.get('unusable'):
            return False
        static_before = getattr(self.model, 'static_selection', None)
        if static_before is not None:
            self.model.static_selection = True
        try:
            return self._map_slot_body(slots, slot, key, n_iters, frames,
                                       coarse)
        finally:
            if static_before is not None:
                self.model.static_selection = static_before

    def _map_slot_body(self, slots, slot, key, n_iters, frames, coarse):
        from ..common.frame import Frame
        from ..engine.optimizers import reset_optimizer_state
        dev = self.device
        self.optimizer_config_update(n_iters, coarse)
        if self.bundle_adjust and len(frames) > 1:
            # bundle adjustment keeps the oldest frame fixed
            # (setup_optimizers): it always takes slot 0
            j = min(range(len(frames)), key=lambda i: frames[i].fid)
            if j != len(frames) - 1:
                frames = [frames[j]] + frames[:j] + frames[j + 1:]
        if slot is None:
            sfs = []
            for i, f in enumerate(frames):
                # a frame that kept no images (Co-SLAM keyframes: pose and
                # bank rays only) gets a pose-only stand-in
                has_img = f.depth is not None or f._dev_cache is not None
                shape_src = f if has_img else frames[-1]
                sf = Frame(i - len(frames), shape_src.rgb, shape_src.depth,
                           init_pose=f.get_pose().detach().cpu().numpy(),
                           gt_pose=None, separate_LR=f.separate_LR,
                           rot_rep=f.rot_rep, device=str(dev))
                sf._slot_images = has_img
                if has_img:
                    fd, fc = f.device_images(dev)
                    sf._dev_cache = (torch.empty_like(fd),
                                     torch.empty_like(fc))
                sfs.append(sf)
            slot = slots[key] = {'frames': sfs, 'opt': None, 'graphs': {}}
        slot['calls'] = slot.get('calls', 0) + 1
        self._last_map_slot_key = key
        sfs = slot['frames']
        with torch.no_grad():
            for sf, f in zip(sfs, frames):
                if getattr(sf, '_slot_images', True):
                    fd, fc = f.device_images(dev)
                    sd, sc = sf._dev_cache
                    sd.copy_(fd, non_blocking=True)
                    sc.copy_(fc, non_blocking=True)
                for ps, pf in zip(sf.get_params(), f.get_params()):
                    ps.copy_(pf.detach().to(ps.device))
        self.pre_precessing(sfs[-1], True)
        first = slot['opt'] is None
        if first:
            slot['opt'] = self.setup_optimizers(n_iters, sfs, True,
                                                coarse=coarse)
            slot['opt'].allreduce = bool(_dist.state.enabled)
            if not self._graphs_ok(slot['opt'], True):
                slot['unusable'] = True
                return False
        else:
            self.refresh_map_selection()
            self.reset_slot_optimizers(slot['opt'])
        opt, graphs = slot['opt'], slot['graphs']
        self.fixed_shape_batches = True
        split = _dist.state.enabled
        if first and split:
            # sharded mapping: [gradient graph] eager all-reduce [step graph]
            seg_key, seg_iter = None, 0
            for step in range(n_iters):
                k = self.graph_segment_key(True, step, n_iters, coarse)
                if k != seg_key:
                    seg_key, seg_iter = k, 0
                if seg_iter == 0:
                    self._iteration(opt, sfs, True, step, n_iters, coarse,
                                    None)
                elif k not in graphs:
                    ga = torch.cuda.CUDAGraph()
                    gen = _dist.state.shard_generator
                    if gen is not None and hasattr(
                            ga, 'register_generator_state'):
                        ga.register_generator_state(gen)
                    with _capture(ga):
                        self._iteration(opt, sfs, True, step, n_iters, coarse,
                                        None, part='grad')
                    ga.replay()
                    jobs = _dist.collect_grad_jobs(
                        opt.stepping_parameters(step))
                    _dist.run_grad_jobs(jobs)
                    gb = torch.cuda.CUDAGraph()
                    with _capture(gb):
                        self._iteration(opt, sfs, True, step, n_iters, coarse,
                                        None, part='step')
                    gb.replay()
                    graphs[k] = [ga, jobs, gb]
                else:
                    ga, jobs, gb = graphs[k]
                    ga.replay()
                    _dist.run_grad_jobs(jobs)
                    gb.replay()
                seg_iter += 1
                opt.scheduler_step_all()
            segs = {self.graph_segment_key(True, s, n_iters, coarse)
                    for s in range(n_iters)}
            if segs - set(graphs):
                slot['unusable'] = True
        elif split:
            for g3 in graphs.values():     # this call's cell selection
                g3[1] = _dist.refresh_grad_jobs(g3[1])
            for step in range(n_iters):
                ga, jobs, gb = graphs[self.graph_segment_key(
                    True, step, n_iters, coarse)]
                ga.replay()
                _dist.run_grad_jobs(jobs)
                gb.replay()
        elif first:
            # per key: first occurrence eager (lazy state gets created),
            # second captured, then replays — the occurrences need not be
            # consecutive (Co-SLAM: every 5th iteration also steps the poses)
            seen = set()
            for step in range(n_iters):
                k = self.graph_segment_key(True, step, n_iters, coarse)
                if k in graphs:
                    graphs[k].replay()
                elif k not in seen:
                    seen.add(k)
                    self._iteration(opt, sfs, True, step, n_iters, coarse,
                                    None)
                else:
                    g = torch.cuda.CUDAGraph()
                    with _capture(g):
                        self._iteration(opt, sfs, True, step, n_iters, coarse,
                                        None)
                    graphs[k] = g
                    g.replay()
                opt.scheduler_step_all()
            segs = {self.graph_segment_key(True, s, n_iters, coarse)
                    for s in range(n_iters)}
            if segs - set(graphs):
                # a one-iteration segment was never captured: replay-only
                # calls are impossible, use the ordinary path from now on
                slot['unusable'] = True
        else:
            for step in range(n_iters):
                graphs[self.graph_segment_key(True, step, n_iters,
                                              coarse)].replay()
        self.fixed_shape_batches = False
        self.before_slot_copy_out()
        if self.bundle_adjust:
            with torch.no_grad():
                for sf, f in zip(sfs, frames):
                    for ps, pf in zip(sf.get_params(), f.get_params()):
                        pf.copy_(ps.detach().to(pf.device))
        return True

    def reset_slot_optimizers(self, optimizers):
        """replay-only mapping call: the state a freshly built Optimizers
        object would have (the reference builds one per call)"""
        from ..engine.optimizers import reset_optimizer_state
        for opt in optimizers.optimizers.values():
            reset_optimizer_state(opt)

    def before_slot_copy_out(self):
        """slot call: last chance to write results into the slot frames
        before their poses are copied back to the real frames"""

    def refresh_map_selection(self):
        """replay-only mapping call: re-select what get_param_groups selected
        when the slot's optimisers were built (model hook)"""
        sel = getattr(self.model, 'select_cells', None)
        if sel is not None:
            sel()

    def optimize_update(self, n_iters, optimize_frames, is_mapping,
                        coarse=False):
        with self.lock:
            if not is_mapping and self.use_graphs and \
                    self.persistent_track_graph and len(optimize_frames) == 1 \
                    and torch.device(self.device).type == 'cuda' and \
                    all(self.config.optimizers[k]['optimizer'].accum_step is
                        None and not self.config.optimizers[k].get('scheduler')
                        for k in self.config.optimizers if
                        k.startswith('tracking_pose')):
                return self._track_slot_run(n_iters, optimize_frames[0])
            if is_mapping and self.use_graphs and self.persistent_map_graph \
                    and (not _dist.state.enabled or
                         self.persistent_map_graph_sharded) \
                    and self.is_initialized() \
                    and torch.device(self.device).type == 'cuda':
                key = self.map_slot_key(n_iters, optimize_frames, coarse)
                if key is not None and self._map_slot_run(
                        key, n_iters, optimize_frames, coarse):
                    self.after_mapping_update()
                    return None
            self.pre_precessing(optimize_frames[-1], is_mapping)
            optimizers = self.setup_optimizers(n_iters, optimize_frames,
                                               is_mapping, coarse=coarse)
            # multi-GPU: mapping gradients are summed over ranks (engine/dist)
            # (an algorithm whose mapping cannot be sharded under its current
            # options runs it replicated: identical gradients on every rank)
            optimizers.allreduce = bool(is_mapping) and \
                not getattr(self, 'replicated_mapping', False)
            track = None
            if not is_mapping:
                pdev = optimize_frames[-1].get_pose().device
                track = {
                    'loss': torch.full((), 10000000000., dtype=torch.float64,
                                       device=pdev),
                    'c2w': torch.zeros(4, 4, device=pdev),
                    'valid': torch.zeros((), dtype=torch.bool, device=pdev)}
            graphed = self._graphs_ok(optimizers, is_mapping)
            # eager_fixed_shapes: the un-compacted batches (and with them the
            # fused iteration kernels) without graph capture — what a counter
            # pass profiles (rocprofv3 --pmc with --no-graphs)
            self.fixed_shape_batches = graphed or \
                (getattr(self, 'eager_fixed_shapes', False) and
                 torch.device(self.device).type == 'cuda')
            # multi-GPU mapping: the per-iteration all-reduce stays eager
            # between a "gradient" graph and a "step" graph
            split = graphed and is_mapping and _dist.state.enabled
            seg_key, seg_iter, graph = None, 0, None
            graph_b, jobs = None, None
            args = (optimizers, optimize_frames, is_mapping)
            for step in range(n_iters):
                self.host_pre_iteration(optimize_frames, is_mapping, step)
                if graphed and split:
                    key = self.graph_segment_key(is_mapping, step, n_iters,
                                                 coarse)
                    if key != seg_key:
                        seg_key, seg_iter, graph, graph_b = key, 0, None, None
                    if seg_iter == 0:
                        self._iteration(*args, step, n_iters, coarse, track)
                    elif graph is None:
                        graph = torch.cuda.CUDAGraph()
                        # the per-rank ray sampling draws from the rank's own
                        # generator: it has to be known to the graph
                        gen = _dist.state.shard_generator
                        if gen is not None and hasattr(
                                graph, 'register_generator_state'):
                            graph.register_generator_state(gen)
                        with _capture(graph):
                            self._iteration(*args, step, n_iters, coarse,
                                            track, part='grad')
                        graph.replay()
                        jobs = _dist.collect_grad_jobs(
                            optimizers.stepping_parameters(step))
                        _dist.run_grad_jobs(jobs)
                        graph_b = torch.cuda.CUDAGraph()
                        with _capture(graph_b):
                            self._iteration(*args, step, n_iters, coarse,
                                            track, part='step')
                        graph_b.replay()
                    else:
                        graph.replay()
                        _dist.run_grad_jobs(jobs)
                        graph_b.replay()
                    seg_iter += 1
                elif graphed:
                    key = self.graph_segment_key(is_mapping, step, n_iters,
                                                 coarse)
                    if key != seg_key:
                        seg_key, seg_iter, graph = key, 0, None
                    if seg_iter == 0:
                        # first iteration of a segment runs eagerly: lazy
                        # state (Adam moments, cached frames) gets created
                        self._iteration(optimizers, optimize_frames,
                                        is_mapping, step, n_iters, coarse,
                                        track)
                    elif graph is None:
                        graph = torch.cuda.CUDAGraph()
                        with _capture(graph):
                            self._iteration(optimizers, optimize_frames,
                                            is_mapping, step, n_iters, coarse,
                                            track)
                        graph.replay()
                    else:
                        graph.replay()
                    seg_iter += 1
                else:
                    self._iteration(optimizers, optimize_frames, is_mapping,
                                    step, n_iters, coarse, track)
                optimizers.scheduler_step_all()
            self.fixed_shape_batches = False
            if is_mapping:
                self.after_mapping_update()
                return None
            if not bool(track['valid'].item()):
                return None
            return track['c2w'].cpu().numpy()

    def select_optimize_frames(self, cur_frame, keyframe_selection_method):
        """window of keyframes to optimise with (base_algorithm.py:277-302)"""
        kfs = self.keyframe_graph
        window = self.config.mapping_window_size
        if len(kfs) <= window:
            frames = kfs[:]
        elif keyframe_selection_method == 'random':
            frames = random.sample(kfs[:-1], window - 2) + [kfs[-1]]
        elif keyframe_selection_method == 'overlap':
            frames = keyframe_selection_overlap(
                camera=self.camera, cur_frame=cur_frame,
                keyframes_graph=kfs[:-1], k=window - 2,
                use_ray_sample=self.config.keyframe_use_ray_sample,
                device=self.device) + [kfs[-1]]
        elif keyframe_selection_method == 'all':
            frames = kfs.copy()
        else:
            frames = []
        if cur_frame is not None:
            frames = frames + [cur_frame]
        return frames
