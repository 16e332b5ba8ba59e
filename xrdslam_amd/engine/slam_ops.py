"""autograd wrappers of the small fused kernels (csrc/slam_ops.hip) that
replace the tiny-op chains around the render call of one NICE-SLAM iteration:
pose -> matrix, pixel sampling + bbox filter, the losses, small-tensor Adam."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np
import torch

from .. import _lib


def _off(t: torch.Tensor, n_elems: int):
    """pointer n_elems floats into a float32 tensor"""
    return C.c_void_p(t.data_ptr() + 4 * n_elems)


class PoseQuat7Fn(torch.autograd.Function):
    """c2w[4,4] = matrix(data[7] = [t, q])  (non-separate pose parameter)"""

    @staticmethod
    def forward(ctx, data):
        c2w = torch.empty(4, 4, dtype=torch.float32, device=data.device)
        _lib.check(_lib.lib().xrd_pose_quat_fwd(
            _lib.ptr(data), _off(data, 3), _lib.ptr(c2w),
            _lib.stream_ptr(data.device)), 'xrd_pose_quat_fwd')
        ctx.save_for_backward(data)
        return c2w

    @staticmethod
    def backward(ctx, g):
        data, = ctx.saved_tensors
        out = torch.empty(7, dtype=torch.float32, device=data.device)
        _lib.check(_lib.lib().xrd_pose_quat_bwd(
            _off(data, 3), _lib.ptr(g.float().contiguous()), _lib.ptr(out),
            _off(out, 3), _lib.stream_ptr(data.device)), 'xrd_pose_quat_bwd')
        return out


class PoseQuatSplitFn(torch.autograd.Function):
    """c2w = matrix(t[3], q[4])  (separate_LR parameters)"""

    @staticmethod
    def forward(ctx, t, q):
        c2w = torch.empty(4, 4, dtype=torch.float32, device=t.device)
        _lib.check(_lib.lib().xrd_pose_quat_fwd(
            _lib.ptr(t), _lib.ptr(q), _lib.ptr(c2w),
            _lib.stream_ptr(t.device)), 'xrd_pose_quat_fwd')
        ctx.save_for_backward(q)
        return c2w

    @staticmethod
    def backward(ctx, g):
        q, = ctx.saved_tensors
        gt = torch.empty(3, dtype=torch.float32, device=q.device)
        gq = torch.empty(4, dtype=torch.float32, device=q.device)
        _lib.check(_lib.lib().xrd_pose_quat_bwd(
            _lib.ptr(q), _lib.ptr(g.float().contiguous()), _lib.ptr(gt),
            _lib.ptr(gq), _lib.stream_ptr(q.device)), 'xrd_pose_quat_bwd')
        return gt, gq


class PoseAxisAngleFn(torch.autograd.Function):
    """c2w [n,4,4] = matrix(r[n,3] axis-angle, t[n,3]) — OptimizablePose.matrix
    for any number of poses in one launch (and one for the backward)"""

    @staticmethod
    def forward(ctx, r, t):
        n = r.shape[0]
        c2w = torch.empty(n, 4, 4, dtype=torch.float32, device=r.device)
        rc, tc = r.detach().contiguous(), t.detach().contiguous()
        _lib.check(_lib.lib().xrd_pose_aa_fwd(
            n, _lib.ptr(rc), _lib.ptr(tc), _lib.ptr(c2w),
            _lib.stream_ptr(r.device)), 'xrd_pose_aa_fwd')
        ctx.save_for_backward(rc)
        return c2w

    @staticmethod
    def backward(ctx, g):
        rc, = ctx.saved_tensors
        n = rc.shape[0]
        gr = torch.empty(n, 3, dtype=torch.float32, device=rc.device)
        gt = torch.empty(n, 3, dtype=torch.float32, device=rc.device)
        _lib.check(_lib.lib().xrd_pose_aa_bwd(
            n, _lib.ptr(rc), _lib.ptr(g.float().contiguous()), _lib.ptr(gr),
            _lib.ptr(gt), _lib.stream_ptr(rc.device)), 'xrd_pose_aa_bwd')
        return gr, gt


class SampleRaysFn(torch.autograd.Function):
    """rays of F frames in one batch: returns rays_o, rays_d (differentiable
    w.r.t. the stacked c2w), target depth/colour, keep mask, dmax"""

    @staticmethod
    def forward(ctx, c2ws, idx, depth_imgs: Sequence[torch.Tensor],
                rgb_imgs: Sequence[torch.Tensor], cam, crop, bound6,
                need_dmax=True):
        lib = _lib.lib()
        F, n = idx.shape
        dev = idx.device
        N = F * n
        ro = torch.empty(N, 3, dtype=torch.float32, device=dev)
        rd = torch.empty(N, 3, dtype=torch.float32, device=dev)
        td = torch.empty(N, 1, dtype=torch.float32, device=dev)
        tc = torch.empty(N, 3, dtype=torch.float32, device=dev)
        keep = torch.empty(N, dtype=torch.uint8, device=dev)
        # (callers that do not read the batch's largest kept depth skip its
        # zero-fill launch; the kernel takes a null pointer)
        dmax = torch.zeros(1, dtype=torch.float32, device=dev) \
            if need_dmax else None
        c2ws = c2ws.detach().float().contiguous()
        H0, W0, wcrop = crop
        st = _lib.stream_ptr(dev)
        b6 = (C.c_double * 6)(*bound6)
        for f in range(F):
            _lib.check(lib.xrd_sample_rays(
                n, cam.width, H0, W0, wcrop, cam.fx, cam.fy, cam.cx, cam.cy,
                b6, C.c_void_p(idx.data_ptr() + 8 * f * n),
                _lib.ptr(depth_imgs[f]), _lib.ptr(rgb_imgs[f]),
                _off(c2ws, 16 * f), _off(ro, 3 * f * n), _off(rd, 3 * f * n),
                _off(td, f * n), _off(tc, 3 * f * n),
                C.c_void_p(keep.data_ptr() + f * n), _lib.ptr(dmax), st),
                'xrd_sample_rays')
        ctx.args = (cam, crop, F, n)
        ctx.save_for_backward(idx)
        if dmax is None:
            dmax = torch.empty(0, dtype=torch.float32, device=dev)
        ctx.mark_non_differentiable(td, tc, keep, dmax)
        # gradients of the non-differentiable outputs arrive as None instead
        # of materialised zero tensors (one fill launch each)
        ctx.set_materialize_grads(False)
        return ro, rd, td, tc, keep, dmax

    @staticmethod
    def backward(ctx, g_ro, g_rd, *unused):
        lib = _lib.lib()
        idx, = ctx.saved_tensors
        cam, (H0, W0, wcrop), F, n = ctx.args
        dev = idx.device
        z = None
        if g_ro is None or g_rd is None:
            z = torch.zeros(F * n, 3, dtype=torch.float32, device=dev)
        g_ro = z if g_ro is None else g_ro.float().contiguous()
        g_rd = z if g_rd is None else g_rd.float().contiguous()
        g_c2w = torch.empty(F, 4, 4, dtype=torch.float32, device=dev)
        st = _lib.stream_ptr(dev)
        for f in range(F):
            _lib.check(lib.xrd_sample_rays_bwd(
                n, cam.width, H0, W0, wcrop, cam.fx, cam.fy, cam.cx, cam.cy,
                C.c_void_p(idx.data_ptr() + 8 * f * n), _off(g_ro, 3 * f * n),
                _off(g_rd, 3 * f * n), _off(g_c2w, 16 * f), st),
                'xrd_sample_rays_bwd')
        return g_c2w, None, None, None, None, None, None, None


def sample_rays_poses(idx, depth_imgs, rgb_imgs, cam, crop, bound6, layout,
                      pose_params):
    """plain (no autograd) form of SampleRaysPosesFn.forward: returns
    ((rays_o, rays_d, target_d, target_rgb, keep, dmax), ctx) — ``ctx`` goes
    to sample_rays_poses_bwd"""
    lib = _lib.lib()
    F, n = idx.shape
    dev = idx.device
    N = F * n
    tp, qp, k = [], [], 0
    for lay in layout:
        if lay == '7':
            d = pose_params[k]
            tp.append(d.data_ptr())
            qp.append(d.data_ptr() + 12)
            k += 1
        else:
            tp.append(pose_params[k].data_ptr())
            qp.append(pose_params[k + 1].data_ptr())
            k += 2
    arr = C.c_void_p * F
    tp, qp = arr(*tp), arr(*qp)
    dp = arr(*[t.data_ptr() for t in depth_imgs])
    cp = arr(*[t.data_ptr() for t in rgb_imgs])
    ro = torch.empty(N, 3, dtype=torch.float32, device=dev)
    rd = torch.empty(N, 3, dtype=torch.float32, device=dev)
    td = torch.empty(N, 1, dtype=torch.float32, device=dev)
    tc = torch.empty(N, 3, dtype=torch.float32, device=dev)
    keep = torch.empty(N, dtype=torch.uint8, device=dev)
    dmax = torch.zeros(1, dtype=torch.float32, device=dev)   # atomicMax
    c2ws = torch.empty(F, 4, 4, dtype=torch.float32, device=dev)
    H0, W0, wcrop = crop
    b6 = (C.c_double * 6)(*bound6)
    _lib.check(lib.xrd_sample_rays_multi(
        F, n, cam.width, H0, W0, wcrop, cam.fx, cam.fy, cam.cx, cam.cy,
        b6, _lib.ptr(idx), dp, cp, tp, qp, _lib.ptr(c2ws), _lib.ptr(ro),
        _lib.ptr(rd), _lib.ptr(td), _lib.ptr(tc), _lib.ptr(keep),
        _lib.ptr(dmax), _lib.stream_ptr(dev)), 'xrd_sample_rays_multi')
    # (ctx[8]: the F camera matrices the launch built on the way)
    return (ro, rd, td, tc, keep, dmax), (cam, crop, F, n, layout, tp, qp,
                                          idx, c2ws)


def sample_rays_poses_bwd(ctx, g_ro, g_rd):
    """[F,7] gradient of the frames' (t, q) pose parameters"""
    cam, (H0, W0, wcrop), F, n, layout, tp, qp, idx = ctx[:8]
    dev = idx.device
    g7 = torch.empty(F, 7, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().xrd_sample_rays_multi_bwd(
        F, n, cam.width, H0, W0, wcrop, cam.fx, cam.fy, cam.cx, cam.cy,
        _lib.ptr(idx), tp, qp, _lib.ptr(g_ro.float().contiguous()),
        _lib.ptr(g_rd.float().contiguous()), _lib.ptr(g7),
        _lib.stream_ptr(dev)), 'xrd_sample_rays_multi_bwd')
    return g7


def pose_param_grads(g7, layout):
    """views of g7 in pose-parameter order (no launches)"""
    grads = []
    for f, lay in enumerate(layout):
        if lay == '7':
            grads.append(g7[f])
        else:
            grads.extend([g7[f, :3], g7[f, 3:]])
    return grads


class SampleRaysPosesFn(torch.autograd.Function):
    """SampleRaysFn with the poses given as PARAMETERS (quaternion poses): one
    launch builds the F camera matrices and samples the F frames, one launch
    returns the pose-parameter gradients.  ``pose_params``: per frame either
    one tensor data[7] = [t, q] or two tensors (t[3], q[4]), flattened in
    frame order; ``layout`` says which ('7' or 'tq') per frame."""

    @staticmethod
    def forward(ctx, idx, depth_imgs, rgb_imgs, cam, crop, bound6, layout,
                *pose_params):
        outs, ctx.args = sample_rays_poses(idx, depth_imgs, rgb_imgs, cam,
                                           crop, bound6, layout, pose_params)
        ctx.save_for_backward(idx, *pose_params)
        ctx.mark_non_differentiable(*outs[2:])
        # gradients of the non-differentiable outputs arrive as None instead
        # of materialised zero tensors (one fill launch each)
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, g_ro, g_rd, *unused):
        if g_ro is None or g_rd is None:
            z = torch.zeros_like(g_rd if g_ro is None else g_ro)
            g_ro = z if g_ro is None else g_ro
            g_rd = z if g_rd is None else g_rd
        g7 = sample_rays_poses_bwd(ctx.args, g_ro, g_rd)
        return (None, None, None, None, None, None, None,
                *pose_param_grads(g7, ctx.args[4]))


class NiceLossFn(torch.autograd.Function):
    """scalar loss of ConvOnet.get_loss_dict (sum of its terms), f64"""

    @staticmethod
    def forward(ctx, depth, var, rgb, tgt_d, tgt_rgb, keep, is_mapping,
                use_color, handle_dynamic, w_color):
        n = depth.shape[0]
        dev = depth.device
        loss = torch.empty((), dtype=torch.float64, device=dev)
        g_d = torch.empty(n, dtype=torch.float64, device=dev)
        g_c = torch.empty(n, 3, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().xrd_nice_loss(
            n, int(is_mapping), int(use_color), int(handle_dynamic),
            float(w_color), _lib.ptr(depth.detach().double().contiguous()),
            _lib.ptr(var.detach().double().contiguous()),
            _lib.ptr(rgb.detach().float().contiguous()),
            _lib.ptr(tgt_d.reshape(-1).float().contiguous()),
            _lib.ptr(tgt_rgb.float().contiguous()), _lib.ptr(keep),
            _lib.ptr(loss), _lib.ptr(g_d), _lib.ptr(g_c),
            _lib.stream_ptr(dev)), 'xrd_nice_loss')
        ctx.save_for_backward(g_d, g_c)
        return loss

    @staticmethod
    def backward(ctx, go):
        g_d, g_c = ctx.saved_tensors
        return (g_d * go, None, g_c * go.float(), None, None, None, None, None,
                None, None)


class FusedDenseAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad) for small dense CUDA tensors,
    one launch per parameter, step count on the device (graph-replayable).
    Parameters whose grad is None are skipped, like torch does."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, **other):
        # torch.optim.Adam options this kernel does not implement must not be
        # dropped silently (their defaults are accepted)
        bad = {k: v for k, v in other.items()
               if v not in (None, False) and not (k == 'foreach' and v)}
        if bad:
            raise ValueError(f'FusedDenseAdam: unsupported options {bad}')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps,
                                      weight_decay=weight_decay))

    def state_dict(self):
        """torch.optim.Adam-compatible: per-parameter exp_avg / exp_avg_sq and
        ``step`` as a 0-d float tensor (the shared device counters are
        expanded)"""
        sd = super().state_dict()
        # torch hands back the LIVE per-parameter dicts (sd['state'][i] is
        # self.state[p]): edit copies, never the running state
        sd['state'] = {k: dict(v) for k, v in sd['state'].items()}
        for st in sd['state'].values():
            if 'step' in st:
                st['step'] = st['step'].detach()[0].float().cpu()
            for k in ('exp_avg', 'exp_avg_sq'):
                if k in st:
                    st[k] = st[k].detach().clone()
        return sd

    def load_state_dict(self, state_dict):
        """accepts torch.optim.Adam state (``step`` a tensor or a number)"""
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            if 'step' in st:
                v = st['step']
                v = int(v.item()) if torch.is_tensor(v) else int(v)
                st['step'] = torch.tensor([v, 0], dtype=torch.int32,
                                          device=p.device)
                st['step']._xrd_members = [p]
            for k in ('exp_avg', 'exp_avg_sq'):
                if k in st:
                    st[k] = st[k].to(p.device, torch.float32).contiguous()

    @staticmethod
    def _consecutive(tensors):
        """the tensors are back-to-back views of one float32 buffer"""
        nxt = None
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous():
                return False
            if nxt is not None and t.data_ptr() != nxt:
                return False
            nxt = t.data_ptr() + 4 * t.numel()
        return True

    @torch.no_grad()
    def step(self, closure=None):
        for launch in self._plan():
            self._launch(launch)

    @staticmethod
    def _launch(a):
        p, g, m, v, n, (lr, b1, b2, eps, wd), step, advance = a
        _lib.check(_lib.lib().xrd_adam_dense_tick(
            _lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), n, lr, b1,
            b2, eps, wd, _lib.ptr(step), advance, _lib.stream_ptr(p.device)),
            'xrd_adam_dense_tick')

    @staticmethod
    @torch.no_grad()
    def step_together(opts):
        """the steps of several optimisers as ONE launch
        (xrd_adam_dense_multi) where each contributes a single launch with a
        step counter of its own and betas / eps / device agree; everything
        else is launched as step() would"""
        plans = [o._plan() for o in opts]
        single = [pl[0] for pl in plans if len(pl) == 1]
        for pl in plans:
            if len(pl) != 1:
                for a in pl:
                    FusedDenseAdam._launch(a)
        while single:
            a0 = single[0]
            key = (a0[5][1:4], a0[0].device)
            grp, rest, seen = [], [], set()
            for a in single:
                if (a[5][1:4], a[0].device) == key and \
                        id(a[6]) not in seen and \
                        len(grp) < _lib.ADAM_DENSE_MAX_SETS:
                    grp.append(a)
                    seen.add(id(a[6]))
                else:
                    rest.append(a)
            single = rest
            if len(grp) == 1:
                FusedDenseAdam._launch(grp[0])
                continue
            sets = (_lib.AdamDenseSet * len(grp))()
            for k, (p, g, m, v, n, (lr, b1, b2, eps, wd), step, adv) in \
                    enumerate(grp):
                sets[k].param, sets[k].grad = _lib.ptr(p), _lib.ptr(g)
                sets[k].m, sets[k].v = _lib.ptr(m), _lib.ptr(v)
                sets[k].n, sets[k].lr, sets[k].weight_decay = n, lr, wd
                sets[k].step_ticket, sets[k].advance = _lib.ptr(step), adv
            _lib.check(_lib.lib().xrd_adam_dense_multi(
                len(grp), sets, key[0][0], key[0][1], key[0][2],
                _lib.stream_ptr(key[1])), 'xrd_adam_dense_multi')

    def _plan(self):
        """-> the launches of this step: (param, grad, m, v, n, (lr, b1, b2,
        eps, wd), step counter, advance) each; sets up fresh state"""
        out = []
        for grp in self.param_groups:
            b1, b2 = grp['betas']
            live = [p for p in grp['params'] if p.grad is not None]
            if not live:
                continue
            fresh = [p for p in live if not self.state[p]]
            if fresh:
                # parameters that start together share one device-side step
                # counter {steps taken, ticket}; the last launch of the group
                # advances it (xrd_adam_dense_tick)
                step = torch.zeros(2, dtype=torch.int32,
                                   device=fresh[0].device)
                step._xrd_members = list(fresh)
                flat = len(fresh) > 1 and self._consecutive(fresh)
                if flat:
                    n = sum(p.numel() for p in fresh)
                    m = torch.zeros(n, dtype=torch.float32,
                                    device=fresh[0].device)
                    v = torch.zeros_like(m)
                    off = 0
                for p in fresh:
                    st = self.state[p]
                    if flat:
                        st['exp_avg'] = m[off:off + p.numel()].view_as(p)
                        st['exp_avg_sq'] = v[off:off + p.numel()].view_as(p)
                        off += p.numel()
                    else:
                        st['exp_avg'] = torch.zeros_like(
                            p, dtype=torch.float32)
                        st['exp_avg_sq'] = torch.zeros_like(
                            p, dtype=torch.float32)
                    st['step'] = step
            # torch.optim.Adam does not advance a parameter it skips (grad is
            # None): a shared counter is only right while ALL its members
            # step together — otherwise every member gets its own copy
            by_counter = {}
            for p in live:
                by_counter.setdefault(id(self.state[p]['step']), []).append(p)
            for ps in by_counter.values():
                cnt = self.state[ps[0]]['step']
                members = getattr(cnt, '_xrd_members', ps)
                if len(members) != len(ps):
                    for q in members:
                        own = cnt.clone()
                        own._xrd_members = [q]
                        self.state[q]['step'] = own
            args = (float(grp['lr']), float(b1), float(b2),
                    float(grp['eps']), float(grp['weight_decay']))
            grads = [p.grad if p.grad.is_contiguous() else
                     p.grad.contiguous() for p in live]
            steps = {id(self.state[p]['step']) for p in live}
            # back-to-back parameters, gradients and moments sharing one
            # counter (a decoder kept in one flat buffer whose gradient comes
            # out of one kernel): ONE launch for the whole group.  Whether the
            # parameters and moments qualify is decided once per state set-up;
            # the gradients are checked every step (they are new tensors)
            key = (len(live), len(steps))
            if fresh or grp.get('_flat_key') != key:
                grp['_flat_key'] = key
                grp['_flat_ok'] = (
                    len(live) > 1 and len(steps) == 1 and
                    len(live) == len(grp['params']) and
                    self._consecutive(live) and
                    self._consecutive([self.state[p]['exp_avg']
                                       for p in live]) and
                    self._consecutive([self.state[p]['exp_avg_sq']
                                       for p in live]))
            if grp['_flat_ok'] and self._consecutive(grads):
                st = self.state[live[0]]
                out.append((live[0], grads[0], st['exp_avg'],
                            st['exp_avg_sq'], sum(p.numel() for p in live),
                            args, st['step'], 1))
            else:
                # the last launch that uses a counter advances it
                last = {}
                for i, p in enumerate(live):
                    last[id(self.state[p]['step'])] = i
                for i, (p, g) in enumerate(zip(live, grads)):
                    st = self.state[p]
                    out.append((p, g, st['exp_avg'], st['exp_avg_sq'],
                                p.numel(), args, st['step'],
                                int(last[id(st['step'])] == i)))
            for p in live:
                # the kernel writes through the raw pointer: torch's version
                # counter does not see it; consumers that cache a derived
                # layout (packed decoder weights) watch this counter instead
                p._xrd_steps = getattr(p, '_xrd_steps', 0) + 1
        return out


@torch.no_grad()
def pose_from_matrix(c2w: torch.Tensor, rot_rep: str,
                     dev_max: torch.Tensor = None) -> torch.Tensor:
    """OptimizablePose.from_matrix on the device: c2w[4,4] f32 -> [t, rot]
    (7 floats for 'quat', 6 for 'axis_angle'), one launch, no host sync.
    ``dev_max`` (1 float on the device): the largest |c2w - matrix(rot)| seen
    is folded into it (Frame's initial-pose check without its host read)"""
    quat = rot_rep == 'quat'
    c2w = c2w.detach().float().contiguous()
    vec = torch.empty(7 if quat else 6, dtype=torch.float32,
                      device=c2w.device)
    if dev_max is not None:
        _lib.check(_lib.lib().xrd_pose_from_matrix_checked(
            1 if quat else 0, _lib.ptr(c2w), _lib.ptr(vec), _lib.ptr(dev_max),
            _lib.stream_ptr(c2w.device)), 'xrd_pose_from_matrix_checked')
        return vec
    _lib.check(_lib.lib().xrd_pose_from_matrix(
        1 if quat else 0, _lib.ptr(c2w), _lib.ptr(vec),
        _lib.stream_ptr(c2w.device)), 'xrd_pose_from_matrix')
    return vec


@torch.no_grad()
def pose_predict(prev: torch.Tensor, prev2: torch.Tensor) -> torch.Tensor:
    """constant-velocity start of the next frame on the device:
    (prev @ inv(prev2)) @ prev (tracker.py:185-199)"""
    prev = prev.detach().float().contiguous()
    prev2 = prev2.detach().float().contiguous()
    out = torch.empty(4, 4, dtype=torch.float32, device=prev.device)
    _lib.check(_lib.lib().xrd_pose_predict(
        _lib.ptr(prev), _lib.ptr(prev2), _lib.ptr(out),
        _lib.stream_ptr(prev.device)), 'xrd_pose_predict')
    return out


def track_best(loss, c2w, track):
    """track: dict(loss f64 [], c2w [4,4] f32, valid uint8 [])"""
    _lib.check(_lib.lib().xrd_track_best(
        _lib.ptr(loss.detach()), _lib.ptr(c2w.detach().contiguous()),
        _lib.ptr(track['loss']), _lib.ptr(track['c2w']),
        _lib.ptr(track['valid']), _lib.stream_ptr(c2w.device)),
        'xrd_track_best')


def sample_distinct(n_total: int, n_out: int, device, generator=None):
    """n_out distinct indices in [0, n_total) (python's random.sample on the
    device): keyed permutation evaluated at 0..n_out-1, keys from torch's RNG"""
    keys = torch.randint(-2**62, 2**62, (4, ), device=device,
                         dtype=torch.int64, generator=generator)
    out = torch.empty(n_out, dtype=torch.int64, device=device)
    _lib.check(_lib.lib().xrd_sample_distinct(
        int(n_total), int(n_out), _lib.ptr(keys), _lib.ptr(out),
        _lib.stream_ptr(torch.device(device))), 'xrd_sample_distinct')
    return out


def sample_distinct_dev(n_total_dev, n_out: int, device, generator=None):
    """sample_distinct with the population size in a device tensor (int64
    [1]) read when the launch executes: replayable from a hipGraph while the
    population grows between replays"""
    keys = torch.randint(-2**62, 2**62, (4, ), device=device,
                         dtype=torch.int64, generator=generator)
    out = torch.empty(n_out, dtype=torch.int64, device=device)
    assert n_total_dev.dtype == torch.int64 and n_total_dev.is_cuda
    _lib.check(_lib.lib().xrd_sample_distinct_dev(
        _lib.ptr(n_total_dev), int(n_out), _lib.ptr(keys), _lib.ptr(out),
        _lib.stream_ptr(torch.device(device))), 'xrd_sample_distinct_dev')
    return out


def coslam_map_rows(bank, bank_idx, rays_per_keyframe, pix, ray_dirs, rgb,
                    depth, cur_id):
    """-> (rows [n,7], ids [n] int64) of a Co-SLAM mapping batch: bank rows at
    ``bank_idx`` + the current frame's pixels ``pix`` (xrd_coslam_map_rows)"""
    dev = bank.device
    nb = 0 if bank_idx is None else bank_idx.shape[0]
    nc = pix.shape[0]
    rows = torch.empty(nb + nc, 7, dtype=torch.float32, device=dev)
    ids = torch.empty(nb + nc, dtype=torch.int64, device=dev)
    assert bank.dtype == torch.float32 and bank.is_contiguous()
    _lib.check(_lib.lib().xrd_coslam_map_rows(
        nb, _lib.ptr(bank_idx), _lib.ptr(bank), int(rays_per_keyframe), nc,
        _lib.ptr(pix), _lib.ptr(ray_dirs.contiguous()),
        _lib.ptr(rgb.contiguous()), _lib.ptr(depth.contiguous()),
        _lib.ptr(cur_id), _lib.ptr(rows), _lib.ptr(ids),
        _lib.stream_ptr(dev)), 'xrd_coslam_map_rows')
    return rows, ids


class PoseRaysFn(torch.autograd.Function):
    """rays_o, rays_d of rays with per-ray pose ids: rays_d = R[id] dir,
    rays_o = t[id]; differentiable w.r.t. c2w [n_pose,4,4].  ``rows`` [n,>=3]
    holds the camera-frame directions in its first three columns."""

    @staticmethod
    def forward(ctx, c2w, rows, ids):
        lib = _lib.lib()
        dev = rows.device
        n = rows.shape[0]
        assert rows.dtype == torch.float32 and rows.stride(1) == 1
        c = c2w.detach().float().contiguous()
        ids = ids.contiguous()
        ro = torch.empty(n, 3, dtype=torch.float32, device=dev)
        rd = torch.empty(n, 3, dtype=torch.float32, device=dev)
        _lib.check(lib.xrd_pose_rays_fwd(
            n, _lib.ptr(rows), rows.stride(0), _lib.ptr(ids), _lib.ptr(c),
            _lib.ptr(ro), _lib.ptr(rd), _lib.stream_ptr(dev)),
            'xrd_pose_rays_fwd')
        ctx.n_pose = c.shape[0]
        ctx.save_for_backward(rows, ids)
        return ro, rd

    @staticmethod
    def backward(ctx, g_ro, g_rd):
        lib = _lib.lib()
        rows, ids = ctx.saved_tensors
        dev = rows.device
        g = torch.empty(ctx.n_pose, 4, 4, dtype=torch.float32, device=dev)
        _lib.check(lib.xrd_pose_rays_bwd(
            rows.shape[0], ctx.n_pose, _lib.ptr(rows), rows.stride(0),
            _lib.ptr(ids), _lib.ptr(g_ro.float().contiguous()),
            _lib.ptr(g_rd.float().contiguous()), _lib.ptr(g),
            _lib.stream_ptr(dev)), 'xrd_pose_rays_bwd')
        return g, None, None


class SsimMapFn(torch.autograd.Function):
    """per-pixel SSIM of img1 [C,H,W] against img2 (no gradient to img2)"""

    @staticmethod
    def forward(ctx, img1, img2):
        lib = _lib.lib()
        a = img1.detach().float().contiguous()
        b = img2.detach().float().contiguous()
        Cn, H, W = a.shape
        out = torch.empty_like(a)
        need = ctx.needs_input_grad[0]
        d = [torch.empty_like(a) for _ in range(3)] if need else [None] * 3
        _lib.check(lib.xrd_ssim_fwd(Cn, H, W, _lib.ptr(a), _lib.ptr(b),
                                    _lib.ptr(out), _lib.ptr(d[0]),
                                    _lib.ptr(d[1]), _lib.ptr(d[2]),
                                    _lib.stream_ptr(a.device)), 'xrd_ssim_fwd')
        if need:
            ctx.save_for_backward(a, b, *d)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, d0, d1, d2 = ctx.saved_tensors
        Cn, H, W = a.shape
        out = torch.empty_like(a)
        _lib.check(_lib.lib().xrd_ssim_bwd(
            Cn, H, W, _lib.ptr(a), _lib.ptr(b),
            _lib.ptr(g.float().contiguous()), _lib.ptr(d0), _lib.ptr(d1),
            _lib.ptr(d2), _lib.ptr(out), _lib.stream_ptr(a.device)),
            'xrd_ssim_bwd')
        return out, None
