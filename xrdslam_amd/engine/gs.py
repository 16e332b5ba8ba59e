"""SplaTAM per-iteration glue on the device (``csrc/gs_prepare.hip``): the
frame transform + render-variable preparation and the tracking / mapping loss
as one launch each way — what the reference assembles from ~40 small torch ops
(slam/model_components/slam_helpers_splatam.py:205-292,
slam/models/gaussian_splatting.py:102-160), two of them K = 4 GEMMs over all
Gaussians and three of them boolean-mask gathers with a host sync."""
import ctypes as C

import torch

from .. import _lib

_ACC = {}


def _acc(dev):
    a = _ACC.get(dev)
    if a is None:
        a = _ACC[dev] = torch.zeros(16, dtype=torch.float32, device=dev)
    return a


class GsPrepareFn(torch.autograd.Function):
    """(means3D [N,3], unnorm_rotations [N,4], logit_opacities [N,1],
    log_scales [N,1], pose [4,4], first_w2c [4,4]) ->
    (pts [N,3], rotations [N,4], opacities [N,1], scales [N,3],
    ds_colors [N,3]); ``pose_is_c2w`` says whether ``pose`` is the frame's
    c2w (inverted in the kernel) or already a w2c.  ``gaussians_grad`` /
    ``camera_grad`` choose which side receives gradient, like
    transform_to_frame's flags."""

    @staticmethod
    def forward(ctx, means, urot, logit, lscale, pose, first_w2c, pose_is_c2w,
                gaussians_grad, camera_grad):
        lib = _lib.lib()
        dev = means.device
        n = means.shape[0]
        if lscale.shape[-1] != 1:
            raise _lib.XrdError('GsPrepareFn: isotropic Gaussians only '
                                '(log_scales [N,1])')
        m = means.detach().float().contiguous()
        u = urot.detach().float().contiguous()
        lo = logit.detach().float().contiguous()
        ls = lscale.detach().float().contiguous()
        p = pose.detach().float().contiguous()
        fw = first_w2c.detach().float().contiguous()
        f = dict(dtype=torch.float32, device=dev)
        pts = torch.empty(n, 3, **f)
        rot = torch.empty(n, 4, **f)
        opac = torch.empty(n, 1, **f)
        scales = torch.empty(n, 3, **f)
        dscol = torch.empty(n, 3, **f)
        _lib.check(lib.xrd_gs_prepare_fwd(
            n, _lib.ptr(m), _lib.ptr(u), _lib.ptr(lo), _lib.ptr(ls),
            _lib.ptr(p), int(pose_is_c2w), _lib.ptr(fw), _lib.ptr(pts),
            _lib.ptr(rot), _lib.ptr(opac), _lib.ptr(scales), _lib.ptr(dscol),
            _lib.stream_ptr(dev)), 'xrd_gs_prepare_fwd')
        ctx.save_for_backward(m, u, lo, ls, p, fw)
        ctx.flags = (int(pose_is_c2w), bool(gaussians_grad),
                     bool(camera_grad))
        return pts, rot, opac, scales, dscol

    @staticmethod
    def backward(ctx, g_pts, g_rot, g_opac, g_scales, g_dscol):
        lib = _lib.lib()
        m, u, lo, ls, p, fw = ctx.saved_tensors
        is_c2w, g_grad, c_grad = ctx.flags
        dev = m.device
        n = m.shape[0]
        f = dict(dtype=torch.float32, device=dev)

        def c(g):
            return None if g is None else g.float().contiguous()

        g_pts, g_rot, g_opac, g_scales, g_dscol = (
            c(g_pts), c(g_rot), c(g_opac), c(g_scales), c(g_dscol))
        gm = gu = glo = gls = gp = acc = None
        if g_grad:
            gm = torch.empty(n, 3, **f)
            gu = torch.empty(n, 4, **f)
            glo = torch.empty(n, 1, **f)
            gls = torch.empty(n, 1, **f)
        if c_grad:
            gp = torch.empty(4, 4, **f)
            acc = _acc(dev)
        if g_grad or c_grad:
            _lib.check(lib.xrd_gs_prepare_bwd(
                n, _lib.ptr(m), _lib.ptr(u), _lib.ptr(lo), _lib.ptr(ls),
                _lib.ptr(p), is_c2w, _lib.ptr(fw), _lib.ptr(g_pts),
                _lib.ptr(g_rot), _lib.ptr(g_opac), _lib.ptr(g_scales),
                _lib.ptr(g_dscol), _lib.ptr(gm), _lib.ptr(gu), _lib.ptr(glo),
                _lib.ptr(gls), _lib.ptr(acc), _lib.ptr(gp),
                _lib.stream_ptr(dev)), 'xrd_gs_prepare_bwd')
        return gm, gu, glo, gls, gp, None, None, None, None


class GsLossFn(torch.autograd.Function):
    """(rgb [3,H,W], depth_sil [3,H,W], target_d [H,W], target_rgb [H,W,3])
    -> (loss_depth, loss_rgb) scalars, weights applied"""

    @staticmethod
    def forward(ctx, rgb, depth_sil, target_d, target_rgb, is_mapping,
                use_sil, sil_thres, w_depth, w_rgb, rgb_l1_scale):
        lib = _lib.lib()
        dev = rgb.device
        r = rgb.detach().float().contiguous()
        ds = depth_sil.detach().float().contiguous()
        td = target_d.detach().float().contiguous()
        tc = target_rgb.detach().float().contiguous()
        _, H, W = r.shape
        if td.numel() != H * W or tc.numel() != 3 * H * W:
            raise _lib.XrdError('GsLossFn: target shapes do not match the '
                                'render')
        stats = torch.empty(8, dtype=torch.float64, device=dev)
        ld = torch.empty((), dtype=torch.float32, device=dev)
        lc = torch.empty((), dtype=torch.float32, device=dev)
        args = (H, W, int(is_mapping), int(use_sil), float(sil_thres),
                float(w_depth), float(w_rgb), float(rgb_l1_scale))
        _lib.check(lib.xrd_gs_loss_fwd(
            *args, _lib.ptr(r), _lib.ptr(ds), _lib.ptr(td), _lib.ptr(tc),
            _lib.ptr(stats), _lib.ptr(ld), _lib.ptr(lc),
            _lib.stream_ptr(dev)), 'xrd_gs_loss_fwd')
        ctx.save_for_backward(r, ds, td, tc, stats)
        ctx.args = args
        return ld, lc

    @staticmethod
    def backward(ctx, g_d, g_c):
        lib = _lib.lib()
        r, ds, td, tc, stats = ctx.saved_tensors
        g_rgb = torch.empty_like(r)
        g_ds = torch.empty_like(ds)
        gd = None if g_d is None else g_d.float().contiguous()
        gc = None if g_c is None else g_c.float().contiguous()
        _lib.check(lib.xrd_gs_loss_bwd(
            *ctx.args, _lib.ptr(r), _lib.ptr(ds), _lib.ptr(td), _lib.ptr(tc),
            _lib.ptr(stats), _lib.ptr(gd), _lib.ptr(gc), _lib.ptr(g_rgb),
            _lib.ptr(g_ds), _lib.stream_ptr(r.device)), 'xrd_gs_loss_bwd')
        return (g_rgb, g_ds) + (None, ) * 8
