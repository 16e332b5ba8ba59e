"""Host side of the fused NICE-SLAM render (autograd wrapper over the C-ABI).

Replaces ``ConvOnet.render_batch_ray`` + ``eval_points`` + ``NICE.forward`` +
``raw2outputs_nerf_color`` of the reference (slam/models/conv_onet.py:339-524,
slam/model_components/decoder_nice.py:386-414,
slam/model_components/utils.py:189-244) by two kernel launches
(``xrd_nice_render_fwd`` / ``xrd_nice_render_bwd``).

PyTorch is plumbing here: it owns the device memory and the stream and carries
the gradient between the plugin's loss and the kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from .. import _lib

STAGES = {'coarse': 0, 'middle': 1, 'fine': 2, 'color': 3}
DEC_KINDS = {'coarse': 0, 'middle': 1, 'fine': 2, 'color': 3}
GRID_KEYS = ('grid_coarse', 'grid_middle', 'grid_fine', 'grid_color')

# parameter order of the flat vectors = the reference's state_dict order
# (decoder_nice.py:145-186 for MLP, :273-288 for MLP_no_xyz)


def mlp_param_shapes(c_dim: int, out_dim: int, emb: int = 93, hidden: int = 32):
    shapes = []
    for i in range(5):
        shapes += [(f'fc_c.{i}.weight', (hidden, c_dim)),
                   (f'fc_c.{i}.bias', (hidden, ))]
    shapes += [('embedder._B', (3, emb))]
    ins = [emb, hidden, hidden, hidden + emb, hidden]
    for i in range(5):
        shapes += [(f'pts_linears.{i}.weight', (hidden, ins[i])),
                   (f'pts_linears.{i}.bias', (hidden, ))]
    shapes += [('output_linear.weight', (out_dim, hidden)),
               ('output_linear.bias', (out_dim, ))]
    return shapes


def noxyz_param_shapes(c_dim: int = 32, hidden: int = 32):
    shapes = []
    ins = [hidden, hidden, hidden, hidden + c_dim, hidden]
    for i in range(5):
        shapes += [(f'pts_linears.{i}.weight', (hidden, ins[i])),
                   (f'pts_linears.{i}.bias', (hidden, ))]
    shapes += [('output_linear.weight', (1, hidden)),
               ('output_linear.bias', (1, ))]
    return shapes


def param_shapes(kind: str):
    return {'coarse': noxyz_param_shapes(),
            'middle': mlp_param_shapes(32, 1),
            'fine': mlp_param_shapes(64, 1),
            'color': mlp_param_shapes(32, 4)}[kind]


def flatten_state_dict(sd: Dict[str, torch.Tensor], kind: str) -> torch.Tensor:
    """state dict (reference key names) -> flat f32 vector"""
    parts = []
    for name, shape in param_shapes(kind):
        t = sd[name]
        assert tuple(t.shape) == tuple(shape), (name, t.shape, shape)
        parts.append(t.reshape(-1).float())
    return torch.cat(parts)


_pack_index_cache: Dict[tuple, torch.Tensor] = {}

# optional per-launch timing (bench.py): {(kernel, stage, n_rays): [(ev0, ev1)]}
# torch events are recorded on the current stream, which is the stream the
# kernels are launched on (_lib.stream_ptr).
PROFILE: Optional[Dict[tuple, list]] = None


class _Timed:
    def __init__(self, key):
        self.key = key if PROFILE is not None and \
            not torch.cuda.is_current_stream_capturing() else None

    def __enter__(self):
        if self.key is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.key is not None:
            self.e1.record()
            PROFILE.setdefault(self.key, []).append((self.e0, self.e1))


def pack_index(kind: str, device) -> torch.Tensor:
    """int64 gather index: packed = cat([flat, 0])[index]"""
    key = (kind, str(device))
    if key not in _pack_index_cache:
        lib = _lib.lib()
        k = DEC_KINDS[kind]
        n = lib.xrd_nice_pack_len(k)
        flat_len = lib.xrd_nice_flat_len(k)
        idx = np.empty(n, dtype=np.int32)
        _lib.check(lib.xrd_nice_pack_index(k, idx.ctypes.data_as(C.c_void_p)),
                   'xrd_nice_pack_index')
        idx64 = idx.astype(np.int64)
        idx64[idx64 < 0] = flat_len  # slot holding 0
        _pack_index_cache[key] = torch.from_numpy(idx64).to(device)
    return _pack_index_cache[key]


def _zero_tailed(flat: torch.Tensor) -> Optional[torch.Tensor]:
    """the [n + 1] buffer (last element 0: what the packing index points its
    padding at) a flat decoder parameter has been re-seated on, if it still
    sits there"""
    ext = getattr(flat, '_xrd_ext', None)
    if ext is not None and ext.device == flat.device and \
            ext.numel() == flat.numel() + 1 and \
            ext.data_ptr() == flat.data_ptr() and flat.is_contiguous():
        return ext
    return None


def seat_on_zero_tail(flat: torch.Tensor) -> None:
    """re-seat a decoder's flat parameter on a buffer with a trailing zero so
    that re-packing it (every iteration of a stage that trains the decoder)
    is ONE gather instead of a fill, a concatenation and a gather.  Values
    kept; idempotent; done once when the scene is built, before anything is
    captured."""
    if _zero_tailed(flat) is not None or flat.dtype != torch.float32 or \
            flat.dim() != 1:
        return
    with torch.no_grad():
        ext = torch.cat([flat.detach(), flat.new_zeros(1)])
        flat.data = ext[:-1]
    flat._xrd_ext = ext


def pack_decoder(flat: torch.Tensor, kind: str,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.lib()
    assert flat.numel() == lib.xrd_nice_flat_len(DEC_KINDS[kind]), \
        (kind, flat.numel())
    ext = _zero_tailed(flat)
    if ext is None:
        ext = torch.cat([flat.detach().float().reshape(-1),
                         flat.new_zeros(1, dtype=torch.float32)])
    idx = pack_index(kind, flat.device)
    if out is not None:
        # in place: launches captured in a hipGraph (and graphs captured
        # earlier, e.g. the tracking graph) keep reading the same buffer
        return torch.index_select(ext, 0, idx, out=out)
    return ext[idx].contiguous()


def to_channels_last_grid(val: torch.Tensor) -> torch.Tensor:
    """[1,32,Z,Y,X] tensor -> same logical tensor stored [Z][Y][X][32]"""
    assert val.dim() == 5 and val.shape[0] == 1 and val.shape[1] == 32
    return val.contiguous(memory_format=torch.channels_last_3d)


def _is_cl(val):
    return val.permute(0, 2, 3, 4, 1).is_contiguous()


class NiceScene:
    """Everything the kernels need that is not per-call: bound, grids, packed
    decoders, sampling constants.  Grids are the live parameter tensors
    (channels_last_3d); packed decoders are refreshed with ``set_decoder``."""

    def __init__(self, bound: torch.Tensor, n_samples=32, n_surface=16,
                 coarse_enlarge=2, device='cuda:0'):
        self.device = torch.device(device)
        self.bound = bound.detach().double().cpu().reshape(3, 2).clone()
        self.n_samples, self.n_surface = int(n_samples), int(n_surface)
        self.coarse_enlarge = float(coarse_enlarge)
        self.grids: Dict[str, Optional[torch.Tensor]] = {k: None
                                                          for k in GRID_KEYS}
        self.packed: Dict[str, Optional[torch.Tensor]] = {
            k: None for k in DEC_KINDS}
        self.dec_flat: Dict[str, Optional[torch.Tensor]] = {
            k: None for k in DEC_KINDS}
        # optional uint8 [Z*Y*X] per-cell masks: gradients only reach cells
        # with a non-zero byte (frustum feature selection)
        self.gmask: Dict[str, Optional[torch.Tensor]] = {}
        # exactly the tensors the reference builds (conv_onet.py:443-444,463)
        self.t_uniform = torch.linspace(0., 1., steps=self.n_samples).to(
            self.device)
        self.t_surface = torch.linspace(
            0., 1., steps=max(self.n_surface, 1)).double().to(self.device)
        _lib.check(_lib.lib().xrd_nice_warmup(), 'xrd_nice_warmup')
        _lib.check(_lib.lib().xrd_nice_map_warmup(), 'xrd_nice_map_warmup')
        self._map_ws: Dict[tuple, torch.Tensor] = {}

    def set_grid(self, key: str, val: torch.Tensor):
        assert key in GRID_KEYS
        assert val.is_cuda and val.dtype == torch.float32 and _is_cl(val), \
            f'{key}: need a CUDA f32 channels_last_3d [1,32,Z,Y,X] tensor'
        self.grids[key] = val

    def set_decoder(self, kind: str, flat: torch.Tensor):
        """flat: state_dict-ordered parameter vector (may require grad)"""
        self.dec_flat[kind] = flat
        old = self.packed.get(kind)
        if old is not None and old.device != flat.device:
            old = None
        if old is None and isinstance(flat, torch.nn.Parameter):
            seat_on_zero_tail(flat)
        self.packed[kind] = pack_decoder(flat, kind, out=old)

    def c_struct(self) -> _lib.NiceScene:
        s = _lib.NiceScene()
        b = self.bound.reshape(-1).tolist()
        for i in range(6):
            s.bound[i] = b[i]
        for gi, k in enumerate(GRID_KEYS):
            g = self.grids[k]
            s.grid[gi] = g.data_ptr() if g is not None else None
            if g is not None:
                s.gdim[3 * gi + 0], s.gdim[3 * gi + 1], s.gdim[3 * gi + 2] = \
                    g.shape[2], g.shape[3], g.shape[4]
        for ki, k in enumerate(DEC_KINDS):
            p = self.packed[k]
            s.dec[ki] = p.data_ptr() if p is not None else None
        for gi, k in enumerate(GRID_KEYS):
            m = self.gmask.get(k)
            s.gmask[gi] = m.data_ptr() if m is not None else None
        s.n_samples, s.n_surface = self.n_samples, self.n_surface
        s.t_uniform = self.t_uniform.data_ptr()
        s.t_surface = self.t_surface.data_ptr()
        s.coarse_enlarge = self.coarse_enlarge
        return s

    def n_total(self, stage: str, has_depth: bool) -> int:
        return self.n_samples + (self.n_surface
                                 if has_depth and stage != 'coarse' else 0)


def _grid_grad_buffer(g: torch.Tensor) -> torch.Tensor:
    """persistent dense gradient buffer the kernel accumulates into"""
    if g.grad is None:
        g.grad = torch.zeros_like(g, memory_format=torch.preserve_format)
    assert _is_cl(g.grad)
    return g.grad


class _NiceRenderFn(torch.autograd.Function):
    """depth(f64), var(f64), rgb(f32) = render(rays_o, rays_d, dec_color_flat)

    Grid gradients are accumulated IN PLACE into ``grid.grad`` (atomics) and
    not returned through autograd — the grids are 87 MiB and the reference
    itself only ever optimises a masked subset of them."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, dec_color_flat, g_coarse, g_middle,
                g_fine, g_color, scene: NiceScene, stage: str, gt_depth,
                dmax_in=None):
        lib = _lib.lib()
        n = rays_o.shape[0]
        dev = rays_o.device
        rays_o = rays_o.detach().float().contiguous()
        rays_d = rays_d.detach().float().contiguous()
        has_d = gt_depth is not None and stage != 'coarse'
        gd = gt_depth.detach().float().reshape(-1).contiguous() if has_d \
            else None
        if has_d:
            dmax = (dmax_in.detach().float().reshape(1) if dmax_in is not None
                    else gd.max().reshape(1))
        else:
            dmax = None
        S = scene.n_total(stage, has_d)
        depth = torch.empty(n, dtype=torch.float64, device=dev)
        var = torch.empty(n, dtype=torch.float64, device=dev)
        rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
        grid_grads = any(ctx.needs_input_grad[3:7])
        need_bwd = any(ctx.needs_input_grad[:7])
        raw = torch.empty(n, S, 4, dtype=torch.float32, device=dev) \
            if need_bwd else None
        cs = scene.c_struct()
        with _Timed(('nice_fwd', stage, n, False, False, False)):
            _lib.check(lib.xrd_nice_render_fwd(
                C.byref(cs), STAGES[stage], n, _lib.ptr(rays_o),
                _lib.ptr(rays_d), _lib.ptr(gd), _lib.ptr(dmax),
                _lib.ptr(depth), _lib.ptr(var), _lib.ptr(rgb), _lib.ptr(raw),
                _lib.stream_ptr(dev)), 'xrd_nice_render_fwd')
        ctx.scene, ctx.stage, ctx.grid_grads = scene, stage, grid_grads
        ctx.save_for_backward(rays_o, rays_d, gd, dmax, raw)
        ctx.mark_non_differentiable()
        return depth, var, rgb

    @staticmethod
    def backward(ctx, g_depth, g_var, g_rgb):
        lib = _lib.lib()
        rays_o, rays_d, gd, dmax, raw = ctx.saved_tensors
        scene, stage = ctx.scene, ctx.stage
        n = rays_o.shape[0]
        dev = rays_o.device
        need_rays = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_dec = ctx.needs_input_grad[2] and stage == 'color'
        if need_rays and stage == 'coarse':
            need_rays = False  # coarse stage never reaches the pose
        g_o = torch.zeros(n, 3, dtype=torch.float32, device=dev) \
            if need_rays else None
        g_d = torch.zeros(n, 3, dtype=torch.float32, device=dev) \
            if need_rays else None
        gg = (C.c_void_p * 4)()
        if ctx.grid_grads:
            used = {'coarse': (0, ), 'middle': (1, ), 'fine': (1, 2),
                    'color': (1, 2, 3)}[stage]
            for gi in used:
                g = scene.grids[GRID_KEYS[gi]]
                if g is not None and g.requires_grad:
                    gg[gi] = _grid_grad_buffer(g).data_ptr()
                    g._xrd_grad_fresh = True  # torch.Adam skips grad=None
        gdec = (C.c_void_p * 4)()
        g_flat = ws = None
        if need_dec:
            g_flat = torch.empty(lib.xrd_nice_flat_len(3), dtype=torch.float32,
                                 device=dev)
            gdec[3] = g_flat.data_ptr()
        if need_dec or need_rays:
            # tile partials of the ray gradients + replicas of the decoder
            # gradient (contents arbitrary on entry)
            ws = torch.empty(lib.xrd_nice_bwd_ws_floats(n),
                             dtype=torch.float32, device=dev)
        cs = scene.c_struct()
        if stage == 'coarse' and ctx.grid_grads and gg[0]:
            # zero-initialised once, kept zero by the call (static pointer:
            # replayable from a hipGraph)
            ws = getattr(scene, '_coarse_ws', None)
            need = lib.xrd_nice_coarse_ws_floats(C.byref(cs))
            if ws is None or ws.numel() != need or ws.device != dev:
                ws = scene._coarse_ws = torch.zeros(
                    need, dtype=torch.float32, device=dev)
        gdp = g_depth.double().contiguous() if g_depth is not None else None
        gvr = g_var.double().contiguous() if g_var is not None else None
        grg = g_rgb.float().contiguous() if g_rgb is not None else None
        with _Timed(('nice_bwd', stage, n, bool(need_rays), bool(need_dec),
                     bool(ctx.grid_grads))):
            _lib.check(lib.xrd_nice_render_bwd(
                C.byref(cs), STAGES[stage], n, _lib.ptr(rays_o),
                _lib.ptr(rays_d), _lib.ptr(gd), _lib.ptr(dmax), _lib.ptr(raw),
                _lib.ptr(gdp), _lib.ptr(gvr), _lib.ptr(grg), _lib.ptr(g_o),
                _lib.ptr(g_d), C.byref(gg), C.byref(gdec), _lib.ptr(ws),
                _lib.stream_ptr(dev)), 'xrd_nice_render_bwd')
        return (g_o, g_d, g_flat, None, None, None, None, None, None, None,
                None)


def nice_render(scene: NiceScene, stage: str, rays_o: torch.Tensor,
                rays_d: torch.Tensor, gt_depth: Optional[torch.Tensor] = None,
                dmax: Optional[torch.Tensor] = None):
    """Fused render.  Returns (depth f64 [n], uncertainty f64 [n], rgb [n,3]).

    ``grid_grads=True`` accumulates d(loss)/d(grid) into ``grid.grad`` of every
    grid of the stage that has ``requires_grad``; decoder-colour gradients flow
    to ``scene.dec_flat['color']`` through autograd when it requires grad."""
    if not rays_o.is_cuda:
        raise _lib.XrdError('nice_render needs CUDA tensors (no CPU fallback)')
    flat = scene.dec_flat.get('color')
    if flat is None or stage != 'color':
        flat = rays_o.new_zeros(0)
    elif not getattr(scene, 'decoder_trainable', True):
        flat = flat.detach()  # tracking / mapping_fix_color: no weight grads
    used = {'coarse': (0, ), 'middle': (1, ), 'fine': (1, 2),
            'color': (1, 2, 3)}[stage]
    empty = rays_o.new_zeros(0)
    gl = [scene.grids[GRID_KEYS[i]] if i in used else empty for i in range(4)]
    return _NiceRenderFn.apply(rays_o, rays_d, flat, gl[0], gl[1], gl[2],
                               gl[3], scene, stage, gt_depth, dmax)


@torch.no_grad()
def nice_map_iter(scene: NiceScene, stage: str, rays_o: torch.Tensor,
                  rays_d: torch.Tensor, gt_depth: torch.Tensor,
                  dmax: Optional[torch.Tensor], tgt_rgb: torch.Tensor,
                  keep: Optional[torch.Tensor], w_color: float,
                  need_rays: bool, need_dec: bool, export=None):
    """``export``: a dict that receives 'points' [n*48,3] and 'g_occ' [n*48]
    (the sample points and d loss / d occupancy logit of every sample,
    xrd_nice_map_iter_export) — what decoder_weight_grad needs.

    One mapping iteration of ``stage`` as one launch (+ one finishing
    launch): render, the mapping loss of conv_onet.py:178-184 and every
    gradient (``xrd_nice_map_iter``).  Grid gradients are accumulated into
    ``grid.grad`` of the stage's grids that require grad; returns
    (loss f64 [], g_rays_o, g_rays_d, g_dec_color) — the last three None when
    not asked for.  No autograd graph is built: the caller owns the chain
    rule beyond the rays (pose parameters) and assigns ``.grad`` itself."""
    lib = _lib.lib()
    n = rays_o.shape[0]
    dev = rays_o.device
    if not rays_o.is_cuda:
        raise _lib.XrdError('nice_map_iter needs CUDA tensors (no CPU '
                            'fallback)')
    rays_o = rays_o.detach().float().contiguous()
    rays_d = rays_d.detach().float().contiguous()
    gd = gt_depth.detach().float().reshape(-1).contiguous()
    tc = tgt_rgb.detach().float().contiguous() if tgt_rgb is not None \
        else None
    if stage != 'coarse' and dmax is None:
        dmax = gd.max()
    dm = dmax.detach().float().reshape(1) if dmax is not None else None
    cs = scene.c_struct()
    # one workspace per ray count serves the middle / fine / colour stages
    # (sized for the colour stage, the largest); the coarse stage has its own
    key = (stage == 'coarse', n)
    ws = scene._map_ws.get(key)
    if ws is None or ws.device != dev:
        # zero before the first use; every call leaves it reusable
        ws = scene._map_ws[key] = torch.zeros(
            lib.xrd_nice_map_ws_floats(
                C.byref(cs), STAGES['coarse' if stage == 'coarse' else
                                    'color'], n),
            dtype=torch.float32, device=dev)
    need_rays = bool(need_rays) and stage != 'coarse'
    need_dec = bool(need_dec) and stage == 'color'
    g_o = torch.empty(n, 3, dtype=torch.float32, device=dev) \
        if need_rays else None
    g_d = torch.empty(n, 3, dtype=torch.float32, device=dev) \
        if need_rays else None
    g_flat = torch.empty(lib.xrd_nice_flat_len(3), dtype=torch.float32,
                         device=dev) if need_dec else None
    loss = torch.empty((), dtype=torch.float64, device=dev)
    gg = (C.c_void_p * 4)()
    used = {'coarse': (0, ), 'middle': (1, ), 'fine': (1, 2),
            'color': (1, 2, 3)}[stage]
    grid_grads = False
    for gi in used:
        g = scene.grids[GRID_KEYS[gi]]
        if g is not None and g.requires_grad:
            gg[gi] = _grid_grad_buffer(g).data_ptr()
            g._xrd_grad_fresh = True  # torch.Adam skips grad=None
            grid_grads = True
    exp_p = exp_g = None
    if export is not None and stage != 'coarse':
        S = scene.n_total(stage, True)
        exp_p = torch.empty(n * S, 3, dtype=torch.float32, device=dev)
        exp_g = torch.empty(n * S, dtype=torch.float32, device=dev)
        export['points'], export['g_occ'] = exp_p, exp_g
    with _Timed(('nice_map', stage, n, need_rays, need_dec, grid_grads)):
        _lib.check(lib.xrd_nice_map_iter_export(
            C.byref(cs), STAGES[stage], n, _lib.ptr(rays_o), _lib.ptr(rays_d),
            _lib.ptr(gd), _lib.ptr(dm), _lib.ptr(tc), _lib.ptr(keep),
            float(w_color), _lib.ptr(g_o), _lib.ptr(g_d), C.byref(gg),
            _lib.ptr(g_flat), _lib.ptr(exp_p), _lib.ptr(exp_g), _lib.ptr(ws),
            _lib.ptr(loss), _lib.stream_ptr(dev)), 'xrd_nice_map_iter_export')
    return loss, g_o, g_d, g_flat


def _grid_features(grid, p, bound):
    """MLP.sample_grid_feature (decoder_nice.py:195-205): f64 normalisation of
    the points to [-1, 1] per axis, then F.grid_sample(bilinear, border,
    align_corners=True) on the [1,32,Z,Y,X] grid -> [P,32]"""
    b = bound
    pn = ((p.double() - b[:, 0]) / (b[:, 1] - b[:, 0]) * 2 - 1).float()
    c = torch.nn.functional.grid_sample(
        grid, pn[None, :, None, None], padding_mode='border',
        align_corners=True, mode='bilinear')
    return c.reshape(grid.shape[1], -1).t()


def decoder_weight_grad(scene: NiceScene, kind: str, flat: torch.Tensor,
                        points: torch.Tensor, g_out: torch.Tensor):
    """d loss / d parameters of the MIDDLE or FINE decoder from the exported
    per-sample upstream gradient (mapping_fix_fine = False, conv_onet.py:62,
    190-195; the fused launch contracts the colour decoder's weight gradient
    only): the decoder of decoder_nice.py:207-234 is re-evaluated on the
    sample points with torch ops ON THE DEVICE (grid features detached: the
    grids' own gradient comes out of the launch) and sum(g_out * occupancy) is
    back-propagated to its flat parameter.  Returns the flat gradient."""
    assert kind in ('middle', 'fine')
    F = torch.nn.functional
    with torch.enable_grad():
        w = flat.detach().requires_grad_(True)
        sd, off = {}, 0
        for name, shape in param_shapes(kind):
            k = 1
            for d in shape:
                k *= d
            sd[name] = w[off:off + k].view(shape)
            off += k
        # (the bound lives on the host; its device copy is made once, outside
        # any graph capture: the first iteration of a segment runs eagerly)
        bound = scene.__dict__.get('_bound_dev')
        if bound is None or bound.device != points.device:
            bound = scene._bound_dev = scene.bound.to(points.device)
        with torch.no_grad():
            c = _grid_features(scene.grids['grid_' + kind], points, bound)
            if kind == 'fine':      # c = [c_fine, c_middle] (:215-217)
                c = torch.cat([c, _grid_features(scene.grids['grid_middle'],
                                                 points, bound)], 1)
        emb = torch.sin(points.float() @ sd['embedder._B'])
        h = emb
        for i in range(5):
            h = F.relu(F.linear(h, sd[f'pts_linears.{i}.weight'],
                                sd[f'pts_linears.{i}.bias']))
            h = h + F.linear(c, sd[f'fc_c.{i}.weight'], sd[f'fc_c.{i}.bias'])
            if i == 2:
                h = torch.cat([emb, h], -1)
        out = F.linear(h, sd['output_linear.weight'],
                       sd['output_linear.bias']).squeeze(-1)
        (out * g_out).sum().backward()
    return w.grad


# Tracking iterations as one launch (xrd_nice_track_iter) instead of the
# forward / loss / backward launches.  OFF by default — measured on MI355X at
# the reference's 200 tracking rays (profiles/r04_nice_track_one_launch.txt):
# the one-launch kernel runs 147 us against 43 + 14 + 79 us for the chain (the
# iteration 176 vs 166 us).  200 rays are 50 blocks of 4 rays: the launch is
# bound by ONE block's dependent chain (three decoders forward, three
# backward, six staging barriers), not by throughput, and dropping the
# backward's forward recompute saves less than the grid barrier and the
# 12-wave blocks cost.  The lever for this batch size was depth, not launches:
# the three decoders of a tile on three blocks at once, 8 tile waves a block
# (round 5, DESIGN 4.1d: the three-launch chain is 28 + 9 + 31 us now).
TRACK_ONE_LAUNCH = False
# the tracking pair: one decoder per block, forward and backward; the forward
# keeps the decoders' ReLU masks for the backward
# (xrd_nice_render_fwd_masks / _bwd_masks: colour stage, <= 340 rays)
TRACK_KEEP_MASKS = True
TRACK_MASKS_MAX_RAYS = 340


@torch.no_grad()
def nice_track_iter(scene: NiceScene, rays_o: torch.Tensor,
                    rays_d: torch.Tensor, gt_depth: torch.Tensor,
                    dmax: Optional[torch.Tensor], tgt_rgb: torch.Tensor,
                    keep: Optional[torch.Tensor], use_color: bool,
                    handle_dynamic: bool, w_color: float):
    """One TRACKING iteration (colour stage) without autograd: forward render,
    the robust tracking loss of conv_onet.py:145-176 (its batch median needs
    the whole batch: a launch of its own) and the backward to the rays — the
    launches of nice_render + NiceLossFn + backward, minus the ~10 torch fill /
    scale / clone kernels autograd wraps around them.  Returns
    (loss f64 [], g_rays_o, g_rays_d)."""
    lib = _lib.lib()
    n = rays_o.shape[0]
    dev = rays_o.device
    if not rays_o.is_cuda:
        raise _lib.XrdError('nice_track_iter needs CUDA tensors (no CPU '
                            'fallback)')
    rays_o = rays_o.detach().float().contiguous()
    rays_d = rays_d.detach().float().contiguous()
    gd = gt_depth.detach().float().reshape(-1).contiguous()
    tc = tgt_rgb.detach().float().contiguous()
    dm = (dmax.detach().float().reshape(1) if dmax is not None
          else gd.max().reshape(1))
    S = scene.n_total('color', True)
    f64 = dict(dtype=torch.float64, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    cs = scene.c_struct()
    st = _lib.stream_ptr(dev)
    if TRACK_ONE_LAUNCH and S == 48 and n <= 1024:
        # forward + robust loss (batch median at a grid barrier) + backward
        # as ONE launch (csrc/nice_map.hip, TRACK variant) + its finishing
        # launch; the three-launch chain below recomputes the forward in the
        # backward
        ws = scene.__dict__.setdefault('_track_ws', {}).get(n)
        if ws is None:
            ws = scene._track_ws[n] = torch.zeros(
                lib.xrd_nice_track_ws_floats(n), **f32)
        loss = torch.empty((), **f64)
        g_o, g_d = torch.empty(n, 3, **f32), torch.empty(n, 3, **f32)
        with _Timed(('nice_track', 'color', n, True, False, False)):
            _lib.check(lib.xrd_nice_track_iter(
                C.byref(cs), n, _lib.ptr(rays_o), _lib.ptr(rays_d),
                _lib.ptr(gd), _lib.ptr(dm), _lib.ptr(tc), _lib.ptr(keep),
                int(use_color), int(handle_dynamic), float(w_color),
                _lib.ptr(g_o), _lib.ptr(g_d), _lib.ptr(ws), _lib.ptr(loss),
                st), 'xrd_nice_track_iter')
        return loss, g_o, g_d
    depth, var = torch.empty(n, **f64), torch.empty(n, **f64)
    rgb, raw = torch.empty(n, 3, **f32), torch.empty(n, S, 4, **f32)
    loss, g_dep = torch.empty((), **f64), torch.empty(n, **f64)
    g_rgb = torch.empty(n, 3, **f32)
    g_o, g_d = torch.empty(n, 3, **f32), torch.empty(n, 3, **f32)
    ws = torch.empty(lib.xrd_nice_bwd_ws_floats(n), **f32)
    # tracking-sized batches: the forward hands its ReLU masks to the backward
    # (no forward recompute there); larger ones take the plain pair
    masks = None
    if TRACK_KEEP_MASKS and S == 48 and n <= TRACK_MASKS_MAX_RAYS:
        masks = torch.empty(lib.xrd_nice_fwd_masks_words(n),
                            dtype=torch.int64, device=dev)
    with _Timed(('nice_fwd', 'color', n, False, False, False)):
        if masks is not None:
            _lib.check(lib.xrd_nice_render_fwd_masks(
                C.byref(cs), STAGES['color'], n, _lib.ptr(rays_o),
                _lib.ptr(rays_d), _lib.ptr(gd), _lib.ptr(dm), _lib.ptr(depth),
                _lib.ptr(var), _lib.ptr(rgb), _lib.ptr(raw), _lib.ptr(masks),
                st), 'xrd_nice_render_fwd_masks')
        else:
            _lib.check(lib.xrd_nice_render_fwd(
                C.byref(cs), STAGES['color'], n, _lib.ptr(rays_o),
                _lib.ptr(rays_d), _lib.ptr(gd), _lib.ptr(dm), _lib.ptr(depth),
                _lib.ptr(var), _lib.ptr(rgb), _lib.ptr(raw), st),
                'xrd_nice_render_fwd')
    _lib.check(lib.xrd_nice_loss(
        n, 0, int(use_color), int(handle_dynamic), float(w_color),
        _lib.ptr(depth), _lib.ptr(var), _lib.ptr(rgb), _lib.ptr(gd),
        _lib.ptr(tc), _lib.ptr(keep), _lib.ptr(loss), _lib.ptr(g_dep),
        _lib.ptr(g_rgb), st), 'xrd_nice_loss')
    gg, gdec = (C.c_void_p * 4)(), (C.c_void_p * 4)()
    with _Timed(('nice_bwd', 'color', n, True, False, False)):
        if masks is not None:
            _lib.check(lib.xrd_nice_render_bwd_masks(
                C.byref(cs), STAGES['color'], n, _lib.ptr(rays_o),
                _lib.ptr(rays_d), _lib.ptr(gd), _lib.ptr(dm), _lib.ptr(raw),
                _lib.ptr(g_dep), None, _lib.ptr(g_rgb), _lib.ptr(masks),
                _lib.ptr(g_o), _lib.ptr(g_d), _lib.ptr(ws), st),
                'xrd_nice_render_bwd_masks')
        else:
            _lib.check(lib.xrd_nice_render_bwd(
                C.byref(cs), STAGES['color'], n, _lib.ptr(rays_o),
                _lib.ptr(rays_d), _lib.ptr(gd), _lib.ptr(dm), _lib.ptr(raw),
                _lib.ptr(g_dep), None, _lib.ptr(g_rgb), _lib.ptr(g_o),
                _lib.ptr(g_d), C.byref(gg), C.byref(gdec), _lib.ptr(ws), st),
                'xrd_nice_render_bwd')
    return loss, g_o, g_d


@torch.no_grad()
def nice_eval_points(scene: NiceScene, stage: str,
                     points: torch.Tensor) -> torch.Tensor:
    """decoder values at free points, [n,3] -> raw [n,4] = (rgb raw or 0,
    occupancy logit); stage 'fine' or 'color' (the mesher's query_fn /
    color_func, conv_onet.py:213-240)"""
    if not points.is_cuda:
        raise _lib.XrdError('nice_eval_points needs CUDA tensors')
    st = {'fine': 2, 'color': 3}[stage]
    p = points.detach().float().reshape(-1, 3).contiguous()
    raw = torch.empty(p.shape[0], 4, dtype=torch.float32, device=p.device)
    cs = scene.c_struct()
    _lib.check(_lib.lib().xrd_nice_eval_points(
        C.byref(cs), st, p.shape[0], _lib.ptr(p), _lib.ptr(raw),
        _lib.stream_ptr(p.device)), 'xrd_nice_eval_points')
    return raw
