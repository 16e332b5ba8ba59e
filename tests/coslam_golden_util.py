"""Drives the host mirror of Co-SLAM's JointEncoding on the inputs of
tests/golden/coslam_render.npz (made by oracle/make_golden_coslam.py from the
reference's own model) and compares outputs, loss terms and gradients."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden',
                      'coslam_render.npz')
TAGS = (('track', False, False), ('map', True, False),
        ('map_first', True, True))


def build_model(g, device):
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.models.joint_encoding import (JointEncoding,
                                                        JointEncodingConfig)
    cfg = JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True,
                              hashsize=int(g['hash_cfg'][1]),
                              trainging_smooth_pts=8)
    model = JointEncoding(cfg, Camera(40., 40., 31.5, 23.5, 64, 48),
                          torch.from_numpy(g['bound']))
    assert model.resolution_sdf == int(g['hash_cfg'][0])
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files
          if k.startswith('dec/')}
    model.decoder.load_state_dict(sd)
    model = model.to(device)
    with torch.no_grad():
        assert model.embed_fn.params.numel() == g['hash_params'].size
        model.embed_fn.params.copy_(torch.from_numpy(g['hash_params']))
    return model


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def run_case(model, g, tag, is_mapping, first, device):
    draws = [torch.from_numpy(g[f'{tag}/rand{i}'])
             for i in range(int(g[f'{tag}/n_rand']))]
    it = iter(draws)

    def fed(shape, like):
        t = next(it)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.to(like)

    model._rand = fed
    for p in model.parameters():
        p.grad = None
    ro = torch.from_numpy(g['rays_o']).to(device).requires_grad_(True)
    rd = torch.from_numpy(g['rays_d']).to(device).requires_grad_(True)
    inp = {'rays_o': ro, 'rays_d': rd, 'first': first,
           'target_s': torch.from_numpy(g['target_s']).to(device),
           'target_d': torch.from_numpy(g['target_d']).to(device)}
    res = model.get_outputs(inp)
    ld = model.get_loss_dict(res, inp, is_mapping, 0)
    sum(ld.values()).backward()
    errs = {}
    for k in ('rgb', 'depth', 'depth_var', 'acc_map', 'z_vals', 'raw'):
        errs[k] = rel_err(res[k].detach().cpu().numpy(), g[f'{tag}/{k}'])
    gold_losses = [k for k in g.files if k.startswith(f'{tag}/loss_')]
    if 'data_loss' in ld:  # fused: one total for the four data terms
        gold = sum(float(g[k]) for k in gold_losses)
        errs['loss_total'] = rel_err(
            sum(v.detach().cpu().numpy() for v in ld.values()), gold)
    else:
        assert len(gold_losses) == len(ld)
        for k, v in ld.items():
            errs[f'loss_{k}'] = rel_err(v.detach().cpu().numpy(),
                                        g[f'{tag}/loss_{k}'])
    errs['g_rays_o'] = rel_err(ro.grad.cpu().numpy(), g[f'{tag}/g_rays_o'])
    errs['g_rays_d'] = rel_err(rd.grad.cpu().numpy(), g[f'{tag}/g_rays_d'])
    errs['g_hash'] = rel_err(model.embed_fn.params.grad.cpu().numpy(),
                             g[f'{tag}/g_hash'])
    for k, p in model.decoder.named_parameters():
        errs[f'g_dec/{k}'] = rel_err(p.grad.cpu().numpy(),
                                     g[f'{tag}/g_dec/{k}'])
    return errs


# ---------------------------------------------------------------------------
# BASELINE-config case: default JointEncodingConfig (2^16-entry table, 16
# levels of which 5..15 are hashed), office0 mapping bound, the ray counts of
# the reference loop (1024 tracking rays; 2048 + 341 mapping rays).  The
# fixture (tests/golden/coslam_office0.npz, oracle/make_golden_coslam_office0.py)
# stores only what cannot be regenerated: the reference's outputs.  Inputs,
# the hash table and the random draws come from seeded generators, the same
# code on both sides.
# ---------------------------------------------------------------------------
OFFICE0 = os.path.join(os.path.dirname(__file__), 'golden',
                       'coslam_office0.npz')
OFFICE0_BOUND = [[-3, 3], [-4, 2.5], [-2, 2.5]]  # input_config.py:224
OFFICE0_CASES = (('track', False, False, 1024), ('map', True, False, 2389))
HASH_SAMPLE = 65536   # table-gradient entries stored in the fixture
RAY_STRIDE = 16       # z_vals / raw are stored for every 16th ray


def office0_inputs(n, seed):
    """rays from inside the office0 bound with sensor depths (some invalid)"""
    g = torch.Generator().manual_seed(seed)
    rays_o = torch.tensor([0.3, -0.6, 0.2]) + \
        (torch.rand(n, 3, generator=g) - 0.5) * 0.8
    rays_d = torch.randn(n, 3, generator=g)
    rays_d = rays_d / rays_d.norm(dim=1, keepdim=True)
    depth = 0.5 + 2.5 * torch.rand(n, 1, generator=g)
    depth[torch.rand(n, 1, generator=g) < 0.1] = 0.0
    color = torch.rand(n, 3, generator=g)
    return rays_o, rays_d, depth, color


def office0_table(n_params, seed=21):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n_params, dtype=np.float32) *
            np.float32(0.05))


def office0_decoder_state(model, seed=22):
    """deterministic non-trivial decoder weights (the module's own init
    depends on the global RNG state at construction)"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in model.decoder.state_dict().items():
        fan = v.shape[-1] if v.dim() > 1 else v.shape[0]
        sd[k] = (torch.rand(v.shape, generator=g) * 2 - 1) / float(fan)**0.5
    return sd


def hash_sample_index(n_params, seed=23):
    return np.random.default_rng(seed).integers(0, n_params, HASH_SAMPLE)


def office0_summary(res, ld, ro, rd, hash_grad, dec_named_grads, n_params):
    """what the fixture stores of one case"""
    out = {}
    for k in ('rgb', 'depth', 'depth_var', 'acc_map'):
        out[k] = res[k].detach().cpu().numpy()
    out['z_vals'] = res['z_vals'].detach().cpu().numpy()[::RAY_STRIDE]
    out['raw'] = res['raw'].detach().cpu().numpy()[::RAY_STRIDE]
    for k, v in ld.items():
        out[f'loss_{k}'] = v.detach().cpu().numpy()
    out['g_rays_o'] = ro.grad.cpu().numpy()
    out['g_rays_d'] = rd.grad.cpu().numpy()
    gh = hash_grad.detach().cpu().numpy().astype(np.float64)
    out['g_hash_sample'] = gh[hash_sample_index(n_params)].astype(np.float32)
    out['g_hash_sum'] = np.array([gh.sum(), np.abs(gh).sum(),
                                  np.sqrt((gh * gh).sum())])
    out['g_hash_nnz'] = np.int64((gh != 0).sum())
    for k, gr in dec_named_grads:
        out[f'g_dec/{k}'] = gr.cpu().numpy().copy()
    return out


def build_office0_model(device):
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.models.joint_encoding import (JointEncoding,
                                                        JointEncodingConfig)
    cfg = JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True)
    model = JointEncoding(cfg, Camera(600., 600., 599.5, 339.5, 1200, 680),
                          torch.tensor(OFFICE0_BOUND, dtype=torch.float64))
    model.decoder.load_state_dict(office0_decoder_state(model))
    model = model.to(device)
    with torch.no_grad():
        n = model.embed_fn.params.numel()
        model.embed_fn.params.copy_(torch.from_numpy(office0_table(n)))
    return model


def run_office0_case(model, tag, is_mapping, first, n, device):
    """-> summary dict of the same layout as the fixture's"""
    gen = torch.Generator().manual_seed(11)

    def fed(shape, like):
        return torch.rand(tuple(shape), generator=gen).to(like)

    model._rand = fed
    for p in model.parameters():
        p.grad = None
    rays_o, rays_d, depth, color = office0_inputs(n, 3 if not is_mapping
                                                  else 4)
    ro = rays_o.to(device).requires_grad_(True)
    rd = rays_d.to(device).requires_grad_(True)
    inp = {'rays_o': ro, 'rays_d': rd, 'first': first,
           'target_s': color.to(device), 'target_d': depth.to(device)}
    res = model.get_outputs(inp)
    ld = model.get_loss_dict(res, inp, is_mapping, 0)
    sum(ld.values()).backward()
    return office0_summary(
        res, ld, ro, rd, model.embed_fn.params.grad,
        [(k, p.grad) for k, p in model.decoder.named_parameters()],
        model.embed_fn.params.numel())


def office0_pairs(got, gold, tag):
    """(name, got, want) for every stored quantity of one case"""
    pairs = []
    for k, v in got.items():
        key = f'{tag}/{k}'
        if k == 'g_hash_nnz':
            continue
        if k.startswith('loss_') and key not in gold.files:
            continue  # fused loss path reports one data term (see run_case)
        pairs.append((f'coslam_office0/{key}', v, gold[key]))
    return pairs


# ---------------------------------------------------------------------------
# non-default model options (tests/golden/coslam_variants.npz, made by
# ``oracle/make_golden_coslam.py variants`` from the reference's own model)
# ---------------------------------------------------------------------------
VARIANT_GOLDEN = os.path.join(os.path.dirname(__file__), 'golden',
                              'coslam_variants.npz')
VARIANTS = {
    'twogrid': dict(oneGrid=False),
    'importance': dict(training_n_importance=8),
    'twogrid_importance': dict(oneGrid=False, training_n_importance=8),
    'importance_det': dict(training_n_importance=8, training_perturb=0),
}
VARIANT_TAGS = (('track', False), ('map', True))


def variant_state(model, grids):
    """seeded tables and decoder weights, identical on the generator and the
    test side"""
    for i, (_, grid) in enumerate(sorted(grids.items())):
        with torch.no_grad():
            grid.params.copy_(torch.from_numpy(
                office0_table(grid.params.numel(), seed=31 + i)))
    model.decoder.load_state_dict(office0_decoder_state(model, seed=33))


def build_variant(g, name, device):
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.models.joint_encoding import (JointEncoding,
                                                        JointEncodingConfig)
    cfg = JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True,
                              hashsize=10, trainging_smooth_pts=8,
                              **VARIANTS[name])
    model = JointEncoding(cfg, Camera(40., 40., 31.5, 23.5, 64, 48),
                          torch.from_numpy(g['bound']))
    grids = {'embed_fn': model.embed_fn}
    if not cfg.oneGrid:
        grids['embed_fn_color'] = model.embed_fn_color
    variant_state(model, grids)
    return model.to(device), grids


def run_variant(model, grids, g, name, tag, is_mapping, device):
    """(label, got, gold) triples for tests/parity.py"""
    pre = f'{name}/{tag}'
    draws = iter([torch.from_numpy(g[f'{pre}/rand{i}'])
                  for i in range(int(g[f'{pre}/n_rand']))])

    def fed(shape, like):
        t = next(draws)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.to(like)

    model._rand = fed
    for p in model.parameters():
        p.grad = None
    ro = torch.from_numpy(g['rays_o']).to(device).requires_grad_(True)
    rd = torch.from_numpy(g['rays_d']).to(device).requires_grad_(True)
    inp = {'rays_o': ro, 'rays_d': rd, 'first': False,
           'target_s': torch.from_numpy(g['target_s']).to(device),
           'target_d': torch.from_numpy(g['target_d']).to(device)}
    res = model.get_outputs(inp)
    ld = model.get_loss_dict(res, inp, is_mapping, 0)
    sum(ld.values()).backward()
    assert next(draws, None) is None, 'unused recorded draws'
    lab = f'coslam_variants/{pre}'
    pairs = [(f'{lab}/{k}', res[k].detach().cpu().numpy(), g[f'{pre}/{k}'])
             for k in ('rgb', 'depth', 'depth_var', 'acc_map', 'z_vals',
                       'raw')]
    gold_losses = [k for k in g.files if k.startswith(f'{pre}/loss_')]
    assert len(gold_losses) == len(ld)
    pairs += [(f'{lab}/loss_{k}', v.detach().cpu().numpy(),
               g[f'{pre}/loss_{k}']) for k, v in ld.items()]
    pairs += [(f'{lab}/g_rays_o', ro.grad.cpu().numpy(), g[f'{pre}/g_rays_o']),
              (f'{lab}/g_rays_d', rd.grad.cpu().numpy(), g[f'{pre}/g_rays_d'])]
    pairs += [(f'{lab}/g_{n_}', grid.params.grad.cpu().numpy(),
               g[f'{pre}/g_{n_}']) for n_, grid in grids.items()]
    pairs += [(f'{lab}/g_dec/{k}', p.grad.cpu().numpy(), g[f'{pre}/g_dec/{k}'])
              for k, p in model.decoder.named_parameters()]
    return pairs
