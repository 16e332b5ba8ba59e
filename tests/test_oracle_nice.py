"""Pin the NICE-SLAM oracle (oracle/nice_oracle.py) against vectors produced by
the reference's own modules (tests/golden/nice_render.npz, made by
oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import nice_oracle as no
from nice_golden_util import load_nice_golden, rel_err

TOL = 1e-5  # oracle vs reference on the same CPU: pure restatement


@pytest.mark.parametrize('tag', ['coarse_map', 'middle_map', 'fine_map',
                                 'color_map', 'color_track'])
def test_oracle_matches_reference(tag):
    g, bound, grids, decs, (fx, fy, cx, cy, W, H) = load_nice_golden()
    stage, mode = tag.split('_')
    is_mapping = mode == 'map'
    c2w = torch.from_numpy(g['c2w']).requires_grad_(True)
    i, j = torch.from_numpy(g['i']), torch.from_numpy(g['j'])
    depth = torch.from_numpy(g['gt_depth'])
    color = torch.from_numpy(g['gt_color'])
    grids = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
    decs = {n: {k: v.clone().requires_grad_(True) for k, v in sd.items()}
            for n, sd in decs.items()}
    rays_o, rays_d = no.rays_from_uv(i, j, c2w, fx, fy, cx, cy)
    rays_o.retain_grad(); rays_d.retain_grad()
    out = no.render_batch_ray(rays_o, rays_d, depth, grids, decs, bound, stage)
    ld = no.loss_dict(out, depth, color, is_mapping, stage)
    loss = sum(ld.values())
    loss.backward()
    assert rel_err(out['depth'].detach(), g[f'{tag}/depth']) < TOL
    assert rel_err(out['rgb'].detach(), g[f'{tag}/rgb']) < TOL or stage != 'color'
    assert rel_err(out['uncertainty'].detach(), g[f'{tag}/uncertainty']) < TOL
    assert rel_err(loss.detach(), g[f'{tag}/loss']) < TOL
    assert rel_err(rays_o.grad, g[f'{tag}/g_rays_o']) < 1e-4
    assert rel_err(rays_d.grad, g[f'{tag}/g_rays_d']) < 1e-4
    assert rel_err(c2w.grad, g[f'{tag}/g_c2w']) < 1e-4
    for k in grids:
        key = f'{tag}/g_{k}'
        if key in g:
            assert rel_err(grids[k].grad, g[key]) < 1e-4, k
    for n in decs:
        for pn, p in decs[n].items():
            key = f'{tag}/g_dec_{n}/{pn}'
            if key in g:
                assert rel_err(p.grad, g[key]) < 1e-4, (n, pn)


def test_oracle_nodepth_render():
    g, bound, grids, decs, (fx, fy, cx, cy, W, H) = load_nice_golden()
    rays_o, rays_d = no.rays_from_uv(torch.from_numpy(g['i']),
                                     torch.from_numpy(g['j']),
                                     torch.from_numpy(g['c2w']), fx, fy, cx, cy)
    with torch.no_grad():
        out = no.render_batch_ray(rays_o, rays_d, None, grids, decs, bound,
                                  'color')
    assert rel_err(out['depth'], g['color_nodepth/depth']) < TOL
    assert rel_err(out['rgb'], g['color_nodepth/rgb']) < TOL
