"""TEST INFRASTRUCTURE ONLY.  Co-SLAM golden at the BASELINE configuration:
the REFERENCE's own ``JointEncoding`` (imported from /root/reference through
ref_harness, tiny-cuda-nn replaced by oracle/tcnn_standin.py) with its DEFAULT
config — 2^16-entry hash table, 16 levels, OneBlob 16 bins, 2x32 decoders,
32^3 smoothness samples — on the office0 mapping bound, at the ray counts of
the reference loop: 1024 tracking rays and 2048 + 341 mapping rays.

Inputs, table, decoder weights and random draws are regenerated from seeds by
tests/coslam_golden_util.py on both sides; tests/golden/coslam_office0.npz
stores the reference's outputs, loss terms and gradients (the 2 M-entry table
gradient as 65 536 seeded samples plus its sum / abs-sum / L2 norm / non-zero
count; z_vals and raw for every 16th ray).

    python oracle/make_golden_coslam_office0.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, 'tests')]
import coslam_golden_util as cg  # noqa: E402
import ref_harness  # noqa: E402
import tcnn_standin  # noqa: E402


def main():
    ref_harness.install()
    tcnn_mod = tcnn_standin.module()
    sys.modules['tinycudann'] = tcnn_mod
    import slam.model_components.encodings_coslam as enc
    enc.tcnn = tcnn_mod
    from slam.common.camera import Camera
    from slam.models.joint_encoding import JointEncoding, JointEncodingConfig

    bb = torch.from_numpy(np.array(cg.OFFICE0_BOUND, dtype=np.float64))
    cfg = JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True)
    assert cfg.hashsize == 16 and cfg.trainging_smooth_pts == 32
    model = JointEncoding(cfg, Camera(600., 600., 599.5, 339.5, 1200, 680), bb)
    model.decoder.load_state_dict(cg.office0_decoder_state(model))
    n_params = model.embed_fn.params.numel()
    with torch.no_grad():
        model.embed_fn.params.copy_(torch.from_numpy(
            cg.office0_table(n_params)))
    out = {'n_params': np.int64(n_params),
           'resolution_sdf': np.int64(model.resolution_sdf)}
    real_rand = torch.rand
    for tag, is_mapping, first, n in cg.OFFICE0_CASES:
        gen = torch.Generator().manual_seed(11)

        def seeded_rand(*shape, **kw):
            shp = shape[0] if len(shape) == 1 and not isinstance(
                shape[0], int) else shape
            return real_rand(tuple(shp), generator=gen)

        rays_o, rays_d, depth, color = cg.office0_inputs(
            n, 3 if not is_mapping else 4)
        torch.rand = seeded_rand
        try:
            for p in model.parameters():
                p.grad = None
            ro = rays_o.clone().requires_grad_(True)
            rd = rays_d.clone().requires_grad_(True)
            inp = {'rays_o': ro, 'rays_d': rd, 'target_s': color,
                   'target_d': depth, 'first': first}
            res = model.get_outputs(inp)
            ld = model.get_loss_dict(res, inp, is_mapping, 0)
            sum(ld.values()).backward()
        finally:
            torch.rand = real_rand
        s = cg.office0_summary(
            res, ld, ro, rd, model.embed_fn.params.grad,
            [(k, p.grad) for k, p in model.decoder.named_parameters()],
            n_params)
        out.update({f'{tag}/{k}': v for k, v in s.items()})
        print(tag, {k: float(v) for k, v in s.items() if k.startswith('loss')},
              'nnz', int(s['g_hash_nnz']))
    path = cg.OFFICE0
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
