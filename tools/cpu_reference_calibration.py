"""BUILD CONTAINER ONLY (needs /root/reference): how fast is the CPU *port*
that bench.py times as ``cpu_baseline`` (kind "port": the oracles / host
mirrors) against the REFERENCE's own PyTorch code on the same host, same
shapes, same thread count?  The reference tree cannot travel to the GPU box,
so the bench line carries the port's rate; this file pins the ratio once per
round:

    python tools/cpu_reference_calibration.py          # all, one child each
    -> profiles/r04_cpu_reference_calibration.json

Per algorithm one tracking-shaped and one mapping-shaped iteration (forward +
losses + backward) at the shapes of bench.py's cpu_baseline legs; the
reference's native extensions are served by the same stand-ins the port uses
(oracle/tcnn_standin, grid_standin, faiss_standin: test infrastructure), so
the ratio isolates the host code that differs.  ``port_over_reference`` =
port seconds / reference seconds (> 1: the port is slower than the reference,
the bench's cpu_baseline UNDER-states the reference's CPU rate by that
factor).  SplaTAM has no entry: its rasteriser is a CUDA-only dependency with
no CPU implementation on either side."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
OUT = os.path.join(ROOT, 'profiles', 'r04_cpu_reference_calibration.json')
THREADS = min(16, os.cpu_count() or 1)


def _best(fn, reps=2):
    fn()            # warm-up (allocator, thread pool)
    return min(_timed(fn) for _ in range(reps))


def _timed(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def nice():
    import numpy as np
    import torch
    import nice_oracle as no
    import ref_harness
    from xrdslam_amd.engine import nice as en
    torch.set_num_threads(THREADS)
    ref_harness.install()
    from slam.common.camera import Camera
    from slam.models.conv_onet import ConvOnet, ConvOnetConfig
    ConvOnet.load_pretrain = lambda self: None     # LFS pointers only
    torch.manual_seed(0)
    bb = torch.from_numpy(np.array([[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]]))
    model = ConvOnet(ConvOnetConfig(coarse=True),
                     Camera(320., 320., 319.5, 239.5, 640, 480), bb)
    dec = model.decoder

    def forward_cpu(p, c_grid, stage='middle', **kw):
        # NICE.forward hard-codes 'cuda:%d' (decoder_nice.py:388): call the
        # sub-decoders exactly as :389-414 does
        if stage in ('coarse', 'middle'):
            occ = getattr(dec, stage + '_decoder')(p, c_grid).squeeze(0)
            raw = torch.zeros(occ.shape[0], 4)
            raw[..., -1] = occ
            return raw
        fine = dec.fine_decoder(p, c_grid)
        raw = torch.zeros(fine.shape[0], 4) if stage == 'fine' else \
            dec.color_decoder(p, c_grid)
        raw[..., -1] = fine + dec.middle_decoder(p, c_grid).squeeze(0)
        return raw
    dec.forward = forward_cpu
    for k in model.grid_c:
        model.grid_c[k] = model.grid_c[k].detach().requires_grad_(True)
    grids = {k: v.detach().clone().requires_grad_() for k, v in
             model.grid_c.items()}
    decs = {kind: {n: v.detach().clone().requires_grad_() for n, v in
                   getattr(dec, kind + '_decoder').state_dict().items()}
            for kind in ('coarse', 'middle', 'fine', 'color')}
    bound = model.bounding_box.double()
    g = torch.Generator().manual_seed(0)

    def batch(n):
        o = ((torch.rand(n, 3, generator=g) - 0.5) * 2)
        d = torch.randn(n, 3, generator=g)
        d = d / d.norm(dim=1, keepdim=True)
        return o, d, 1.0 + 2.0 * torch.rand(n, 1, generator=g), \
            torch.rand(n, 3, generator=g)

    res = {}
    for tag, n, stage, is_mapping in (('track', 200, 'color', False),
                                      ('map_middle', 1000, 'middle', True),
                                      ('map_fine', 1000, 'fine', True),
                                      ('map_color', 1000, 'color', True),
                                      ('map_coarse', 1000, 'coarse', True)):
        o, d, dep, col = batch(n)

        def port():
            ro, rd = o.clone().requires_grad_(), d.clone().requires_grad_()
            out = no.render_batch_ray(ro, rd, dep, grids, decs, bound, stage)
            sum(no.loss_dict(out, dep, col, is_mapping,
                             stage).values()).backward()

        def ref():
            ro, rd = o.clone().requires_grad_(), d.clone().requires_grad_()
            inp = {'rays_o': ro, 'rays_d': rd, 'target_s': col,
                   'target_d': dep, 'stage': stage}
            out = model.get_outputs(inp)
            sum(model.get_loss_dict(out, inp, is_mapping,
                                    stage).values()).backward()
        res[tag] = {'port_s': _best(port), 'reference_s': _best(ref)}
    w = {'track': 10, 'map_middle': 24 / 5, 'map_fine': 12 / 5,
         'map_color': 24 / 5, 'map_coarse': 60 / 5}
    return res, w, ('oracle/nice_oracle.py vs the reference ConvOnet '
                    '(get_outputs + get_loss_dict + backward), office0 grids')


def coslam():
    import numpy as np
    import torch
    import ref_harness
    import tcnn_standin
    torch.set_num_threads(THREADS)
    bound = torch.from_numpy(np.array([[-3.0, 3.0], [-4.0, 2.5],
                                       [-2.0, 2.5]]))
    cam = (320., 320., 319.5, 239.5, 640, 480)
    # port: the host mirror on the oracle encodings
    import xrdslam_amd.slam.model_components.encodings_coslam as enc
    from xrdslam_amd.slam.common.camera import Camera as PCam
    from xrdslam_amd.slam.models.joint_encoding import (
        JointEncoding as PJoint, JointEncodingConfig as PCfg)
    real = enc.tcnn
    enc.tcnn = tcnn_standin.module()
    try:
        port_model = PJoint(PCfg(cam_depth_trunc=100.0, tcnn_encoding=True),
                            PCam(*cam), bound)
    finally:
        enc.tcnn = real
    ref_harness.install()
    sys.modules['tinycudann'] = tcnn_standin.module()
    import slam.model_components.encodings_coslam as renc
    renc.tcnn = sys.modules['tinycudann']
    from slam.common.camera import Camera
    from slam.models.joint_encoding import JointEncoding, JointEncodingConfig
    ref_model = JointEncoding(
        JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True),
        Camera(*cam), bound)
    g = torch.Generator().manual_seed(0)
    res = {}
    for tag, n, is_mapping in (('track', 1024, False),
                               ('map', 2048 + 341, True)):
        o = (torch.rand(n, 3, generator=g) - 0.5) * 2
        d = torch.randn(n, 3, generator=g)
        d = d / d.norm(dim=1, keepdim=True)
        dep = 1.0 + 2.0 * torch.rand(n, 1, generator=g)
        col = torch.rand(n, 3, generator=g)

        def run(model):
            inp = {'rays_o': o.clone().requires_grad_(),
                   'rays_d': d.clone().requires_grad_(), 'first': False,
                   'target_d': dep, 'target_s': col}
            out = model.get_outputs(inp)
            sum(model.get_loss_dict(out, inp, is_mapping,
                                    0).values()).backward()
        res[tag] = {'port_s': _best(lambda: run(port_model)),
                    'reference_s': _best(lambda: run(ref_model))}
    return res, {'track': 10, 'map': 10 / 5}, (
        'host mirror of JointEncoding vs the reference JointEncoding, both on '
        'oracle/tcnn_standin.py, 43 samples/ray')


def pointslam():
    import numpy as np
    import torch
    import faiss_standin
    import pointslam_golden_util as pg
    import ref_harness
    torch.set_num_threads(THREADS)
    # reference
    ref_harness.install()
    sys.modules['faiss'] = faiss_standin.module()
    import slam.model_components.neural_point_cloud as npc_mod
    npc_mod.faiss = sys.modules['faiss']
    from slam.common.camera import Camera
    from slam.models.conv_onet_pointslam import ConvOnet2, ConvOnet2Config
    ConvOnet2.load_pretrain = lambda self: None
    torch.manual_seed(0)
    ref_model = ConvOnet2(ConvOnet2Config(), Camera(*pg.TUM_CAM))
    ref_model.model_update(pg.tum_add_inputs(0))
    ref_model.get_param_groups()
    # port
    from xrdslam_amd.slam.common.camera import Camera as PCam
    from xrdslam_amd.slam.models.conv_onet_pointslam import (
        ConvOnet2 as PModel, ConvOnet2Config as PCfg)
    torch.manual_seed(0)
    port_model = PModel(PCfg(), PCam(*pg.TUM_CAM))
    port_model.knn_factory = faiss_standin.TorchKNN
    port_model.model_update(pg.tum_add_inputs(0))
    port_model.get_param_groups()
    res = {}
    shrink = 10     # bench.py's cpu_baseline leg times a tenth of the rays
    for tag, is_mapping in (('track', False), ('map', True)):
        q = pg.tum_query(is_mapping)
        n = q['o'].shape[0] // shrink

        def run(model):
            inp = {'rays_o': q['o'][:n].clone().requires_grad_(),
                   'rays_d': q['d'][:n].clone().requires_grad_(),
                   'target_s': q['color'][:n],
                   'target_d': q['depth'][:n].reshape(-1, 1),
                   'stage': 'color', 'batch_dynamic_r': q['r'][:n]}
            out = model.get_outputs(inp)
            sum(model.get_loss_dict(out, inp, is_mapping,
                                    'color').values()).backward()
        res[tag] = {'port_s': _best(lambda: run(port_model), 1),
                    'reference_s': _best(lambda: run(ref_model), 1),
                    'rays': n}
    return res, {'track': 40, 'map': 300 / 5}, (
        'host mirror of ConvOnet2 vs the reference ConvOnet2, both on the '
        'exact brute-force 8-NN stand-in, colour stage, a tenth of the '
        f'reference ray counts, cloud of {ref_model.neural_point_cloud.pts_num()} points')


def voxfusion():
    import torch
    import build_ref_octree
    import grid_standin
    import ref_harness
    import voxfusion_golden_util as vg
    torch.set_num_threads(THREADS)
    # port first (ctypes octree of the product library, host mirror)
    import xrdslam_amd.slam.model_components.voxel_helpers_voxfusion as pvh
    from xrdslam_amd.slam.common.camera import Camera as PCam
    from xrdslam_amd.slam.models.sparse_voxel import SparseVoxelConfig as PCfg
    pvh._ext = grid_standin.module()
    torch.manual_seed(0)
    port_model = PCfg().setup(camera=PCam(320., 320., 319.5, 239.5, 640, 480),
                              bounding_box=None)
    io = vg.office0_inputs()
    pts = io['points']
    rays = {'o': io['rays_o'], 'd': io['rays_d'], 'color': io['target_s'],
            'depth': io['target_d']}
    port_model.insert_points(pts, dedup=False)
    # reference: its own SparseVoxel on its own compiled octree
    ref_harness.install()
    sys.modules['grid'] = grid_standin.module()
    build_ref_octree.load()
    real_zeros = torch.zeros

    def zeros_cpu(*a, **k):
        if k.get('device') == 'cuda':
            k['device'] = 'cpu'
        return real_zeros(*a, **k)
    torch.zeros = zeros_cpu
    import slam.model_components.voxel_helpers_voxfusion as vh
    vh._ext = sys.modules['grid']
    from slam.common.camera import Camera
    from slam.models.sparse_voxel import SparseVoxel, SparseVoxelConfig
    torch.manual_seed(0)
    ref_model = SparseVoxel(SparseVoxelConfig(),
                            Camera(320., 320., 319.5, 239.5, 640, 480), None)
    torch.zeros = real_zeros
    ref_model.insert_points(pts)
    res = {}
    for tag, is_mapping in (('track', False), ('map', True)):
        def run(model):
            for p in model.parameters():
                p.grad = None
            inp = {'rays_o': rays['o'].clone().requires_grad_(),
                   'rays_d': rays['d'].clone().requires_grad_(),
                   'target_s': rays['color'], 'target_d': rays['depth']}
            out = model.get_outputs(inp)
            sum(model.get_loss_dict(out, inp, is_mapping,
                                    0).values()).backward()
        res[tag] = {'port_s': _best(lambda: run(port_model)),
                    'reference_s': _best(lambda: run(ref_model))}
    return res, {'track': 30, 'map': 15}, (
        'host mirror of SparseVoxel vs the reference SparseVoxel (its own '
        'compiled octree), both on oracle/grid_standin.py, 1024 rays of the '
        'office0-shaped golden scene')


ALGOS = {'nice-slam': nice, 'co-slam': coslam, 'point-slam': pointslam,
         'vox-fusion': voxfusion}


def child(name):
    res, weights, what = ALGOS[name]()
    port = sum(weights[k] * v['port_s'] for k, v in res.items())
    ref = sum(weights[k] * v['reference_s'] for k, v in res.items())
    print('XRD_CALIBRATION ' + json.dumps({
        'what': what, 'threads': THREADS, 'iterations': res,
        'per_frame_weights': weights, 'port_s_per_frame': port,
        'reference_s_per_frame': ref, 'port_over_reference': port / ref}))


def main():
    out = {'host': {'cores': os.cpu_count(), 'threads_used': THREADS,
                    'where': 'build container (no GPU); the GPU box\'s host '
                             'differs: use the RATIO, not the seconds'},
           'splaTAM': None,
           'splaTAM_reason': 'the reference rasteriser '
           '(diff-gaussian-rasterization-w-depth) is CUDA-only; neither side '
           'has a CPU path to time'}
    for name in ALGOS:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), name],
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines()
                if l.startswith('XRD_CALIBRATION ')]
        if not line:
            out[name] = {'error': (r.stderr or r.stdout)[-400:]}
            print(name, 'FAILED', out[name]['error'])
            continue
        out[name] = json.loads(line[0][len('XRD_CALIBRATION '):])
        print(name, 'port/reference =',
              round(out[name]['port_over_reference'], 3))
    with open(OUT, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', OUT)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        main()
