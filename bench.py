"""bench.py — tracking+mapping FPS of the MI355X-native NICE-SLAM / Co-SLAM
hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--algo nice-slam|co-slam]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): NICE-SLAM, Replica/office0 bounds,
640x480 synthetic RGB-D (analytic room), reference default
hyper-parameters (slam/configs/input_config.py:45-156): 10 tracking iterations
x 200 rays per frame; every 5th frame a mapping call of 60 iterations x 1000
rays (stages middle/fine/color) + 60 coarse iterations; 48 samples per ray.
A STEP = one frame (tracking, plus the mapping call when the frame is a map
frame).  The first-frame initialisation (mapping_first_n_iters=1500) runs in
the untimed set-up, and so does the generation of the synthetic frames: they
are resident in HBM before the timed region starts (SyntheticRoom.preload).
value = frames / second of the whole job.  Defaults: 100 timed frames after 10
warm-up frames (20 / 5 for vox-fusion and splaTAM, 5 / 2 for point-slam).
Python's cyclic garbage collector is paused inside the timed region (like
``timeit``): a generation-2 pass over the frame / graph objects landed on a
random frame and moved the Co-SLAM line between 138 and 167 frames/s
(paused: 155-156 in four runs); reference counting frees everything else.

N > 1: one process per GPU; tracking is replicated, the mapping rays are
sharded over ranks and the selected-cell/decoder gradients are summed with one
RCCL all-reduce per iteration (engine/dist.py) -> "strong" scaling of one
frame stream (the reference's 1000-ray mapping batch split N ways: at these
sizes every launch is latency-bound, so expect the all-reduce to cost about
what the smaller shards save).

--algo co-slam (single GPU) runs the same frame loop with Co-SLAM (hash grid +
OneBlob + 2x32 MLPs, 10 tracking it x 1024 rays per frame, every 5th frame 10
mapping it x (2048 bank rays + current-frame rays) with bundle adjustment,
43 samples per ray, office0 bound).  The default (nice-slam, N=1) line carries
that run as the extra object "co_slam".

The JSON line also carries
  roofline     for the dominant kernel of the timed region (per-launch HIP-event
               timing on the launch stream; algorithmic bytes from SURVEY §8d)
  cpu_baseline the CPU oracle (oracle/nice_oracle.py, a port of the reference's
               PyTorch path) timed on this host's cores on a bounded sample.
  torch_gpu_baseline  the same oracle ops run unfused by torch ON the MI355X
               (SURVEY 8d "reference PyTorch on ROCm" row), N=1 only
  config.render_img_ms  full 640x480 render_img (307 200 rays), outside the
               timed region
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

BOUND = [[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]]  # office0, input_config.py:66
NICE_TRAJ_FRAMES = 600  # samples of the synthetic trajectory: ~5 mm a frame
# the NICE leg's three-seed mean ATE must stay below this (110 frames at 5 mm
# a frame; measured 2-7 cm per seed over rounds 3-5)
NICE_ATE_BOUND = 0.08
CAM = dict(fx=320.0, fy=320.0, cx=319.5, cy=239.5, width=640, height=480)
# algorithmic HBM bytes per ray sample: 8 corners x 32 ch x 4 B
# per distinct grid lookup; backward read-modify-writes the same cells.
GRIDS_PER_STAGE = {'coarse': 1, 'middle': 1, 'fine': 2, 'color': 3}
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md (spec; 6.29e12 measured copy)
MFMA_F32_PEAK = 157.3e12  # FLOP/s, v_mfma_f32_16x16x4_f32 (= fp32 vector peak)
# cpu_baseline threads = the thread count of the port-vs-reference calibration
# (profiles/r04_cpu_reference_calibration.json was measured at 8 threads)
CPU_THREADS = 8
# algorithmic forward FLOPs per ray sample (2 x MACs
# of the decoders a stage evaluates); backward ~ 2x forward (§8d)
FWD_FLOPS = {'coarse': 12.4e3, 'middle': 31.0e3, 'fine': 72.0e3,
             'color': 103.0e3}


def calibrated(cpu, algo):
    """attach the port-vs-reference calibration of the build container
    (tools/cpu_reference_calibration.py -> profiles/
    r04_cpu_reference_calibration.json: seconds of this ``kind: port`` code
    over seconds of the REFERENCE's own PyTorch code on the same host, shapes
    and thread count) to a cpu_baseline object, plus the reference rate it
    implies on THIS host"""
    if not cpu:
        return cpu
    try:
        with open(os.path.join(ROOT, 'profiles',
                               'r04_cpu_reference_calibration.json')) as f:
            cal = json.load(f).get(algo)
        ratio = float(cal['port_over_reference'])
    except Exception:
        cpu['port_over_reference'] = None
        return cpu
    cpu['port_over_reference'] = ratio
    cpu['reference_estimate'] = {
        'value': cpu['value'] * ratio, 'unit': cpu.get('unit', 'frames/s'),
        'how': 'port rate x (port seconds / reference seconds) measured '
               'once in the build container on the same shapes '
               f'({cal["threads"]} threads there; profiles/'
               'r04_cpu_reference_calibration.json): the reference tree '
               'does not exist on the GPU box'}
    return cpu


PMC_FILE = [None]


def pmc_traffic(kernels, which='r04_pmc.json'):
    """bytes per launch of a launch group from the committed PMC pass
    (profiles/r02_pmc*.json, made by tools/run_pmc.sh on this same workload):
    sum over the group's kernels of 2 x FETCH_SIZE (gfx950 correction) +
    WRITE_SIZE; None when the file or a kernel is missing"""
    # (the newest round's pass of that name that lists every kernel; its file
    # name is left in PMC_FILE[0] for the line's traffic_source)
    PMC_FILE[0] = None
    for rnd in ('r06', 'r05', 'r04'):
        path = os.path.join(ROOT, 'profiles', rnd + which[3:])
        if not os.path.exists(path):
            continue
        pmc = json.load(open(path))
        try:
            total = 1024.0 * sum(2.0 * pmc['FETCH_SIZE'][k]['mean'] +
                                 pmc['WRITE_SIZE'][k]['mean'] for k in kernels)
        except KeyError:
            continue
        PMC_FILE[0] = rnd + which[3:]
        return total
    return None


def add_counters(roofline, kernel, which='r06_pmc.json'):
    """SQ counter evidence of the roofline's kernel from the committed
    counter passes (tools/run_pmc.sh MFMA / LDS passes, merged by
    tools/pmc_merge.py): mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs
    x 256 CUs x GRBM_GUI_ACTIVE per XCD) — the share of SIMD cycles an MFMA
    was executing, the counter-based twin of ``frac`` (which divides
    ALGORITHMIC flops by time) —, executed MFMA flops per launch (f32
    16x16x4: 64 flop a busy cycle), LDS bank-conflict share and the
    LDS-issue-stall share.  Fields stay absent when the pass is not there."""
    if roofline is None:
        return roofline
    path = os.path.join(ROOT, 'profiles', which)
    try:
        pmc = json.load(open(path))
        d = pmc['_derived'][kernel]
    except (OSError, KeyError, ValueError):
        return roofline
    for k in ('mfma_busy_frac', 'lds_conflict_frac',
              'wait_inst_lds_over_issue', 'wait_inst_any_over_issue',
              'clock_ghz', 'duration_us_profiled'):
        if k in d:
            roofline[k] = d[k]
    busy = pmc.get('SQ_VALU_MFMA_BUSY_CYCLES', {}).get(kernel, {}).get('mean')
    if busy is not None and roofline.get('dtype_flop_per_busy_cycle', 64):
        roofline['executed_mfma_flops_per_launch'] = busy * 64.0
        if roofline.get('algorithmic_flops_per_launch'):
            roofline['executed_over_algorithmic_flops'] = \
                busy * 64.0 / roofline['algorithmic_flops_per_launch']
        if roofline.get('avg_launch_us') and roofline.get('unit') == \
                'TFLOP/s':
            roofline['frac_executed'] = busy * 64.0 / (
                roofline['avg_launch_us'] * 1e-6) / MFMA_F32_PEAK
    roofline['counters_kernel'] = kernel
    roofline['counters_source'] = (
        f'profiles/{which} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES '
        'SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE | SQ_LDS_BANK_'
        'CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY '
        'SQ_ACTIVE_INST_ANY, separate passes with --kernel-trace only)')
    return roofline


def nice_group_kernels(kernel, stage, need_pose, need_dec):
    """rocprof kernel names behind one xrd_nice_render_fwd/bwd call, or one
    xrd_nice_map_iter call (forward + loss + backward)"""
    dp = 'true' if need_pose else 'false'
    dw = 'true' if (need_dec and stage == 'color') else 'false'
    if kernel == 'nice_map':
        if stage == 'coarse':
            return ['nice_map_coarse_kernel', 'nice_map_coarse_finish_kernel']
        st = {'middle': 1, 'fine': 2, 'color': 3}[stage]
        return [f'nice_map_fused<stage={st},NT=3,dp={dp},dw={dw}>',
                'nice_map_finish_kernel']
    if kernel != 'nice_bwd':
        return None
    if stage == 'coarse':
        return ['nice_bwd_coarse_kernel', 'coarse_rep_reduce_kernel']
    st = {'middle': 1, 'fine': 2, 'color': 3}[stage]
    out = [f'nice_bwd_fused<stage={st},NT=3,dp={dp},dw={dw}>']
    if need_pose or dw == 'true':
        out.append('nice_bwd_finish_kernel')
    return out


def algorithmic_flops(kernel, stage, n_rays):
    """unit = one ray-sample point; forward FWD_FLOPS, backward
    ~2x forward.  'nice_map' = the one-launch mapping iteration: forward +
    backward of the sample (8d: "fwd+bwd ~ 310 kFLOP" for the colour stage)"""
    S = 32 if stage == 'coarse' else 48
    mult = {'nice_bwd': 2.0, 'nice_map': 3.0}.get(kernel, 1.0)
    return n_rays * S * FWD_FLOPS[stage] * mult


def algorithmic_bytes(kernel, stage, n_rays, grid_grads):
    S = 32 if stage == 'coarse' else 48
    per_sample = GRIDS_PER_STAGE[stage] * 8 * 32 * 4
    if kernel == 'nice_bwd' and grid_grads:
        per_sample *= 2  # mapping: cells read + gradient read-modify-write
    if kernel == 'nice_map':
        # forward read + (mapping) gradient read-modify-write of the cells
        per_sample *= 3 if grid_grads else 2
    return n_rays * S * per_sample


def cpu_baseline(threads, device='cpu'):
    """oracle timed on the host: 1 tracking iteration (200 rays, colour stage,
    fwd+bwd) and 1 mapping iteration per stage (1000 rays) + 1 coarse, with
    office0-sized grids; converted to frames/s with the reference's iteration
    counts (10 tracking it/frame; (24 middle + 12 fine + 24 color + 60 coarse)
    mapping it / 5 frames).  With device='cuda:0' the same unfused torch ops
    run on the GPU (the "reference PyTorch on this GPU" row of SURVEY 8d)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import nice_oracle as no
    from xrdslam_amd.engine import nice as en
    torch.set_num_threads(threads)
    on_gpu = str(device) != 'cpu'
    g = torch.Generator().manual_seed(0)
    bound = torch.tensor([[-5.5, 6.0199995], [-6.7, 5.4599998],
                          [-4.7, 5.5399998]], dtype=torch.float64,
                         device=device)
    shapes = {'grid_coarse': (10, 12, 11), 'grid_middle': (31, 37, 35),
              'grid_fine': (63, 75, 71), 'grid_color': (63, 75, 71)}
    grids = {k: (torch.randn(1, 32, *s, generator=g) * 0.01).to(device)
             .requires_grad_() for k, s in shapes.items()}
    decs = {kind: {n: (torch.randn(*s, generator=g) *
                       (25. if n == 'embedder._B' else 0.2)).to(device)
                   .requires_grad_()
                   for n, s in en.param_shapes(kind)}
            for kind in ('coarse', 'middle', 'fine', 'color')}

    def one(n, stage, is_mapping):
        o = ((torch.rand(n, 3, generator=g) - 0.5) * 2).to(device) \
            .requires_grad_()
        d = torch.randn(n, 3, generator=g)
        d = (d / d.norm(dim=1, keepdim=True)).to(device).requires_grad_()
        dep = (1.0 + 2.0 * torch.rand(n, 1, generator=g)).to(device)
        col = torch.rand(n, 3, generator=g).to(device)
        if on_gpu:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = no.render_batch_ray(o, d, dep, grids, decs, bound, stage)
        loss = sum(no.loss_dict(out, dep, col, is_mapping, stage).values())
        loss.backward()
        if on_gpu:
            torch.cuda.synchronize()
        return time.perf_counter() - t0

    def best(n, stage, is_mapping):
        # the GPU leg is cheap: warm up each shape, keep the best of three
        if not on_gpu:
            return one(n, stage, is_mapping)
        one(n, stage, is_mapping)
        return min(one(n, stage, is_mapping) for _ in range(3))

    one(50, 'color', True)  # warm up the allocator / threads
    t_track = best(200, 'color', False)
    t_mid, t_fine, t_col = (best(1000, s, True)
                            for s in ('middle', 'fine', 'color'))
    t_coarse = best(1000, 'coarse', True)
    per_frame = 10 * t_track + (24 * t_mid + 12 * t_fine + 24 * t_col +
                                60 * t_coarse) / 5.0
    return {
        'value': 1.0 / per_frame, 'unit': 'frames/s',
        'cores': 0 if on_gpu else threads,
        'kind': 'port',
        'sample': ('1 tracking iter (200 rays) + 1 mapping iter per stage '
                   '(1000 rays: middle/fine/color/coarse), fwd+bwd, office0 '
                   'grids; scaled by the reference iteration counts; '
                   f'iter seconds track={t_track:.4f} middle={t_mid:.4f} '
                   f'fine={t_fine:.4f} color={t_col:.4f} '
                   f'coarse={t_coarse:.4f}' +
                   ('; unfused torch autograd ops on the MI355X (eager, no '
                    'optimizer step)' if on_gpu else ''))}


# ---- Co-SLAM (BASELINE.json north_star names it next to NICE-SLAM) ----------
CO_BOUND = [[-3.0, 3.0], [-4.0, 2.5], [-2.0, 2.5]]  # office0, co-slam config
CO_S = 43                  # 11 depth-guided + 32 uniform samples per ray
CO_FLOPS = 2 * 5184        # decoders, forward, per sample
CO_GATHER = 16 * 8 * 8     # hash-grid corner reads per sample (bytes)


def co_algorithmic(kernel, n_rays, ray_grads, map_grads):
    """(bytes, flops) of one launch, SURVEY.md 8(d) per-sample figures:
    forward 1024 B (16 levels x 8 corners x 2 features x 4 B) and 10.4 kFLOP;
    backward 2048 B read-modify-write of the table (mapping) and/or 1024 B
    re-read for the input gradient (tracking / bundle adjustment), 2 x the
    forward FLOPs.  What the kernels move ON TOP of this (forward recompute
    gather, level-major dy staging for the chunked scatter) is waste and shows
    up in `traffic` (PMC counters) against these bytes."""
    n = n_rays * CO_S
    if kernel == 'coslam_fwd':
        return n * CO_GATHER, n * CO_FLOPS
    by = (2 * CO_GATHER if map_grads else 0) + (CO_GATHER if ray_grads else 0)
    return n * max(by, CO_GATHER), n * 2 * CO_FLOPS


def co_cpu_baseline(threads):
    """the host mirror of the reference's Co-SLAM model on the CPU with the
    oracle encodings (the configuration tests/test_coslam_host.py pins against
    the reference-generated golden): 1 tracking iteration (1024 rays) and 1
    mapping iteration (2048 + 341 rays, smoothness term), fwd+bwd; converted
    with the reference iteration counts (10 tracking it/frame, 10 mapping
    it / 5 frames)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import tcnn_standin
    import xrdslam_amd.slam.model_components.encodings_coslam as enc
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.models.joint_encoding import (JointEncoding,
                                                        JointEncodingConfig)
    torch.set_num_threads(threads)
    real = enc.tcnn
    enc.tcnn = tcnn_standin.module()
    try:
        model = JointEncoding(
            JointEncodingConfig(cam_depth_trunc=100.0, tcnn_encoding=True),
            Camera(**CAM), torch.from_numpy(np.array(CO_BOUND)))
    finally:
        enc.tcnn = real
    g = torch.Generator().manual_seed(0)

    def one(n, is_mapping):
        o = ((torch.rand(n, 3, generator=g) - 0.5) * 2).requires_grad_()
        d = torch.randn(n, 3, generator=g)
        d = (d / d.norm(dim=1, keepdim=True)).requires_grad_()
        inp = {'rays_o': o, 'rays_d': d, 'first': False,
               'target_d': 1.0 + 2.0 * torch.rand(n, 1, generator=g),
               'target_s': torch.rand(n, 3, generator=g)}
        t0 = time.perf_counter()
        out = model.get_outputs(inp)
        loss = sum(model.get_loss_dict(out, inp, is_mapping, 0).values())
        loss.backward()
        return time.perf_counter() - t0

    one(64, False)
    t_track, t_map = one(1024, False), one(2048 + 341, True)
    per_frame = 10 * t_track + 10 * t_map / 5.0
    return {'value': 1.0 / per_frame, 'unit': 'frames/s', 'cores': threads,
            'kind': 'port',
            'sample': ('1 tracking iter (1024 rays) + 1 mapping iter (2389 '
                       'rays + smoothness), fwd+bwd, 43 samples/ray; scaled '
                       'by the reference iteration counts; iter seconds '
                       f'track={t_track:.3f} map={t_map:.3f}')}


def run_coslam(args, dev, with_cpu, world=1):
    """Co-SLAM frame loop -> result object.  world > 1: tracking replicated
    (shared RNG stream, pose broadcast), the mapping batch sliced over ranks,
    loss normalisers made batch-global by an all-reduce of seven sums, one
    all-reduce (SUM) of hash-table / decoder / pose gradients per step"""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.engine import coslam as ec
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import cadence, coslam_config
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)   # keyframe window sampling: same on every rank
    cfg = coslam_config(CO_BOUND)
    if args.first_iters is not None:
        cfg.mapping_first_n_iters = args.first_iters
    cam = Camera(**CAM)
    algo = cfg.setup(camera=cam, device=str(dev))
    algo.use_graphs = not args.no_graphs
    if world > 1:
        from xrdslam_amd.engine import dist as xdist
        xdist.state.setup(dev, seed=0)
    data = SyntheticRoom(CO_BOUND, H=cam.height, W=cam.width, fx=cam.fx,
                         fy=cam.fy, cx=cam.cx, cy=cam.cy,
                         n_frames=max(args.warmup + args.steps + 12, 200),
                         device=dev)
    cad = cadence['co-slam']
    getattr(data, 'data', data).preload(
        range(args.warmup + args.steps + 1))
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device=str(dev))
    for k in range(1 + args.warmup):
        slam.step(k)
    ec.PROFILE = {}
    slam.t_track = slam.t_map = 0.0
    import torch.distributed as tdist
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize()
    gc_was = _gc_pause()
    t0 = time.perf_counter()
    frame = None
    for k in range(1 + args.warmup, 1 + args.warmup + args.steps):
        frame = slam.step(k)
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _gc_resume(gc_was)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())
    if frame is not None and getattr(algo, 'persistent_map', False):
        # launches inside replayed hipGraphs cannot be event-timed one by
        # one: right after the timed region, ten more frames with mapping on
        # the per-call (eager) path feed the per-launch statistics
        algo.persistent_map = False
        nxt = 1 + args.warmup + args.steps
        for k in range(nxt, nxt + 10):        # two mapping calls
            slam.step(k)
        torch.cuda.synchronize()
        algo.persistent_map = True
    prof, ec.PROFILE = ec.PROFILE, None
    stats = []
    for key, evs in prof.items():
        ms = [a.elapsed_time(b) for a, b in evs]
        stats.append((sum(ms), key, len(ms), sum(ms) / len(ms)))
    stats.sort(reverse=True)
    total_ms, key, calls, avg_ms = stats[0]
    kernel, n_rays, ray_grads, map_grads = key
    abytes, aflops = co_algorithmic(kernel, n_rays, ray_grads, map_grads)
    hbm, mfma = abytes / (avg_ms * 1e-3), aflops / (avg_ms * 1e-3)
    compute_bound = aflops / abytes > MFMA_F32_PEAK / HBM_PEAK
    roofline = {
        'bound': 'mfma' if compute_bound else 'hbm',
        'achieved': mfma / 1e12 if compute_bound else hbm / 1e9,
        'peak': MFMA_F32_PEAK / 1e12 if compute_bound else HBM_PEAK / 1e9,
        'unit': 'TFLOP/s' if compute_bound else 'GB/s',
        'frac': mfma / MFMA_F32_PEAK if compute_bound else hbm / HBM_PEAK,
        # map + pose gradients run as two launches (map-only, pose-only)
        'traffic': pmc_traffic(
            (['coslam_bwd<dp=false,dg=true>'] if map_grads else []) +
            (['coslam_bwd<dp=true,dg=false>'] if ray_grads else []) +
            (['coslam_reduce_kernel', 'hash_chunk_scatter_runs_kernel']
             if map_grads else [])) if kernel == 'coslam_bwd'
        else pmc_traffic(['coslam_fwd_kernel']),
        'traffic_source': f'profiles/{PMC_FILE[0]} (see NICE line)',
        'intensity_flop_per_byte': aflops / abytes,
        'kernel': f'{kernel}[rays={n_rays},ray_grad={int(ray_grads)},'
                  f'map_grad={int(map_grads)}] (launch group: zero-fill, '
                  'render backward, dW reduce, chunked table scatter)'
                  if kernel == 'coslam_bwd' else f'{kernel}[rays={n_rays}]',
        'avg_launch_us': avg_ms * 1e3, 'launches': calls,
        'algorithmic_bytes_per_launch': abytes,
        'algorithmic_flops_per_launch': aflops,
        'share_of_timed_kernel_time': total_ms / sum(s[0] for s in stats),
        'note': 'launches inside replayed hipGraphs (tracking) are not '
                'event-timed; eager launches (mapping, first tracking '
                'iteration of a frame) are',
        # the 6.6 MB hash table is L2-resident: HBM is not what binds this
        # launch group.  The counters name the resource: the render backward
        # with table gradients spends most of its issue cycles waiting to
        # ISSUE LDS instructions (SQ_WAIT_INST_LDS: its per-layer operand
        # transposes at one wave a SIMD), the chunk scatter is LDS-atomic
        # bound (profiles/r04_coslam_scatter_experiments.txt)
        'binding_resource': 'LDS issue (SQ_WAIT_INST_LDS / issue cycles of '
                            'coslam_bwd<dp=false,dg=true>, hash_chunk_'
                            'scatter_runs_kernel: see wait_inst_lds_over_'
                            'issue, scatter_wait_inst_lds_over_issue)'}
    try:
        _d = json.load(open(os.path.join(ROOT, 'profiles', 'r06_pmc.json')))[
            '_derived']
        roofline['scatter_wait_inst_lds_over_issue'] = _d[
            'hash_chunk_scatter_runs_kernel']['wait_inst_lds_over_issue']
        roofline['bwd_map_wait_inst_lds_over_issue'] = _d[
            'coslam_bwd<dp=false,dg=true>']['wait_inst_lds_over_issue']
    except (OSError, KeyError, ValueError):
        pass
    return {
        'metric': 'tracking+mapping FPS @640x480',
        'value': args.steps / elapsed, 'unit': 'frames/s',
        'ms_per_step': elapsed / args.steps * 1e3, 'dtype': 'f32',
        'config': {
            'workload': 'Co-SLAM Replica/office0-shaped 640x480 RGB-D: 10 '
                        'tracking it x 1024 rays/frame + every 5th frame 10 '
                        'mapping it x (2048 bank rays + current-frame rays) '
                        'with bundle adjustment, 43 samples/ray, hash grid '
                        '16x2 (2^16) + OneBlob + 2x32 MLPs',
            'track_ms_per_frame': slam.t_track / args.steps * 1e3,
            'map_ms_per_frame': slam.t_map / args.steps * 1e3,
            'render_img_ms': render_img_ms(algo, data,
                                           args.warmup + args.steps, dev),
            'ate_rmse_m': slam.ate_rmse(),
            'ate_rmse_aligned_m': slam.trajectory_stats()[
                'absolute_translational_error.rmse']},
        'roofline': add_counters(
            # (a mapping launch group runs the ray-gradient and the
            # table-gradient backward: the counters quoted are the latter's,
            # the longer one)
            roofline, {'coslam_bwd': 'coslam_bwd<dp=false,dg=true>'
                       if map_grads else 'coslam_bwd<dp=true,dg=false>',
                       'coslam_fwd': 'coslam_fwd_kernel'}.get(kernel, kernel),
            'r06_pmc.json'),
        'cpu_baseline': calibrated(
            co_cpu_baseline(min(CPU_THREADS, os.cpu_count() or 1)), 'co-slam')
        if with_cpu else None}


def render_img_ms(algo, data, k, dev, reps=3):
    """full-image render (307 200 rays at 640x480, in the algorithm's
    ray_batch_size chunks) of frame k at its estimated pose, device to device
    + the copy of colour/depth to the host that render_img returns; best of
    ``reps`` after one warm-up.  Outside the timed FPS region."""
    c2w = algo.get_estimate_c2w_list()[k].to(dev)
    depth = data[k]['depth']
    algo.render_img(c2w, gt_depth=depth)
    best = float('inf')
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        algo.render_img(c2w, gt_depth=depth)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def _timed_frames(slam, args, dev, world):
    """untimed frame 0 + warm-up, then args.steps frames between barriers;
    -> seconds (max over ranks)"""
    import torch.distributed as tdist
    for k in range(1 + args.warmup):
        slam.step(k)
    slam.t_track = slam.t_map = 0.0
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize()
    gc_was = _gc_pause()
    t0 = time.perf_counter()
    for k in range(1 + args.warmup, 1 + args.warmup + args.steps):
        slam.step(k)
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _gc_resume(gc_was)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def _gc_pause():
    """no cyclic-GC pass inside a timed region (a generation-2 collection over
    the frame / graph objects costs tens of ms at a random frame); reference
    counting still frees everything that is not a cycle"""
    import gc
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    return was


def _gc_resume(was):
    import gc
    if was:
        gc.enable()


def _rccl_report(world):
    """what the exchange ran on and how fast it is: rank count as the
    communicator sees it and the ring all-reduce bus bandwidth of an 8 MiB
    fp32 bucket (2 (N-1)/N x bytes / time), so that a scaling run validates
    itself; None on one GPU"""
    if world <= 1:
        return None
    from xrdslam_amd.engine import dist as xdist
    st = xdist.state
    nbytes = 8 << 20
    bw = st.measure_busbw(nbytes)
    # the mapping iteration's own exchange: the flat gradient bucket the run
    # actually sent (selected grid cells + decoder + pose gradients) and what
    # one all-reduce of that size costs on this path
    bucket = int(st.stats['bucket_bytes_max'])
    ar_ms = st.measure_allreduce_ms(bucket) if bucket else None
    return {'ranks': st.comm.world if st.comm is not None else st.world,
            'exchange': st.backend_name(),
            'deterministic_shards': bool(st.deterministic),
            'allreduce_bytes': nbytes, 'busbw_GBps': bw,
            'xgmi_link_peak_GBps': 153.0,
            'bucket_bytes': bucket, 'allreduce_ms': ar_ms,
            'exchanges_enqueued_from_python': int(st.stats['exchanges'])}


def _per_rank(world, **values):
    """gather scalars of every rank on rank 0: {name: [rank 0, rank 1, ...]}
    (what the north star's "mapping-step speed-up at 8 GPUs" is read from);
    every rank must call it"""
    if world <= 1:
        return {k: [v] for k, v in values.items()}
    import torch.distributed as dist
    box = [None] * world
    dist.all_gather_object(box, values)
    return {k: [b[k] for b in box] for k in values}


def _setup_dist(dev, world):
    if world > 1:
        from xrdslam_amd.engine import dist as xdist
        xdist.state.setup(dev, seed=0)


VOX_FLOPS = 2 * (16 * 128 + 128 * 128 + 129 * 128 + 144 * 128 + 3 * 128)
VOX_BYTES = 8 * 4 + 12 + 8 * 16 * 4   # 556 B per point forward


def vox_cpu_baseline(threads, leaf_voxels=800):
    """the host mirror of the reference's SparseVoxel on the CPU with the C
    oracle standing in for the two CUDA operators (the configuration
    tests/test_voxfusion_host.py pins against the reference-made golden): one
    tracking-sized iteration (1024 rays, pose gradient only) and one
    mapping-sized one (1024 rays, map + decoder gradients), forward+backward;
    converted with the reference iteration counts (30 + 15 per frame, the
    mapping window of the first frames: 1 frame)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import grid_standin
    import xrdslam_amd.slam.model_components.voxel_helpers_voxfusion as vh
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.models.sparse_voxel import SparseVoxelConfig
    torch.set_num_threads(threads)
    real = vh._ext
    vh._ext = grid_standin.module()
    try:
        torch.manual_seed(0)
        model = SparseVoxelConfig().setup(camera=Camera(**CAM),
                                          bounding_box=None)
        g = torch.Generator().manual_seed(0)
        # a wall of surface points 2 m in front of the camera at the 10 m
        # offset Vox-Fusion works at
        n_pts = 40000
        pts = torch.stack([10 + (torch.rand(n_pts, generator=g) - 0.5) * 4,
                           10 + (torch.rand(n_pts, generator=g) - 0.5) * 3,
                           torch.full((n_pts, ), 8.0)], 1)
        model.insert_points(pts, dedup=False)

        def one(n, is_mapping):
            for p in model.parameters():
                p.grad = None
            o = torch.tensor([10.0, 10.0, 10.0]).expand(n, 3).clone() \
                .requires_grad_()
            d = torch.stack([(torch.rand(n, generator=g) - 0.5) * 1.2,
                             (torch.rand(n, generator=g) - 0.5) * 0.9,
                             -torch.ones(n)], 1).requires_grad_()
            inp = {'rays_o': o, 'rays_d': d,
                   'target_d': torch.full((n, 1), 2.0),
                   'target_s': torch.rand(n, 3, generator=g)}
            t0 = time.perf_counter()
            out = model.get_outputs(inp)
            loss = sum(model.get_loss_dict(out, inp, is_mapping, 0).values())
            loss.backward()
            return time.perf_counter() - t0

        one(64, True)
        t_track, t_map = one(1024, False), one(1024, True)
    finally:
        vh._ext = real
    per_frame = 30 * t_track + 15 * t_map
    return {'value': 1.0 / per_frame, 'unit': 'frames/s', 'cores': threads,
            'kind': 'port',
            'sample': ('1 tracking iter + 1 mapping iter (1024 rays each, '
                       'sampled inside the hit voxels at 1 cm), fwd+bwd, host '
                       'mirror of SparseVoxel + C oracle of the two CUDA '
                       'operators; scaled by the reference iteration counts '
                       f'(30 + 15 per frame); iter seconds track='
                       f'{t_track:.3f} map={t_map:.3f}')}


def run_voxfusion(args, dev, world=1):
    """Vox-Fusion frame loop (every frame tracked with 30 it x 1024 rays and
    mapped with 15 it x 1024 rays x <=6 frames; relative poses + 10 m offset)
    on the fused ray pipeline (csrc/vox_rays.hip) + voxel-feature / decoder
    kernels (csrc/vox_render.hip), every iteration inside a captured graph."""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import (cadence,
                                                       voxfusion_config)
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)   # keyframe window sampling: same on every rank
    cam = Camera(**CAM)
    algo = voxfusion_config().setup(camera=cam, device=str(dev))
    _setup_dist(dev, world)
    data = SyntheticRoom(CO_BOUND, H=cam.height, W=cam.width, fx=cam.fx,
                         fy=cam.fy, cx=cam.cx, cy=cam.cy,
                         n_frames=max(args.warmup + args.steps + 4, 200),
                         device=dev)
    cad = cadence['vox-fusion']
    getattr(data, 'data', data).preload(
        range(args.warmup + args.steps + 1))
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device=str(dev),
                          use_relative_pose=cad.use_relative_pose,
                          init_pose_offset=cad.init_pose_offset)
    from xrdslam_amd.engine import vox as evox
    algo.use_graphs = not args.no_graphs
    elapsed = _timed_frames(slam, args, dev, world)
    t_track, t_map = slam.t_track, slam.t_map
    sizes = dict(getattr(algo, 'last_batch_sizes', None) or {})
    pose = algo.get_estimate_c2w_list()[-1].to(dev)
    algo.render_img(pose)
    torch.cuda.synchronize(dev)
    t_r = time.perf_counter()
    algo.render_img(pose)
    torch.cuda.synchronize(dev)
    render_ms = (time.perf_counter() - t_r) * 1e3
    # per-launch HIP-event timing: replayed graph nodes cannot be event-timed,
    # so two more frames run eagerly (same kernels, same shapes) right after
    # the timed region
    algo.use_graphs = False
    evox.PROFILE = {}
    log = []
    orig = algo.optimize_update

    def logged(n_iters, frames, is_mapping, coarse=False):
        out = orig(n_iters, frames, is_mapping, coarse=coarse)
        log.append((is_mapping,
                    dict(getattr(algo, 'last_batch_sizes', None) or {})))
        return out
    algo.optimize_update = logged
    for k in range(1 + args.warmup + args.steps,
                   3 + args.warmup + args.steps):
        slam.step(k)
    torch.cuda.synchronize(dev)
    algo.optimize_update = orig
    prof, evox.PROFILE = evox.PROFILE, None
    roofline = None
    if prof:
        groups = {}
        for key, evs in prof.items():
            g_ = groups.setdefault((key[0], bool(key[2]) if len(key) > 2
                                    else False), [0.0, 0])
            g_[0] += sum(a.elapsed_time(b) for a, b in evs)
            g_[1] += len(evs)
        per_launch = {f'{k[0]}[decoder_grad={int(k[1])}]': v[0] / v[1] * 1e3
                      for k, v in groups.items()}
        # the dominant MFMA launch (the ray-side launches are in launch_us)
        (kern, need_w), (ms, calls) = max(
            ((k, v) for k, v in groups.items()
             if k[0].startswith(('vox_points', 'vox_dw'))),
            key=lambda kv: kv[1][0])
        # live points of a launch: the size record of the probe's calls
        pts = np.mean([s_['n_pts'] for m_, s_ in log
                       if m_ == need_w and s_]) if log else 0.0
        bwd = kern.endswith('bwd')
        mlp = kern.startswith('vox_points')
        dw = kern.startswith('vox_dw')
        flops = pts * VOX_FLOPS * (2 if bwd else 1)
        # vox_dw reads the operand rows the forward / backward left in HBM:
        # x [16] + h1, h2, f, hc [128 each] + gc3 [4] + ghc, gf, gh2, gh1
        # [128 each] = 4176 B a point — what-if builds (tools/whatif_build.py,
        # profiles/r06_vox_dw_whatif.txt) show the launch follows that read
        # stream, not its MFMAs
        byts = pts * (4176 if dw else
                      VOX_BYTES + (1024 if bwd and need_w else 0))
        us = ms / calls * 1e3
        roofline = {
            'bound': 'mfma', 'achieved': flops / (us * 1e-6) / 1e12,
            'peak': MFMA_F32_PEAK / 1e12, 'unit': 'TFLOP/s',
            'frac': flops / (us * 1e-6) / MFMA_F32_PEAK,
            # the counter pass of THIS variant (tracking: no weight-gradient
            # operands; mapping: with them — two kernel names since round 5)
            'traffic': pmc_traffic(
                {'vox_dw': ['vox_dw_kernel', 'vox_dw_reduce_kernel'],
                 'vox_points_fwd': ['vox_points_fwd_kernel'],
                 'vox_points_bwd': ['vox_points_bwd<dw=%s>' %
                                    ('true' if need_w else 'false')]}[kern],
                'r04_pmc_vox.json'),
            'traffic_source': f'profiles/{PMC_FILE[0]} (rocprofv3 --pmc '
                              'FETCH_SIZE / WRITE_SIZE passes of this '
                              'workload, FETCH x2 on gfx950)',
            'kernel': f'{kern}[decoder_grad={int(need_w)}]' + (
                ' (launch: gather, trilinear feature, 16-128-128-129 / '
                '144-128-3 decoder' + (', embedding scatter, dW operands'
                                       if bwd else '') + ')' if mlp else
                ' (launch: the five weight-gradient contractions over the '
                'live points + bias sums)' if dw else ''),
            'avg_launch_us': us, 'launches': calls,
            'avg_points_per_launch': float(pts),
            'algorithmic_flops_per_point': VOX_FLOPS * (2 if bwd else 1),
            'other_bound': {'bound': 'hbm', 'unit': 'GB/s',
                            'achieved': byts / (us * 1e-6) / 1e9,
                            'frac': byts / (us * 1e-6) / HBM_PEAK},
            'launch_us': per_launch,
            'timing_source': 'HIP events around the eager launches of two '
                             'frames run right after the timed region (the '
                             'timed region replays captured graphs)'}
        if dw:
            # the binding resource first: the operand read stream
            ob = roofline['other_bound']
            roofline['other_bound'] = {
                'bound': 'mfma', 'unit': 'TFLOP/s',
                'achieved': roofline['achieved'],
                'peak': MFMA_F32_PEAK / 1e12, 'frac': roofline['frac']}
            roofline.update(bound='hbm', achieved=ob['achieved'],
                            peak=HBM_PEAK / 1e9, unit='GB/s',
                            frac=ob['frac'],
                            algorithmic_bytes_per_point=4176)
        # the launch group with the largest share of the FRAME: a mapping
        # iteration's decoder trio (forward, backward with the weight-
        # gradient operands, the weight products) — 15 iterations a frame
        # against 30 tracking iterations of the cheaper decoder_grad=0 pair
        # the roofline above quotes — with its OWN counter traffic
        trio = [per_launch.get(f'{k}[decoder_grad=1]') for k in
                ('vox_points_fwd', 'vox_points_bwd', 'vox_dw')]
        if all(t is not None for t in trio):
            pts_w = np.mean([s_['n_pts'] for m_, s_ in log
                             if m_ and s_]) if log else 0.0
            t_us = float(sum(trio))
            fl = pts_w * VOX_FLOPS * 3.0      # forward + (input + weight) grads
            tr = pmc_traffic(['vox_points_fwd_kernel',
                              'vox_points_bwd<dw=true>', 'vox_dw_kernel',
                              'vox_dw_reduce_kernel'], 'r04_pmc_vox.json')
            roofline['mapping_group'] = {
                'kernels': 'vox_points_fwd + vox_points_bwd<dw> + vox_dw '
                           '(+ reduce) of one mapping iteration',
                'launch_group_us': t_us, 'points': float(pts_w),
                'algorithmic_flops': fl,
                'frac': fl / (t_us * 1e-6) / MFMA_F32_PEAK if t_us else None,
                'traffic': tr,
                'algorithmic_bytes': pts_w * (VOX_BYTES + 1024),
                'traffic_over_algorithmic': (
                    tr / (pts_w * (VOX_BYTES + 1024)) if tr and pts_w else
                    None),
                'share_of_frame': 15.0 * t_us * 1e-3 /
                (elapsed / args.steps * 1e3),
                'traffic_source': f'profiles/{PMC_FILE[0]}'}
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = calibrated(vox_cpu_baseline(min(os.cpu_count() or 1, CPU_THREADS)),
                         'vox-fusion')
    return {
        'metric': 'tracking+mapping FPS @640x480',
        'value': args.steps / elapsed, 'unit': 'frames/s',
        'ms_per_step': elapsed / args.steps * 1e3, 'dtype': 'f32',
        'config': {
            'workload': 'Vox-Fusion 640x480 synthetic RGB-D: 30 tracking it x '
                        '1024 rays + 15 mapping it x 1024 rays x <=6 frames '
                        'every frame, 0.2 m voxels, 16-d embeddings, 2x128 '
                        'MLP',
            'track_ms_per_frame': t_track / args.steps * 1e3,
            'map_ms_per_frame': t_map / args.steps * 1e3,
            'render_img_ms': render_ms,
            'ate_rmse_m': slam.ate_rmse(),
            'leaf_voxels': int(algo.model.svo.count_leaf_nodes()),
            'last_batch': sizes,
            'graphs': bool(not args.no_graphs)},
        'roofline': add_counters(
            roofline, None if roofline is None else
            {'vox_dw': 'vox_dw_kernel', 'vox_points_fwd':
             'vox_points_fwd_kernel', 'vox_points_bwd':
             'vox_points_bwd<dw=%s>' % ('true' if need_w else 'false')}[kern],
            'r06_pmc_vox.json'), 'cpu_baseline': cpu}


class _CvPoses:
    """synthetic sequence with OpenCV-convention poses (camera looks down +z,
    what SplaTAM's back-projection assumes) and numpy images"""

    def __init__(self, data):
        self.data = data

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        d = dict(self.data[i])
        c2w = np.array(d['c2w'], dtype=np.float64)
        c2w[:3, 1] *= -1
        c2w[:3, 2] *= -1
        d['c2w'] = c2w
        for k in ('rgb', 'depth'):
            if torch.is_tensor(d[k]):
                d[k] = d[k].cpu().numpy()
        return d


def run_splatam(args, dev, world=1):
    """SplaTAM frame loop: 40 tracking + 60 mapping iterations per frame, two
    full-image raster passes (colour; depth/silhouette) per iteration over
    ~4e5 Gaussians, on the HIP rasteriser (tile binning on the device)."""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import cadence, splatam_config
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)   # keyframe window sampling: same on every rank
    cam = Camera(**CAM)
    algo = splatam_config().setup(camera=cam, device=str(dev))
    algo.use_graphs = not args.no_graphs
    # N > 1: tracking replicated, every mapping iteration's image split into
    # tile-row bands over the ranks, Gaussian gradients all-reduced
    _setup_dist(dev, world)
    data = _CvPoses(SyntheticRoom(
        CO_BOUND, H=cam.height, W=cam.width, fx=cam.fx, fy=cam.fy, cx=cam.cx,
        cy=cam.cy, n_frames=max(args.warmup + args.steps + 1, 200),
        device=dev))
    cad = cadence['splaTAM']
    getattr(data, 'data', data).preload(
        range(args.warmup + args.steps + 1))
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device=str(dev),
                          use_relative_pose=cad.use_relative_pose)
    elapsed = _timed_frames(slam, args, dev, world)
    t_track, t_map = slam.t_track, slam.t_map
    # per-launch HIP-event timing of one more frame (100 iterations, 200
    # raster passes each way) right after the timed region
    # (eagerly: events cannot be read back from inside a captured graph)
    from xrdslam_amd.compat import diff_gaussian_rasterization as dgr
    algo.use_graphs = False
    dgr.PROFILE = {}
    slam.step(1 + args.warmup + args.steps)
    torch.cuda.synchronize()
    prof, dgr.PROFILE = dgr.PROFILE, None
    us = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) * 1e3
          for k, v in prof.items() if k.startswith('gs_')}
    pairs = float(torch.stack(prof['pairs']).double().mean())
    keys = float(torch.stack(prof['keys']).double().mean())
    n_g = float(np.mean(prof['gaussians']))
    # per (pixel x contributing Gaussian) pair ~30 FLOP forward,
    # ~80 backward (VALU fp32, same 157.3 TFLOP/s peak as v_mfma_f32); per
    # Gaussian and pass 48 B read + ~64 B written by the preprocess, 16 B of
    # key/value per (Gaussian x tile) pair through the sort
    # one DUAL pass per view (rgb and depth / silhouette colours blended by
    # the same weights): SURVEY's 80 FLOP per pair backward + ~30 for the
    # second colour set
    flops_per_pair = 110.0
    bwd_flops = pairs * flops_per_pair
    hbm_bytes = n_g * (48 + 64) + keys * 16 * 2
    pre_us = us.get('gs_preprocess', 0.0) + us.get('gs_bin', 0.0)
    roofline = {
        'bound': 'mfma', 'achieved': bwd_flops / (us['gs_render_bwd'] * 1e-6)
        / 1e12, 'peak': MFMA_F32_PEAK / 1e12, 'unit': 'TFLOP/s',
        'frac': bwd_flops / (us['gs_render_bwd'] * 1e-6) / MFMA_F32_PEAK,
        'traffic': pmc_traffic(['gs_blend_bwd_kernel',
                                'gs_key_reduce_kernel'],
                               'r04_pmc_splatam.json'),
        'traffic_source': f'profiles/{PMC_FILE[0]} (rocprofv3 --pmc '
                          'FETCH_SIZE / WRITE_SIZE passes of this workload, '
                          'FETCH x2 on gfx950)',
        'kernel': 'xrd_gs_blend_bwd<dual> = gs_blend_bwd_kernel + '
                  'gs_key_reduce_kernel (blend backward of the rgb and the '
                  'depth/silhouette colours in one pass, front to back: one '
                  'wave per 8x8 sub-tile with exact sub-tile culling, list '
                  'entries as packed records through the scalar path, '
                  'transposed in-wave reduction + LDS rows, no global '
                  'atomics; fp32 VALU — the peak is the fp32 FMA rate, equal '
                  'to the f32 MFMA peak; algorithmic pairs = Gaussians up to '
                  'each pixel\'s last contributor, as the published kernels '
                  'evaluate them)',
        'avg_launch_us': us['gs_render_bwd'],
        'launches': len(prof['gs_render_bwd']),
        'pixel_gaussian_pairs_per_pass': pairs,
        'gaussian_tile_pairs_per_pass': keys, 'gaussians': n_g,
        'algorithmic_flops_per_pair': flops_per_pair,
        'other_bound': {
            'bound': 'hbm', 'unit': 'GB/s', 'what': 'preprocess + binning',
            'achieved': hbm_bytes / (pre_us * 1e-6) / 1e9 if pre_us else None,
            'frac': hbm_bytes / (pre_us * 1e-6) / HBM_PEAK if pre_us
            else None},
        'launch_us': us,
        'gaussian_tile_pairs_per_s': keys / ((us['gs_render_fwd'] +
                                              us['gs_render_bwd']) * 1e-6),
        'timing_source': 'HIP events around the launches of the frame run '
                         'right after the timed region'}
    cpu = None
    if not args.no_cpu_baseline:
        cpu = splatam_cpu_baseline(min(os.cpu_count() or 1, CPU_THREADS), n_g,
                                   cam.height * cam.width)
    return {
        'metric': 'tracking+mapping FPS @640x480',
        'value': args.steps / elapsed, 'unit': 'frames/s',
        'ms_per_step': elapsed / args.steps * 1e3, 'dtype': 'f32',
        'config': {
            'workload': 'SplaTAM 640x480 synthetic RGB-D: 40 tracking it + 60 '
                        'mapping it per frame, rgb + depth/silhouette render '
                        'of one view each (the reference: 2 raster passes; '
                        'here one dual-colour pass), window 24',
            'track_ms_per_frame': t_track / args.steps * 1e3,
            'map_ms_per_frame': t_map / args.steps * 1e3,
            'ate_rmse_m': slam.ate_rmse(),
            'gaussians': int(algo.model.gaussian_cloud.params['means3D']
                             .shape[0]),
            'binning_overflowed_passes': dgr._BIN.overflowed},
        'roofline': add_counters(roofline, 'gs_blend_bwd_kernel',
                                 'r06_pmc_splatam.json'),
        'cpu_baseline': cpu}


def splatam_cpu_baseline(threads, n_gaussians, n_pixels):
    """no CPU baseline for SplaTAM: the reference's rasteriser
    (diff-gaussian-rasterization-w-depth) is a CUDA-only dependency with no
    CPU path, and the only CPU rasteriser in this repo is the parity checker
    (oracle/gs_oracle.py: every Gaussian at every pixel, O(N H W)) — scaling
    it by N H W (round 3: 3.7e-7 frames/s) measures the checker, not a
    rasteriser.  The object keeps the contract's keys with value null."""
    return {'value': None, 'unit': 'frames/s', 'cores': threads,
            'kind': 'none',
            'sample': 'not measured: the reference rasteriser is CUDA-only '
                      '(no CPU path to time) and the dense parity checker '
                      'oracle/gs_oracle.py is O(N H W) per pass — not a '
                      'baseline; compare SplaTAM through roofline.frac and '
                      f'launch_us ({int(n_gaussians)} Gaussians x '
                      f'{int(n_pixels)} pixels)'}


class _NumpyImages:
    """synthetic sequence handing out numpy images (Point-SLAM derives its
    per-pixel radii from a Sobel filter on the host, like the reference)"""

    def __init__(self, data):
        self.data = data

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        d = dict(self.data[i])
        for k in ('rgb', 'depth'):
            if torch.is_tensor(d[k]):
                d[k] = d[k].cpu().numpy()
        return d


def run_pointslam(args, dev, world=1):
    """Point-SLAM frame loop: 40 tracking it x 1500 rays per frame, every 5th
    frame (every frame for the first 20) 300 mapping it x 5000 rays, 5 samples
    per ray, 8-NN feature interpolation from the neural point cloud.  Random-
    initialised decoders (the pretrained checkpoint is not available offline).
    kNN search, geometry path and colour path on the HIP kernels; the
    iterations of a stage run as captured hipGraphs (fixed-shape batches)."""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import (cadence,
                                                       pointslam_config)
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)   # keyframe window sampling: same on every rank
    cam = Camera(**CAM)
    cfg = pointslam_config()
    if args.first_iters is not None:
        cfg.mapping_first_n_iters = args.first_iters
    algo = cfg.setup(camera=cam, device=str(dev))
    algo.use_graphs = not args.no_graphs
    _setup_dist(dev, world)
    data = _NumpyImages(SyntheticRoom(
        CO_BOUND, H=cam.height, W=cam.width, fx=cam.fx, fy=cam.fy, cx=cam.cx,
        cy=cam.cy, n_frames=max(args.warmup + args.steps + 8, 200),
        device=dev))
    cad = cadence['point-slam']
    getattr(data, 'data', data).preload(
        range(args.warmup + args.steps + 1))
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          lazy_start=cad.lazy_start, pose_device=str(dev))
    elapsed = _timed_frames(slam, args, dev, world)
    t_track, t_map = slam.t_track, slam.t_map
    # the dominant launch of this path (the colour path's backward with weight
    # gradients, once per mapping iteration): per-launch timing over a further
    # frame run EAGERLY (events cannot be read back from inside a captured
    # graph)
    from xrdslam_amd.engine import point as epoint
    algo.use_graphs = False
    epoint.PROFILE = {}
    nxt = 1 + args.warmup + args.steps
    for k in range(nxt, nxt + 6):      # until a mapping frame has been seen
        slam.step(k)
        if epoint.PROFILE.get('color_bwd_w'):
            break
    torch.cuda.synchronize()
    prof, epoint.PROFILE = epoint.PROFILE, None
    roofline = None
    key = 'color_bwd_w'
    if prof.get(key):
        # mapping launches (the eager batches differ by a few rays: the
        # batch filter compacts them), each priced with its own point count
        top = max(n for _, _, n in prof[key])
        rows = [(a.elapsed_time(b) * 1e3, n) for a, b, n in prof[key]
                if n > 0.9 * top]
        sel = [t for t, _ in rows]
        us = float(np.mean(sel))
        big = float(np.mean([n for _, n in rows]))
        fwd = [a.elapsed_time(b) * 1e3 for a, b, n in prof.get('color_fwd', [])
               if n > 0.9 * top]
        # per sample point the colour path multiplies
        # 8 x (52x128 + 128x32) (F_theta per neighbour) + 128 x (40 + 128 +
        # 128 + 168 + 128) (trunk) + 5 x 32x128 (feature injections) + 3x128
        # = 182 656 weights: 365 312 flop forward; the backward computes the
        # input AND the weight gradient of each product: 730 624 flop
        flop = 730624.0 * big
        roofline = {
            'bound': 'mfma', 'achieved': flop / (us * 1e-6) / 1e12,
            'peak': MFMA_F32_PEAK / 1e12, 'unit': 'TFLOP/s',
            'frac': flop / (us * 1e-6) / MFMA_F32_PEAK,
            'traffic': pmc_traffic(['point_color_bwd_w_kernel',
                                    'pc_dw_reduce_kernel'],
                                   'r05_pmc_pointslam.json'),
            'traffic_source': f'profiles/{PMC_FILE[0]} (rocprofv3 '
                              '--pmc FETCH_SIZE / WRITE_SIZE passes of this '
                              'workload, FETCH x2 on gfx950; mean over the '
                              "passes' launches)",
            'kernel': 'xrd_point_color_bwd = point_color_bwd_w_kernel (weight '
                      'gradients contracted inside the block) + '
                      'pc_dw_reduce_kernel (colour path backward incl. all '
                      'weight gradients, one call per mapping iteration)',
            'avg_launch_us': us, 'launches': len(sel),
            'points_per_launch': big,
            'algorithmic_flop_per_point': 730624,
            'forward_avg_launch_us': float(np.mean(fwd)) if fwd else None,
            'forward_frac': (365312.0 * big / (float(np.mean(fwd)) * 1e-6) /
                             MFMA_F32_PEAK) if fwd else None,
            'timing_source': 'HIP events around the calls of the (eager) '
                             'frame run right after the timed region'}
    # the steady-state regime RUN, not derived: frames past the lazy start
    # (every 5th frame maps) — the sequence is continued up to frame
    # lazy_start + 1, then 10 frames are timed
    steady = None
    if world == 1 and getattr(args, 'steady_state', True):
        algo.use_graphs = not args.no_graphs
        k = len(algo.get_estimate_c2w_list())
        getattr(data, 'data', data).preload(range(k, cad.lazy_start + 12))
        while k <= cad.lazy_start:
            slam.step(k)
            k += 1
        torch.cuda.synchronize()
        slam.t_track = slam.t_map = 0.0
        t_s = time.perf_counter()
        for j in range(k, k + 10):
            slam.step(j)
        torch.cuda.synchronize()
        t_s = time.perf_counter() - t_s
        steady = {'value': 10 / t_s, 'unit': 'frames/s', 'frames':
                  [k, k + 9], 'mapping_frames': sum(
                      1 for j in range(k, k + 10) if j % cad.map_every == 0),
                  'track_ms_per_frame': slam.t_track / 10 * 1e3,
                  'map_ms_per_frame': slam.t_map / 10 * 1e3,
                  'ate_rmse_m': slam.ate_rmse()}
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = calibrated(pointslam_cpu_baseline(
                min(os.cpu_count() or 1, CPU_THREADS)), 'point-slam')
        except Exception as e:   # the baseline must not take the line down
            cpu = {'value': None, 'unit': 'frames/s', 'kind': 'port',
                   'cores': 0, 'sample': f'failed: {type(e).__name__}: {e}'}
    return {
        'metric': 'tracking+mapping FPS @640x480',
        'value': args.steps / elapsed, 'unit': 'frames/s',
        'ms_per_step': elapsed / args.steps * 1e3, 'dtype': 'f32',
        'config': {
            'workload': 'Point-SLAM 640x480 synthetic RGB-D: 40 tracking it x '
                        '1500 rays + 300 mapping it x 5000 rays (every frame '
                        'during the first 20, then every 5th), 5 samples/ray',
            # which of the two cadences the timed frames fall into, and the
            # rate the other one implies from the same timers
            'regime': ('every timed frame is a mapping frame (frame id <= '
                       f'{cad.lazy_start}: the reference maps every frame '
                       'there) - the conservative rate' if
                       args.warmup + args.steps <= cad.lazy_start else
                       'timed frames straddle the every-frame and the '
                       'every-5th-frame cadence'),
            'steady_state_fps_estimate': (
                1.0 / ((t_track + t_map / cad.map_every) / args.steps)
                if args.warmup + args.steps <= cad.lazy_start else None),
            'steady_state_note': 'frames past the lazy start map every '
                                 f'{cad.map_every}th frame: 1 / (track + map '
                                 f'/ {cad.map_every}) from this run\'s timers',
            # ... and the same regime run: 10 frames past the lazy start
            'steady_state_run': steady,
            'steady_state_fps': steady['value'] if steady else None,
            'track_ms_per_frame': t_track / args.steps * 1e3,
            'map_ms_per_frame': t_map / args.steps * 1e3,
            'ate_rmse_m': slam.ate_rmse(),
            'neural_points': int(algo.model.neural_point_cloud.pts_num())},
        'roofline': add_counters(roofline, 'point_color_bwd_w_kernel',
                                 'r06_pmc_pointslam.json'),
        'cpu_baseline': cpu}


def pointslam_cpu_baseline(threads):
    """the host mirror of the reference's Point-SLAM model on the CPU with the
    exact brute-force neighbour search the golden was made with
    (oracle/faiss_standin.py; tests/test_pointslam_host.py pins this
    configuration against the reference-made golden): frame 0 builds the
    cloud, then ONE tracking iteration (1500 rays) and ONE mapping iteration
    (5000 rays) are timed, forward+backward; converted with the reference
    iteration counts (40 tracking it per frame + 300 mapping it every 5th)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import faiss_standin
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import (cadence,
                                                       pointslam_config)
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.set_num_threads(threads)
    cam = Camera(**CAM)
    cfg = pointslam_config()
    cfg.mapping_first_n_iters = 1
    # a tenth of the rays (bounded sample: the full batch takes minutes per
    # iteration on the host); times are scaled back by the ray count
    shrink = 10
    cfg.tracking_sample //= shrink
    cfg.mapping_sample //= shrink
    algo = cfg.setup(camera=cam, device='cpu')
    algo.model.knn_factory = faiss_standin.TorchKNN
    data = _NumpyImages(SyntheticRoom(
        CO_BOUND, H=cam.height, W=cam.width, fx=cam.fx, fy=cam.fy, cx=cam.cx,
        cy=cam.cy, n_frames=8, device='cpu'))
    cad = cadence['point-slam']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          lazy_start=cad.lazy_start, pose_device='cpu')
    frame = slam.step(0)            # cloud from frame 0, 1 mapping iteration
    t0 = time.perf_counter()
    algo.optimize_update(1, [frame], is_mapping=False)
    t_track = time.perf_counter() - t0
    t0 = time.perf_counter()
    algo.optimize_update(1, [frame], is_mapping=True)
    t_map = time.perf_counter() - t0
    per_frame = shrink * (40 * t_track + 300 * t_map / 5)
    return {'value': 1.0 / per_frame, 'unit': 'frames/s', 'cores': threads,
            'kind': 'port',
            'sample': f'1 tracking iteration ({cfg.tracking_sample} rays x 5 '
                      f'samples) + 1 mapping iteration ({cfg.mapping_sample} '
                      'rays x 5 samples), fwd+bwd, '
                      f'{int(algo.model.neural_point_cloud.pts_num())} neural '
                      'points (cloud seeded from the same reduced batch), '
                      f'exact brute-force 8-NN; scaled x{shrink} to the '
                      'reference ray counts and by its iteration counts; '
                      f'iter seconds track={t_track:.2f} map={t_map:.2f}'}


def _files_ingest(room, n_frames, cam, dev):
    """the synthetic sequence as a Replica-format folder, read back through
    xrdslam_amd.data.datasets (file decode + prefetch + H2D in the loop)"""
    import tempfile

    from PIL import Image

    from xrdslam_amd.data import datasets as fds
    path = tempfile.mkdtemp(prefix='xrd_replica_')
    os.makedirs(os.path.join(path, 'results'))
    with open(os.path.join(path, 'devices.yaml'), 'w') as f:
        f.write(f'cam:\n  H: {cam.height}\n  W: {cam.width}\n  fx: {cam.fx}\n'
                f'  fy: {cam.fy}\n  cx: {cam.cx}\n  cy: {cam.cy}\n'
                '  png_depth_scale: 6553.5\n')
    lines = []
    for k in range(n_frames):
        it = room[k]
        Image.fromarray(np.clip(np.rint(it['rgb'] * 255), 0, 255).astype(
            np.uint8)).save(os.path.join(path, 'results', f'frame{k:06d}.jpg'),
                            quality=95)
        Image.fromarray(np.clip(np.rint(it['depth'] * 6553.5), 0, 65535)
                        .astype(np.uint16)).save(
            os.path.join(path, 'results', f'depth{k:06d}.png'))
        cv = it['c2w'].copy()
        cv[:3, 1] *= -1
        cv[:3, 2] *= -1
        lines.append(' '.join(f'{v:.9e}' for v in cv.reshape(-1)))
    with open(os.path.join(path, 'traj.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return fds.Prefetcher(fds.Replica(path), dev, depth=3)


def _ingest_files_leg(cfg, cam, dev, cad, n_timed=30, n_warm=5):
    """NICE-SLAM frames/s with the sequence read back from Replica-format
    files INSIDE the timed region (decode, pinned staging, one H2D per frame
    on a side stream): a second, short run next to the headline's
    HBM-resident one"""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)
    algo = cfg.setup(camera=cam, device=str(dev))
    algo.use_graphs = True
    n_frames = n_timed + n_warm + 1
    room = SyntheticRoom(BOUND, H=cam.height, W=cam.width, fx=cam.fx,
                         fy=cam.fy, cx=cam.cx, cy=cam.cy,
                         n_frames=NICE_TRAJ_FRAMES, device=dev)
    data = _files_ingest(room, n_frames, cam, dev)
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device=str(dev))
    for k in range(1 + n_warm):
        slam.step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1 + n_warm, n_frames):
        slam.step(k)
    torch.cuda.synchronize()
    return {'value': n_timed / (time.perf_counter() - t0),
            'unit': 'frames/s', 'steps': n_timed, 'warmup': n_warm,
            'note': 'file dataset + prefetching loader inside the timed '
                    'region (PIL decode, pinned staging, H2D on a side '
                    'stream)'}


def nice_side_run(args, dev, seed, host_pose):
    """one more NICE-SLAM run of the headline workload (own model, own seed):
    -> frames/s and ATE.  host_pose=True: the per-frame contract of the
    reference's Tracker — the tracking result is read back to the host as a
    numpy matrix every frame and the next frame's initial pose is predicted in
    numpy from host copies of the last two estimates
    (slam/pipeline/tracker.py:107-112,185-199): one device->host sync a frame,
    i.e. what ds-run would see with this engine under its own Tracker.  (The
    Frame's pose PARAMETERS stay device tensors: the captured iterations read
    them; the reference's CPU shared-memory parameters, frame.py:33-38, would
    be one more 28-byte upload per frame.)  The headline keeps the whole pose
    chain on the device (SequentialSLAM(device_poses=True))."""
    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import (cadence,
                                                       nice_slam_config)
    from xrdslam_amd.slam.pipeline import SequentialSLAM
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    cfg = nice_slam_config(BOUND)
    if args.first_iters is not None:
        cfg.mapping_first_n_iters = args.first_iters
    pre = os.path.join(ROOT, 'xrdslam_amd', 'data', 'pretrained',
                       'nice_decoders_synth.pt')
    if os.path.exists(pre) and not args.random_decoders:
        cfg.model.pretrained_decoders_xrd = pre
    cam = Camera(**CAM)
    algo = cfg.setup(camera=cam, device=str(dev))
    algo.use_graphs = not args.no_graphs
    n_frames = args.warmup + args.steps + 1
    data = SyntheticRoom(BOUND, H=cam.height, W=cam.width, fx=cam.fx,
                         fy=cam.fy, cx=cam.cx, cy=cam.cy,
                         n_frames=max(n_frames, NICE_TRAJ_FRAMES), device=dev)
    data.preload(range(n_frames))
    cad = cadence['nice-slam']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device=str(dev),
                          device_poses=False if host_pose else None)
    for k in range(1 + args.warmup):
        slam.step(k)
    torch.cuda.synchronize()
    gc_was = _gc_pause()
    t0 = time.perf_counter()
    for k in range(1 + args.warmup, n_frames):
        slam.step(k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _gc_resume(gc_was)
    return {'fps': args.steps / elapsed, 'ate_rmse_m': slam.ate_rmse(),
            'ate_rmse_aligned_m': slam.trajectory_stats()[
                'absolute_translational_error.rmse']}


def c1_leg(dev):
    """BASELINE.json configs[0]: Co-SLAM on the 64-frame 320x240 synthetic
    sequence, end to end (every frame tracked, every 5th mapped, first-frame
    initialisation included), next to the trajectory the REFERENCE's own
    CoSLAM produced on the same sequence on the CPU (tests/golden/
    c1_coslam.npz, oracle/make_golden_c1.py: a committed fixture, read for
    the comparison only)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import c1_util
    g = c1_util.fixture('coslam')
    ref_ate, _ = c1_util.ref_stats(g)
    runs = []
    for seed in range(len(ref_ate)):
        est, gt, sec, slam = c1_util.run_engine('coslam', seed, str(dev))
        runs.append((c1_util.ate(est, gt), sec))
    n = int(g['seq/n_frames'])
    ate = [a for a, _ in runs]
    ref_sec = [float(g[f'seconds/{s}']) for s in range(len(ref_ate))]
    return {
        'workload': 'Co-SLAM, 64 frames 320x240 synthetic RGB-D, hash grid + '
                    '2x32 MLPs, reference iteration counts (BASELINE '
                    'configs[0])',
        'fps_end_to_end': n / float(np.mean([s for _, s in runs])),
        'seconds_per_sequence': [s for _, s in runs],
        'ate_rmse_m': ate, 'ate_rmse_mean_m': float(np.mean(ate)),
        'reference_ate_rmse_m': [float(a) for a in ref_ate],
        'reference_ate_rmse_mean_m': float(np.mean(ref_ate)),
        'reference_fps_cpu': n / float(np.mean(ref_sec)),
        'reference': "the reference's CoSLAM + JointEncoding run on the "
                     'build container\'s CPU (8 threads) over the same '
                     'sequence, tiny-cuda-nn served by the oracle encodings',
        'ate_parity': abs(float(np.mean(ate)) - float(np.mean(ref_ate)))
        <= 0.005}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--algo', default='nice-slam',
                    choices=['nice-slam', 'co-slam', 'vox-fusion', 'splaTAM',
                             'point-slam'])
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None,
                    help='timed frames (default 100 for nice-slam/co-slam: '
                         'a 20-frame region is 0.3 s and one sporadic ~75 ms '
                         'runtime stall moves it by 25 %%; 20 for vox-fusion/'
                         'splaTAM, 5 for point-slam)')
    ap.add_argument('--warmup', type=int, default=None,
                    help='untimed frames after frame 0 (default 10 / 5 / 2)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--ingest', default='resident',
                    choices=['resident', 'files'],
                    help='resident: frames in HBM before the timed region; '
                         'files: the sequence is written once as a '
                         'Replica-format folder (jpg/png/traj.txt) and read '
                         'back INSIDE the timed region through the file '
                         'dataset + prefetching loader (decode, pinned '
                         'staging, one H2D per frame on a side stream)')
    ap.add_argument('--no-coslam', action='store_true',
                    help='skip the Co-SLAM leg of the default run')
    ap.add_argument('--no-others', action='store_true',
                    help='default run: skip the Vox-Fusion / SplaTAM / '
                         'Point-SLAM objects and the files-ingest leg')
    ap.add_argument('--random-decoders', action='store_true',
                    help='NICE-SLAM: random-init decoders instead of the '
                         'checkpoint pre-trained on the synthetic room')
    ap.add_argument('--no-side-runs', action='store_true',
                    help='NICE-SLAM: skip the two extra seeds (ATE mean / '
                         'spread) and the host-pose-contract run')
    ap.add_argument('--no-steady-state', dest='steady_state',
                    action='store_false',
                    help='Point-SLAM: skip the run of the every-5th-frame '
                         'regime past the lazy start (21 more frames)')
    ap.add_argument('--no-graphs', action='store_true',
                    help='run every iteration eagerly (no hipGraph capture)')
    ap.add_argument('--first-iters', type=int, default=None,
                    help='override mapping_first_n_iters (untimed set-up)')
    args = ap.parse_args()
    d_steps, d_warm = {'nice-slam': (100, 10), 'co-slam': (100, 10),
                       'point-slam': (5, 2)}.get(args.algo, (20, 5))
    if args.steps is None:
        args.steps = d_steps
    if args.warmup is None:
        args.warmup = d_warm

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('launch with torch.distributed.run for --gpus > 1')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP engine has no CPU '
                         'fallback')
    # functional test of the multi-process path on a box with ONE GPU:
    # XRD_DIST_SAME_GPU=1 XRD_DIST_BACKEND=gloo puts every rank on cuda:0
    same_gpu = os.environ.get('XRD_DIST_SAME_GPU') == '1'
    backend = os.environ.get('XRD_DIST_BACKEND', 'nccl')
    dev = torch.device('cuda:0' if same_gpu else f'cuda:{local_rank}')
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.algo in ('co-slam', 'vox-fusion', 'splaTAM', 'point-slam'):
        res = run_coslam(args, dev, not args.no_cpu_baseline and world == 1,
                         world) \
            if args.algo == 'co-slam' else run_voxfusion(args, dev, world) \
            if args.algo == 'vox-fusion' else run_splatam(args, dev, world) \
            if args.algo == 'splaTAM' else run_pointslam(args, dev, world)
        res.update({'n_gpus': world, 'steps': args.steps,
                    'warmup': args.warmup, 'higher_is_better': True,
                    'scaling': 'strong',
                    'vs_baseline': None, 'data': 'synthetic',
                    'rccl': _rccl_report(world)})
        if rank == 0:
            print(json.dumps(res), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from xrdslam_amd.data.synthetic import SyntheticRoom
    from xrdslam_amd.engine import dist as xdist
    from xrdslam_amd.engine import nice as en
    from xrdslam_amd.slam.common.camera import Camera
    from xrdslam_amd.slam.configs.input_config import (cadence,
                                                       nice_slam_config)
    from xrdslam_amd.slam.pipeline import SequentialSLAM

    torch.manual_seed(0)  # identical on all ranks: tracking stays in lock-step
    np.random.seed(0)
    random.seed(0)   # keyframe window sampling: same on every rank
    cfg = nice_slam_config(BOUND)
    if args.first_iters is not None:
        cfg.mapping_first_n_iters = args.first_iters
    # decoders with an occupancy prior (tools/pretrain_nice_decoders.py: the
    # reference loads pretrained/{coarse,middle_fine}.pt, LFS pointers in its
    # tree); same architecture and work per iteration as random init
    pre = os.path.join(ROOT, 'xrdslam_amd', 'data', 'pretrained',
                       'nice_decoders_synth.pt')
    if os.path.exists(pre) and not args.random_decoders:
        cfg.model.pretrained_decoders_xrd = pre
    cam = Camera(**CAM)
    algo = cfg.setup(camera=cam, device=str(dev))
    algo.use_graphs = not args.no_graphs
    # --no-graphs (counter passes): same kernels as the captured iterations
    algo.eager_fixed_shapes = bool(args.no_graphs)
    if os.environ.get('XRD_PERSISTENT_SHARDED') == '0':   # A/B switch
        algo.persistent_map_graph_sharded = False
    xdist.state.setup(dev, seed=0)
    n_frames = args.warmup + args.steps + 1
    # trajectory sampled at Replica's pace (~5 mm a frame; room0: ~12 m in
    # 2000 frames): NICE-SLAM tracks with 10 iterations at lr 1e-3 and cannot
    # follow the 14 mm a frame of the 200-frame sampling the other (stronger)
    # trackers are run on — measured ATE 15 cm vs 3 cm, same work per frame
    data = SyntheticRoom(BOUND, H=cam.height, W=cam.width, fx=cam.fx,
                         fy=cam.fy, cx=cam.cx, cy=cam.cy,
                         n_frames=max(n_frames, NICE_TRAJ_FRAMES), device=dev)
    if args.ingest == 'files':
        data = _files_ingest(data, n_frames, cam, dev)
    else:
        # inputs resident in HBM before the timed region
        data.preload(range(n_frames))
    cad = cadence['nice-slam']
    slam = SequentialSLAM(algo, data, map_every=cad.map_every,
                          keyframe_every=cad.keyframe_every,
                          pose_device=str(dev))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed set-up: frame 0 (initialisation mapping) + warm-up ------
    # (clocked on the side: SURVEY 8d also asks for the rate INCLUDING the
    # first-frame initialisation; it is reported next to the headline value)
    t_setup = time.perf_counter()
    slam.step(0)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t_setup
    for k in range(1, 1 + args.warmup):
        slam.step(k)
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t_setup - t_init
    # ---- timed region ------------------------------------------------------
    en.PROFILE = {}
    slam.t_track = slam.t_map = 0.0
    barrier()
    gc_was = _gc_pause()
    t0 = time.perf_counter()
    stamps = []
    frame = None
    for k in range(1 + args.warmup, 1 + args.warmup + args.steps):
        frame = slam.step(k)
        stamps.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    _gc_resume(gc_was)
    ate = slam.ate_rmse()
    ate_aligned = slam.trajectory_stats()['absolute_translational_error.rmse']
    if frame is not None:
        # launches inside replayed hipGraphs cannot be event-timed one by one,
        # and with mapping graphs kept across calls the timed region holds
        # few eager launches: right after it, five mapping calls with
        # per-call capture (same shapes, first iteration of every stage
        # segment eager) feed the per-launch statistics (every rank takes
        # part: the sharded mapping calls all-reduce)
        keep = algo.persistent_map_graph
        algo.persistent_map_graph = False
        torch.cuda.synchronize()
        t_calls = time.perf_counter()
        for _ in range(5):
            algo.do_mapping(frame)
        torch.cuda.synchronize()
        t_calls = (time.perf_counter() - t_calls) / 5
        algo.persistent_map_graph = keep
    else:
        t_calls = None
    # per rank: one mapping call (its iterations + the coarse mapper on its
    # side stream + the gradient exchange) and the mapping time per frame of
    # the timed region — every rank takes part in the gather
    per_rank = _per_rank(
        world, mapping_call_ms=None if t_calls is None else t_calls * 1e3,
        mapping_step_ms=None if t_calls is None else
        t_calls * 1e3 / max(1, int(cfg.mapping_n_iters)),
        map_ms_per_frame=slam.t_map / args.steps * 1e3,
        track_ms_per_frame=slam.t_track / args.steps * 1e3)
    if os.environ.get('XRD_BENCH_TRACE'):   # host-side time per frame
        print('frame ms:', ' '.join(
            f'{k + 1 + args.warmup}:{(b - a) * 1e3:.1f}' for k, (a, b) in
            enumerate(zip([t0] + stamps[:-1], stamps))), file=sys.stderr)
    prof, en.PROFILE = en.PROFILE, None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rccl = _rccl_report(world)   # collective: every rank takes part

    if rank == 0:
        # per-kernel launch statistics of the timed region
        stats = []
        for key, evs in prof.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            stats.append((sum(ms), key, len(ms), sum(ms) / len(ms)))
        stats.sort(reverse=True)
        if not stats:
            raise SystemExit('bench.py: no event-timed launches (PROFILE '
                             'hooks not reached)')
        total_ms, key, calls, avg_ms = stats[0]
        kernel, stage, n_rays, need_pose, need_dec, grid_grads = key
        abytes = algorithmic_bytes(kernel, stage, n_rays, grid_grads)
        aflops = algorithmic_flops(kernel, stage, n_rays)
        hbm = abytes / (avg_ms * 1e-3)
        mfma = aflops / (avg_ms * 1e-3)
        # the bound is the one the kernel's arithmetic intensity puts it under
        # (ridge = 157.3 TF / 8 TB/s = 19.7 FLOP/B); the other is reported too
        compute_bound = aflops / abytes > MFMA_F32_PEAK / HBM_PEAK
        roofline = {
            'bound': 'mfma' if compute_bound else 'hbm',
            'achieved': mfma / 1e12 if compute_bound else hbm / 1e9,
            'peak': MFMA_F32_PEAK / 1e12 if compute_bound else HBM_PEAK / 1e9,
            'unit': 'TFLOP/s' if compute_bound else 'GB/s',
            'frac': mfma / MFMA_F32_PEAK if compute_bound else hbm / HBM_PEAK,
            'traffic': (pmc_traffic(nice_group_kernels(kernel, stage,
                                                       need_pose, need_dec))
                        if nice_group_kernels(kernel, stage, need_pose,
                                              need_dec) else None),
            'traffic_source': f'profiles/{PMC_FILE[0]} (rocprofv3 --pmc '
                              'FETCH_SIZE, WRITE_SIZE passes of this workload;'
                              ' bytes per launch group, FETCH x2 on gfx950)',
            'intensity_flop_per_byte': aflops / abytes,
            'other_bound': {
                'bound': 'hbm' if compute_bound else 'mfma',
                'achieved': hbm / 1e9 if compute_bound else mfma / 1e12,
                'unit': 'GB/s' if compute_bound else 'TFLOP/s',
                'frac': hbm / HBM_PEAK if compute_bound else
                mfma / MFMA_F32_PEAK},
            'algorithmic_flops_per_launch': aflops,
            'kernel': f'{kernel}[stage={stage},rays={n_rays},pose_grad='
                      f'{int(need_pose)},decoder_grad={int(need_dec)},'
                      f'grid_grad={int(grid_grads)}]',
            'avg_launch_us': avg_ms * 1e3, 'launches': calls,
            'algorithmic_bytes_per_launch': abytes,
            'share_of_event_timed_time': total_ms / sum(s[0] for s in stats),
            'timing_source': 'HIP events around eager launches on the launch '
                             'stream: first iteration of each stage segment '
                             'of mapping graphs captured in the timed region '
                             '+ 5 per-call-capture mapping calls right after '
                             'it (same shapes); replayed graph nodes are not '
                             'event-timed',
        }
        cpu = torch_gpu = None
        if not args.no_cpu_baseline and world == 1:
            # torch CPU ops on these small tensors stop scaling (and collapse)
            # beyond a few tens of threads: use at most 16 host cores
            cpu = calibrated(cpu_baseline(min(CPU_THREADS, os.cpu_count() or 1)),
                             'nice-slam')
            torch_gpu = cpu_baseline(1, device=str(dev))
        fps = args.steps / elapsed
        out = {
            'metric': 'tracking+mapping FPS @640x480', 'value': fps,
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            # the per-iteration ray batch is fixed by the reference's config
            # and split over the ranks: total work does not grow with N
            'scaling': 'strong', 'rccl': rccl,
            # one entry a rank: mapping_step_ms = one mapping call / its
            # iterations (sharded rays + the all-reduce of rccl.bucket_bytes,
            # rccl.allreduce_ms each); DESIGN 5 holds the predicted curve
            'per_rank': per_rank,
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'NICE-SLAM Replica/office0-shaped 640x480 RGB-D: '
                            '10 tracking it x 200 rays/frame + every 5th '
                            'frame 60 mapping it x 1000 rays + 60 coarse it, '
                            '48 samples/ray, coarse/middle/fine/color grids; '
                            f'synthetic trajectory sampled at {NICE_TRAJ_FRAMES}'
                            ' frames (~5 mm a frame, Replica\'s pace; SURVEY '
                            '8d names 200 frames = 14 mm a frame, which '
                            'NICE-SLAM\'s 10 tracking iterations cannot '
                            'follow: same work per frame, ATE 3-6 cm instead '
                            'of 15 cm)',
                'parallelism': 'replicated tracking, ray-sharded mapping, '
                               'all-reduce of selected-cell gradients'
                               if world > 1 else 'single GPU',
                'track_ms_per_frame': slam.t_track / args.steps * 1e3,
                'map_ms_per_frame': slam.t_map / args.steps * 1e3,
                'render_img_ms': render_img_ms(algo, data,
                                               args.warmup + args.steps, dev),
                # frame 0 = 1500 mapping iterations + graph captures
                'first_frame_init_s': t_init,
                'fps_including_init_and_warmup':
                    (1 + args.warmup + args.steps) /
                    (t_init + t_warm + elapsed),
                'ingest': ('frames resident in HBM before the timed region '
                           '(the per-frame 4.9 MB upload is outside it)'
                           if args.ingest == 'resident' else
                           'Replica-format files decoded + uploaded inside '
                           'the timed region (file dataset, prefetch depth 3, '
                           'one H2D per frame on a side stream)'),
                'ate_rmse_m': ate,
                # after rigid alignment, the number ds-eval reports
                'ate_rmse_aligned_m': ate_aligned,
                'decoders': ('pre-trained on the synthetic room (tools/'
                             'pretrain_nice_decoders.py), middle/fine fixed '
                             'like the reference' if
                             cfg.model.pretrained_decoders_xrd else
                             'random init (seeded)')},
            'roofline': add_counters(
                roofline, (nice_group_kernels(kernel, stage, need_pose,
                                              need_dec) or [None])[0]),
            'cpu_baseline': cpu,
            # the oracle's unfused torch ops on this GPU (a second baseline,
            # not a product path)
            'torch_gpu_baseline': torch_gpu,
        }
        # ATE as mean +- spread over three seeds (a single NICE-SLAM run is
        # chaotic: float atomics + Adam), the leg FAILS above the bound; and
        # the rate under the reference Tracker's host-pose contract
        if world == 1 and not args.no_side_runs:
            try:
                seeds = [{'fps': fps, 'ate_rmse_m': ate,
                          'ate_rmse_aligned_m': ate_aligned}]
                for sd in (1, 2):
                    seeds.append(nice_side_run(args, dev, sd, False))
                a = [s['ate_rmse_m'] for s in seeds]
                out['config']['ate_rmse_3seed_mean_m'] = float(np.mean(a))
                out['config']['ate_rmse_3seed_spread_m'] = \
                    float(np.max(a) - np.min(a))
                out['config']['ate_rmse_3seed_m'] = a
                out['config']['fps_3seed'] = [s['fps'] for s in seeds]
                out['config']['ate_bound_m'] = NICE_ATE_BOUND
                out['config']['ate_within_bound'] = \
                    bool(np.mean(a) <= NICE_ATE_BOUND)
                host = nice_side_run(args, dev, 0, True)
                out['config']['fps_host_pose_contract'] = host['fps']
                out['config']['host_pose_contract'] = dict(
                    host, what='same workload with the tracking result read '
                    'back to the host every frame and the numpy constant-'
                    'velocity prediction of the reference Tracker '
                    '(tracker.py:107-112,185-199): the rate ds-run would see')
            except Exception as e:
                out['config']['side_runs_error'] = \
                    f'{type(e).__name__}: {str(e)[:200]}'
        if world == 1 and not args.no_others:
            try:
                out['c1'] = c1_leg(dev)
                out['config']['c1_coslam_fps'] = out['c1']['fps_end_to_end']
                out['config']['c1_ate_parity'] = out['c1']['ate_parity']
            except Exception as e:
                out['c1'] = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
        # (secondary legs: a failure is recorded, it must not lose the line)
        if world == 1 and args.ingest == 'resident' and not args.no_others:
            try:
                leg = _ingest_files_leg(cfg, cam, dev, cad)
                # a scalar (the driver's record keeps config's scalars) + the
                # leg's details next to it
                out['config']['ingest_files_fps'] = leg['value']
                out['config']['ingest_files_leg'] = leg
            except Exception as e:
                out['config']['ingest_files_fps'] = None
                out['config']['ingest_files_error'] = \
                    f'{type(e).__name__}: {str(e)[:200]}'
        if world == 1 and not args.no_coslam:
            # the second algorithm the north star names, same frame loop
            co_args = argparse.Namespace(**vars(args))
            co_args.first_iters = None
            # (its own region: >= 100 timed frames after >= 10 warm-up frames
            # — a 20-frame region is 0.1 s; the object carries its steps)
            co_args.steps = max(args.steps, 100)
            co_args.warmup = max(args.warmup, 10)
            try:
                co = run_coslam(co_args, dev, not args.no_cpu_baseline)
            except Exception as e:
                co = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
            co.setdefault('steps', co_args.steps)
            co.setdefault('warmup', co_args.warmup)
            out['co_slam'] = co
            # flat copy: the driver's record truncates nested objects
            out['config']['co_slam_fps'] = co.get('value')
        if world == 1 and not args.no_others:
            # BASELINE configs[2..4] in the same (driver-run) line, each on a
            # budget that keeps the whole default run within a few minutes:
            # every object carries its own steps / warm-up, roofline and
            # cpu_baseline
            for name, fn, steps, warm, first in (
                    ('vox_fusion', run_voxfusion, 20, 5, None),
                    # (7 warm-up frames: the first keyframe-overlap selection
                    # — frame 6 — loads ~150 ms worth of library kernels once)
                    ('splatam', run_splatam, 10, 7, None),
                    ('point_slam', run_pointslam, 5, 2, 300)):
                a2 = argparse.Namespace(**vars(args))
                a2.steps, a2.warmup, a2.first_iters = steps, warm, first
                a2.ingest = 'resident'
                t_leg = time.perf_counter()
                try:
                    res = fn(a2, dev)
                except Exception as e:   # one leg must not lose the line
                    res = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
                res.update({'steps': steps, 'warmup': warm,
                            'leg_wall_s': time.perf_counter() - t_leg})
                if first is not None:
                    res['first_iters_override'] = first
                out[name] = res
                out['config'][name + '_fps'] = res.get('value')
                torch.cuda.empty_cache()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
