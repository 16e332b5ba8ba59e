"""Executable statement (numpy, one ray at a time) of the WAVE-COOPERATIVE
formulation that ``inverse_cdf_kernel`` in xrdslam_amd/csrc/svo.hip uses, so
that the re-formulation can be checked against the compiled reference kernel
on the CPU (tests/test_svo_oracle.py) before it is checked on the GPU.

The reference (third_party/sparse_voxels/src/sample_gpu.cu:133-239) walks the
steps of a ray serially, carrying (bin, z_low).  Here every step is
independent:

  cum[b]   serial float prefix sum of probs (same order of additions)
  bin(c)   = running max over c' <= c of (first b with not cdf(c') > cum[b])
  done     = first step whose bin reaches nb (the number of valid bins)
  step c (not done) writes the in-bin sample at slot c + bin(c); the bin
  boundaries crossed between step c-1 and step c are written by step c at
  slots c + b, b in [bin(c-1), bin(c)); z_low of a sample is the previous
  step's z when that step lies in the same bin, else the bin's min depth.

The reference's trailing "remaining bins" loop (with its quirks: ``~done`` is
always true, ``pts_idx[curr_bin]`` lacks the ray offset, the
``num_rays > H + curr_bin`` guard) stays serial: it runs at most max_hits
iterations on one lane."""
import numpy as np

f32 = np.float32


def _fma(a, b, c):
    """round(a*b + c) once, like nvcc's default contraction of
    ``min + u * (max - min)`` (float64 holds the f32 product exactly)"""
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def sample_ray(j, num_rays, max_hits, max_steps, fixed, PI, MN, MX, UN, PR,
               ST, SI, SD, SS, pi_len):
    """PI/MN/MX/PR: flat arrays of ONE batch; writes row j of SI/SD/SS"""
    H, K = j * max_hits, j * max_steps

    def pi(i):  # one-past-the-buffer reads are defined as -1
        return int(PI[i]) if i < pi_len else -1

    # the serial loop tests "the next bin is invalid" AFTER leaving a bin, so
    # bin 0 is entered unconditionally
    nbv = 1
    while nbv < max_hits and pi(H + nbv) != -1:
        nbv += 1
    cum = np.zeros(max_hits, f32)
    acc = f32(0)
    for b in range(min(nbv, max_hits)):
        acc = f32(acc + PR[H + b])
        cum[b] = acc
    steps = ST[j]
    step_size = f32(np.float64(1.0) / np.float64(steps))
    if fixed > 0.0:
        step_size = f32(fixed)
    total = int(np.ceil(np.float64(steps)))

    def emit(s, idx, dist, dep):
        if s < max_steps:
            SI[K + s], SS[K + s], SD[K + s] = idx, dist, dep

    run_bin, prev_bin, prev_z = 0, 0, MN[H]
    done_at = None
    for c in range(total):  # independent per lane; (prev_*) come by shuffle
        cdf = f32(f32(f32(c) + UN[K + c]) * step_size)
        cnt = 0  # first bin whose upper cdf is not exceeded
        while cnt < nbv and cdf > cum[cnt]:
            cnt += 1
        run_bin = max(run_bin, cnt)
        bin_c = run_bin
        # boundaries crossed since the previous step
        for b in range(prev_bin, min(bin_c, nbv)):
            zl = prev_z if b == prev_bin else MN[H + b]
            emit(c + b, pi(H + b), f32(MX[H + b] - zl),
                 f32(np.float64(f32(MX[H + b] + zl)) * 0.5))
        if bin_c >= nbv:
            done_at = c
            zl_last = prev_z if nbv - 1 == prev_bin else MN[H + nbv - 1]
            break
        cmin = cum[bin_c - 1] if bin_c > 0 else f32(0)
        u = f32(f32(cdf - cmin) / f32(cum[bin_c] - cmin))
        z = _fma(u, f32(MX[H + bin_c] - MN[H + bin_c]), MN[H + bin_c])
        zl = prev_z if bin_c == prev_bin else MN[H + bin_c]
        emit(c + bin_c, pi(H + bin_c), f32(z - zl),
             f32(np.float64(f32(z + zl)) * 0.5))
        prev_bin, prev_z = bin_c, z
    # state handed to the serial tail
    if done_at is not None:
        curr_bin, s = nbv, done_at + nbv
        z_low, cur_max = zl_last, MX[H + nbv - 1]
    else:
        curr_bin, s = prev_bin, total + prev_bin
        z_low, cur_max = prev_z, MX[H + prev_bin]
    while z_low < cur_max and num_rays > H + curr_bin:
        emit(s, pi(H + curr_bin), f32(cur_max - z_low),
             f32(np.float64(f32(cur_max + z_low)) * 0.5))
        curr_bin += 1
        s += 1
        if curr_bin >= max_hits or pi(curr_bin) == -1:
            break
        cur_max = MX[H + curr_bin]
        z_low = MN[H + curr_bin]


def inverse_cdf_parallel_model(pts_idx, mn, mx, noise, probs, steps, fixed):
    G, R, Pn = mn.shape
    S = noise.shape[-1]
    sidx = -np.ones((G, R, S), np.int32)
    sdep = np.zeros((G, R, S), np.float32)
    sdis = np.zeros((G, R, S), np.float32)
    flat = pts_idx.reshape(-1)
    for g in range(G):
        # like the kernel's per-batch pointers: reads past the batch land in
        # the next batch (or are -1 past the whole buffer)
        PI = flat[g * R * Pn:]
        for j in range(R):
            sample_ray(j, R, Pn, S, fixed, PI, mn[g].reshape(-1),
                       mx[g].reshape(-1), noise[g].reshape(-1),
                       probs[g].reshape(-1), steps[g], sidx[g].reshape(-1),
                       sdep[g].reshape(-1), sdis[g].reshape(-1), PI.shape[0])
    return sidx, sdep, sdis
