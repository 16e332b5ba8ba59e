"""Parity metrics shared by the GPU and CPU tests.

BASELINE.json's bar is 1e-4 relative (fp32).  Two readings are checked side
by side:

* ``rel_max``  max|a-b| / max|b| — the max-norm reading (what round 1 used);
* ``elementwise`` |a-b| <= RTOL*|b| + ATOL_FRAC*max|b| for EVERY element —
  entries of small magnitude (a grid cell touched by one sample, a rotation
  component of a pose gradient) are held to 1e-4 of their own size plus an
  absolute floor.

The floor exists because every compared quantity is an fp32 sum whose terms
partly cancel: the rounding error of such a sum scales with the sum of the
|terms| (~ the tensor's large entries), not with the result.  ATOL_FRAC = 1e-5
puts the floor at a tenth of the max-norm bar, i.e. an entry smaller than
1e-5 * max|b| is checked absolutely, everything above it relatively.

``report`` appends one line per comparison to $XRD_PARITY_REPORT (if set), so
that a GPU run leaves the measured margins behind
(gpurun_out/ -> profiles/rNN_parity_margins.txt)."""
import os

import numpy as np

RTOL = 1e-4
ATOL_FRAC = 1e-5


def _np(a):
    if hasattr(a, 'detach'):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def rel_max(a, b):
    a, b = _np(a), _np(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def elementwise_excess(a, b, rtol=RTOL, atol_frac=ATOL_FRAC):
    """max over elements of |a-b| / (rtol*|b| + atol_frac*max|b|); <= 1 passes"""
    a, b = _np(a), _np(b)
    scale = max(np.abs(b).max(), 1e-30)
    return float((np.abs(a - b) / (rtol * np.abs(b) + atol_frac * scale))
                 .max())


def report(name, a, b):
    path = os.environ.get('XRD_PARITY_REPORT')
    rm, ex = rel_max(a, b), elementwise_excess(a, b)
    if path:
        with open(path, 'a') as f:
            f.write(f'{name}\trel_max={rm:.3e}\telementwise={ex:.3f}\n')
    return rm, ex


def check(name, a, b, tol=RTOL):
    """both readings; returns an error string or None"""
    a, b = _np(a), _np(b)
    if a.shape != b.shape:
        return f'{name}: shape {a.shape} vs {b.shape}'
    rm, ex = report(name, a, b)
    if not rm < tol:
        return f'{name}: max-norm relative error {rm:.3e} >= {tol:g}'
    if not ex <= 1.0:
        return (f'{name}: element-wise |a-b| <= {RTOL:g}|b| + '
                f'{ATOL_FRAC:g}max|b| violated by x{ex:.2f}')
    return None


def assert_all(pairs, tol=RTOL):
    """pairs: iterable of (name, got, want)"""
    bad = [m for m in (check(n, a, b, tol) for n, a, b in pairs) if m]
    assert not bad, '\n'.join(bad)


def row_outliers(a, b, tol=RTOL):
    """per-row max deviation relative to max|b|; returns (fraction of rows
    above tol, largest row deviation).  For per-ray gradients: a ReLU kink or
    a trilinear cell border crossed by ONE sample under a last-bit rounding
    difference changes that ray's gradient discontinuously (by ~1e-3 of the
    largest gradient) while every other ray agrees to 1e-6 — two correct f32
    evaluations (torch CPU / torch CUDA / these kernels) show such rows
    against each other and against an f64 evaluation."""
    a, b = _np(a), _np(b)
    scale = max(np.abs(b).max(), 1e-30)
    dev = np.abs(a - b).reshape(a.shape[0], -1).max(1) / scale
    return float((dev > tol).mean()), float(dev.max())


def row_stats(a, b):
    """per-row max deviation relative to max|b| -> (mean, median, p99, max,
    p90)"""
    a, b = _np(a), _np(b)
    scale = max(np.abs(b).max(), 1e-30)
    dev = np.abs(a - b).reshape(a.shape[0], -1).max(1) / scale
    return (float(dev.mean()), float(np.median(dev)),
            float(np.percentile(dev, 99)), float(dev.max()),
            float(np.percentile(dev, 90)))
