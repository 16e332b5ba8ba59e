"""Which sequence regime makes a trajectory fixture certify TRACKING?  (GPU box.)
Runs the engine (parity-held to the reference loop) over variants of a c1
sequence and prints, per variant: path length, the ATE of a pose frozen at
frame 0, and the engine's ATE over seeds.  The reference fixture
(oracle/make_golden_c1.py, hours of CPU) is then generated in a regime where
engine ATE << frozen ATE.

    python tools/c1_regime.py nice|pointslam|splatam
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import c1_util  # noqa: E402


ROOM = [[-3.0, 3.0], [-4.0, 2.5], [-2.0, 2.5]]
# the sequences of the committed round-5 fixtures (oracle/make_golden_c1.py)
SEQ = {
    'nice': dict(bound=[[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]], H=120, W=160,
                 fx=80.0, fy=80.0, cx=79.5, cy=59.5, n_frames=600, shrink=0.3),
    'pointslam': dict(bound=ROOM, H=120, W=160, fx=80.0, fy=80.0, cx=79.5,
                      cy=59.5, n_frames=200),
    'splatam': dict(bound=ROOM, H=48, W=64, fx=32.0, fy=32.0, cx=31.5,
                    cy=23.5, n_frames=200),
}


class Fx(dict):
    @property
    def files(self):
        return list(self.keys())


def variant(name, seq, cfg, n_run, lazy=None):
    g = Fx()
    g['seq/bound'] = np.array(seq['bound'])
    g['seq/intrinsics'] = np.array([seq['fx'], seq['fy'], seq['cx'], seq['cy'],
                                    seq['W'], seq['H']])
    g['seq/n_frames'] = np.array(seq['n_frames'])
    g['seq/run_frames'] = np.array(n_run)
    if 'shrink' in seq:
        g['seq/shrink'] = np.array(seq['shrink'])
    if name == 'splatam':
        g['seq/cv_poses'] = np.array(1)
    if lazy is not None:
        g['cad/lazy_start'] = np.array(lazy)
    for k, v in cfg.items():
        g[f'cfg/{k}'] = np.array(v)
    return g


def run(name, label, g, seeds=(0, 1, 2)):
    c1_util.fixture = lambda _n: g
    ates, secs = [], []
    for sd in seeds:
        est, gt, sec, _ = c1_util.run_engine(name, sd)
        ates.append(c1_util.ate(est, gt))
        secs.append(sec)
    t = gt[:, :3, 3]
    path = float(np.linalg.norm(np.diff(t, axis=0), axis=1).sum())
    frozen = float(np.sqrt(((t - t[0])**2).sum(1).mean()))
    print(f'{name} {label}: path {path*100:.1f} cm, frozen ATE '
          f'{frozen*100:.2f} cm, engine ATE ' +
          ' '.join(f'{a*100:.2f}' for a in ates) +
          f' (mean {np.mean(ates)*100:.2f} cm = {np.mean(ates)/frozen:.2f} x '
          f'frozen), {np.mean(secs):.1f} s a run', flush=True)


def main():
    which = sys.argv[1]
    base = dict(SEQ[which])
    if which == 'nice':
        red = dict(tracking_Hedge=10, tracking_Wedge=10,
                   mapping_first_n_iters=150, mapping_n_iters=30)
        full = dict(tracking_Hedge=10, tracking_Wedge=10)
        for n_run in (11, 40, 60):
            run('nice', f'{n_run} frames reduced counts',
                variant('nice', base, red, n_run))
        for n_run in (40, 60):
            run('nice', f'{n_run} frames reference counts',
                variant('nice', base, full, n_run))
        fast = dict(base, n_frames=300)
        run('nice', '40 frames, 10 mm a frame, reduced',
            variant('nice', fast, red, 40))
        big = dict(base, H=240, W=320, fx=160.0, fy=160.0, cx=159.5, cy=119.5)
        run('nice', '40 frames 320x240 reduced',
            variant('nice', big, dict(red, tracking_Hedge=20,
                                      tracking_Wedge=20), 40))
        mid = dict(red, mapping_first_n_iters=400, mapping_n_iters=60)
        run('nice', '40 frames, first 400 / 60 a mapping call',
            variant('nice', base, mid, 40))
    elif which == 'pointslam':
        hw = dict(tracking_Hedge=10, tracking_Wedge=10)
        variants = {
            'A track 40x1500, map 100x2500, first 500':
                dict(hw, mapping_n_iters=100, mapping_sample=2500,
                     mapping_first_n_iters=500),
            'B track 40x1500, map 150x5000, first 750':
                dict(hw, mapping_n_iters=150, mapping_first_n_iters=750),
            'C track 40x1500, map 60x1000, first 300':
                dict(hw, mapping_n_iters=60, mapping_sample=1000,
                     mapping_first_n_iters=300),
            'D track 40x1000, map 100x2000, first 500':
                dict(hw, tracking_sample=1000, mapping_n_iters=100,
                     mapping_sample=2000, mapping_first_n_iters=500),
            'E track 40x600, map 100x2500, first 500':
                dict(hw, tracking_sample=600, mapping_n_iters=100,
                     mapping_sample=2500, mapping_first_n_iters=500),
            'F reference counts': dict(hw),
        }
        only = sys.argv[2] if len(sys.argv) > 2 else ''
        for label, cfg in variants.items():
            if only and label[0] not in only:
                continue
            run('pointslam', f'20 frames {label}',
                variant('pointslam', base, cfg, 20, lazy=20))
    elif which == 'splatam':
        for (W, H, f) in ((64, 48, 32.0), (160, 120, 80.0)):
            sq = dict(base, W=W, H=H, fx=f, fy=f, cx=W / 2 - 0.5,
                      cy=H / 2 - 0.5)
            for n_run in (4, 10, 20):
                run('splatam', f'{W}x{H} {n_run} frames',
                    variant('splatam', sq, {}, n_run))
        sq = dict(base, W=160, H=120, fx=80.0, fy=80.0, cx=79.5, cy=59.5,
                  n_frames=400)
        run('splatam', '160x120 20 frames half pace',
            variant('splatam', sq, {}, 20))


if __name__ == '__main__':
    t0 = time.time()
    main()
    print(f'{time.time() - t0:.0f} s')
