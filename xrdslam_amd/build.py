"""Build the gfx950 shared library (libxrdslam_hip.so) in-tree with hipcc.

    python -m xrdslam_amd.build          # incremental
    python -m xrdslam_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but
travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libxrdslam_hip.so')
OBJ = os.path.join(HERE, 'csrc', '_obj')

FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
         '-munsafe-fp-atomics', '-fno-gpu-rdc',
         '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + \
    os.environ.get('XRD_HIPCC_EXTRA', '').split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith('.hip') or f.endswith('.cpp'))


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
               if f.endswith('.h')] + [os.path.join(ROOT, 'include',
                                                    'xrdslam_hip.h')]
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src) + '.o')
        objs.append(obj)
        stale = force or _newer(src, obj) or any(_newer(h, obj)
                                                 for h in headers)
        if stale:
            cmd = [hipcc] + FLAGS + (['-x', 'hip'] if src.endswith('.hip')
                                     else []) + ['-c', src, '-o', obj]
            if verbose:
                print('[xrdslam_amd.build]', ' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f'hipcc failed for {failed}')
    if procs or force or not os.path.exists(LIB):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB
               ] + objs + ['-ldl']
        if verbose:
            print('[xrdslam_amd.build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
