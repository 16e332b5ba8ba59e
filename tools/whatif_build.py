"""What-if builds: a copy of the library with text edits applied to ONE source
file (phases switched off, constants changed), for timing experiments that
must never reach the product library.

    python tools/whatif_build.py NAME csrc_file 'old text=>new text' ['old=>new' ...]

-> tools/scratch/libxrdslam_hip_NAME.so (git-ignored, travels to the GPU box);
tools that take ``--lib`` load it instead of the product library."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tools', 'scratch')


def main():
    name, fname, edits = sys.argv[1], sys.argv[2], sys.argv[3:]
    from xrdslam_amd import build as b
    b.build(verbose=False)
    path = os.path.join(b.CSRC, fname)
    src = open(path).read()
    for e in edits:
        old, new = e.split('=>', 1)
        assert src.count(old) >= 1, f'not found: {old!r}'
        src = src.replace(old, new)
    os.makedirs(OUT, exist_ok=True)
    tmp = os.path.join(OUT, f'whatif_{name}_{fname}')
    open(tmp, 'w').write(src)
    obj = tmp + '.o'
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + b.FLAGS +
                          ['-x', 'hip', '-c', tmp, '-o', obj])
    objs = [os.path.join(b.OBJ, os.path.basename(s) + '.o')
            for s in b.sources() if os.path.basename(s) != fname]
    lib = os.path.join(OUT, f'libxrdslam_hip_{name}.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950',
                           '-shared', '-fPIC', '-o', lib] + objs +
                          [obj, '-ldl'])
    print('built', lib)


if __name__ == '__main__':
    main()
