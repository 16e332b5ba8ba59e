"""TEST INFRASTRUCTURE ONLY.  Builds the C parts of the oracle with gcc into
oracle/_build/ (git-ignored, shipped to the GPU box like the product .so) and,
when the reference tree is present, the compiled reference under oracle/_ref/.
Called by __graft_entry__.build(); building the checker is not using it."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_build')


def build_c():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, 'svo_oracle.c')
    lib = os.path.join(OUT, 'libsvo_oracle.so')
    if not os.path.exists(lib) or os.path.getmtime(src) > os.path.getmtime(lib):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fPIC',
                               '-shared', '-o', lib, src, '-lm'])
    return lib


if __name__ == '__main__':
    print(build_c())
    sys.path.insert(0, HERE)
    import build_ref_grid
    import build_ref_octree
    if build_ref_grid.available():
        print(build_ref_grid.build())
    if build_ref_octree.available() and '--with-ref' in sys.argv:
        build_ref_octree.load()
