"""TEST INFRASTRUCTURE ONLY.  Compiles the REFERENCE's sparse_octree sources
where they lie under /root/reference (third_party/sparse_octree/src/*.cpp) into
oracle/_ref/ with torch.utils.cpp_extension (no copy of the sources; not the
reference's own setup.py).  Eigen is not installed: only `decode()`'s return
type needs it (include/utils.h:3,98-104), so a 10-line stand-in header is
generated under oracle/_ref/shim/.  The resulting TorchScript class
``torch.classes.svo.Octree`` is the golden for "voxel indices bit-exact"."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('XRDSLAM_REFERENCE', '/root/reference')
SRC = os.path.join(REF, 'third_party', 'sparse_octree')
OUT = os.path.join(HERE, '_ref', 'sparse_octree')

EIGEN_SHIM = '''#pragma once
// stand-in for <eigen3/Eigen/Dense>: only Vector3i as decode()'s return type
namespace Eigen {
struct Vector3i {
  int v[3];
  Vector3i(unsigned long long a, unsigned long long b, unsigned long long c)
      : v{(int)a, (int)b, (int)c} {}
  int operator[](int i) const { return v[i]; }
};
}  // namespace Eigen
'''


def available():
    return os.path.isdir(os.path.join(SRC, 'src'))


def load():
    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT, exist_ok=True)
    shim = os.path.join(OUT, 'shim', 'eigen3', 'Eigen')
    os.makedirs(shim, exist_ok=True)
    with open(os.path.join(shim, 'Dense'), 'w') as f:
        f.write(EIGEN_SHIM)
    cpp_extension.load(
        name='svo_ref',
        sources=[os.path.join(SRC, 'src', 'octree.cpp'),
                 os.path.join(SRC, 'src', 'bindings.cpp')],
        extra_include_paths=[os.path.join(SRC, 'include'),
                             os.path.join(OUT, 'shim')],
        extra_cflags=['-O2', '-w'], build_directory=OUT,
        is_python_module=False, verbose=False)
    return torch.classes.svo.Octree


if __name__ == '__main__':
    if not available():
        print('reference tree not present: nothing to build')
        sys.exit(0)
    cls = load()
    print('built', OUT, cls)
