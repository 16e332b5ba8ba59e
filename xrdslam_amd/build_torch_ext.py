"""Build the TorchScript seam library (csrc_torch/svo_class.cpp ->
_torch_ext/xrd_svo_class.so): ``torch.classes.svo.Octree`` on top of the C-ABI
octree.  Host C++ only (no HIP); needs the torch headers, so it is built with
torch.utils.cpp_extension, in-tree (the .so is git-ignored and travels with
the gpurun snapshot).

    python -m xrdslam_amd.build_torch_ext
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, '_torch_ext')
LIB = os.path.join(OUT, 'xrd_svo_class.so')
SRC = os.path.join(HERE, 'csrc_torch', 'svo_class.cpp')


def build(verbose=False):
    from torch.utils import cpp_extension

    from . import build as core
    core.build(verbose=verbose)  # libxrdslam_hip.so must exist to link
    if os.path.exists(LIB) and os.path.getmtime(LIB) > os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    cpp_extension.load(
        name='xrd_svo_class', sources=[SRC],
        extra_include_paths=[os.path.join(ROOT, 'include')],
        extra_cflags=['-O2'],
        # rpath = the package directory next to _torch_ext/ ('$$' for ninja,
        # the backslash for the shell that runs the link line)
        extra_ldflags=[f'-L{HERE}', '-lxrdslam_hip',
                       '-Wl,-rpath,\\$$ORIGIN/..'],
        build_directory=OUT, is_python_module=False, verbose=verbose)
    return LIB


def load():
    """register torch.classes.svo (idempotent)"""
    import torch
    if not os.path.exists(LIB):
        build()
    torch.classes.load_library(LIB)
    return torch.classes.svo.Octree


if __name__ == '__main__':
    print(build(verbose=True))
