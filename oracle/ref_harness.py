"""TEST INFRASTRUCTURE ONLY — never imported by the product path.

Stub-import harness that makes the *reference's own* pure-PyTorch hot path
(`/root/reference/slam/...`) importable on a CPU-only box, so that
  * the torch restatements under ``oracle/`` can be validated against it, and
  * golden vectors under ``tests/golden/`` can be generated from it
    (``oracle/make_golden.py``).

`/root/reference` only exists in the build container, never on the GPU box:
nothing in ``tests -m gpu``, ``smoke()`` or ``bench.py`` may call this module.

Recipe: put ``MagicMock`` modules into ``sys.modules`` for the
third-party packages that are not installed, add the reference root to
``sys.path``, and neutralise the hard-coded ``.cuda()`` calls.
"""
import os
import sys
import types
from unittest import mock

REF_ROOT = os.environ.get('XRDSLAM_REFERENCE', '/root/reference')

_STUBS = [
    'cv2', 'open3d', 'trimesh', 'skimage', 'skimage.measure', 'skimage.color',
    'skimage.filters', 'transforms3d', 'diff_gaussian_rasterization',
    'pytorch_msssim', 'torchmetrics', 'torchmetrics.image',
    'torchmetrics.image.lpip', 'tinycudann', 'pytorch3d',
    'pytorch3d.transforms', 'faiss', 'grid', 'tyro', 'mathutils',
    'evaluate_3d_reconstruction', 'matplotlib', 'matplotlib.pyplot',
]


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, 'slam'))


def install(cuda_noop: bool = True):
    """Make ``import slam.*`` resolve to the reference tree."""
    if not available():
        raise RuntimeError(f'reference tree not found at {REF_ROOT}')
    for name in _STUBS:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = mock.MagicMock(name=name)
                m.__path__ = []  # behave like a package
                sys.modules[name] = m
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if cuda_noop:
        import torch
        if not torch.cuda.is_available():
            torch.Tensor.cuda = lambda self, *a, **k: self
            torch.nn.Module.cuda = lambda self, *a, **k: self
    return types.SimpleNamespace(root=REF_ROOT)
