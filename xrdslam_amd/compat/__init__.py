"""Import-name shims for the native modules the reference's model code
imports: ``tinycudann`` (Encoding), ``grid`` (svo_intersect,
inverse_cdf_sampling), ``faiss``, ``diff_gaussian_rasterization`` and the TorchScript class
``torch.classes.svo.Octree``.
``install()`` registers the shims under the reference's import names so that
reference-style L1-L3 code runs unmodified on the MI355X engine."""
import importlib
import sys

_NAMES = {
    'tinycudann': 'xrdslam_amd.compat.tinycudann',
    'grid': 'xrdslam_amd.compat.grid',
    'faiss': 'xrdslam_amd.compat.faiss',
    'diff_gaussian_rasterization':
    'xrdslam_amd.compat.diff_gaussian_rasterization',
}


def install(names=None):
    """register the shims; ``svo`` = the TorchScript class library behind
    ``torch.classes.svo.Octree`` (xrdslam_amd/csrc_torch/svo_class.cpp), which
    the reference reaches through ``torch.classes`` and not through an import
    (slam/models/sparse_voxel.py:22-35,307)"""
    for ref_name, ours in _NAMES.items():
        if names is None or ref_name in names:
            sys.modules[ref_name] = importlib.import_module(ours)
    if names is None or 'svo' in names:
        from .. import build_torch_ext
        build_torch_ext.load()
