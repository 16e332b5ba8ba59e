"""Import-name shims for the native modules the reference's model code
imports: ``tinycudann`` (Encoding), ``grid`` (svo_intersect,
inverse_cdf_sampling), ``svo`` Octree, ``diff_gaussian_rasterization``.
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
    for ref_name, ours in _NAMES.items():
        if names is None or ref_name in names:
            sys.modules[ref_name] = importlib.import_module(ours)
