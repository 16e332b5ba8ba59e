"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/xrdslam_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

from xrdslam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'xrdslam_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(xrd_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in the header but not exported'


def test_ctypes_table_matches_header():
    assert set(_lib.declared_symbols()) == set(header_symbols())


def test_abi_version_and_lengths():
    lib = _lib.lib()
    assert lib.xrd_abi_version() >= 1
    assert [lib.xrd_nice_flat_len(k) for k in range(4)] == \
        [6337, 15800, 20920, 15899]  # SURVEY.md §8a A7 parameter counts
    assert lib.xrd_nice_flat_len(9) == -1


def test_bad_arguments_are_reported_not_fatal():
    lib = _lib.lib()
    assert lib.xrd_nice_pack_index(0, None) == 1  # XRD_ERR_ARG
    assert lib.xrd_adam_cells(None, None, None, None, None, 0, 32, 0.1, 0.9,
                              0.999, 1e-8, 1, 0, None) == 1
