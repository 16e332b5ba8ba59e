"""TEST INFRASTRUCTURE ONLY.  ``tinycudann.Encoding`` stand-in evaluated by the
torch oracle (oracle/tcnn_oracle.py; "parity unpinned": tiny-cuda-nn v1.7 is an
unvendored dependency, see that file's header).  Used (a) to execute the
reference's Co-SLAM model on the CPU when generating
tests/golden/coslam_render.npz and (b) by the CPU tests to run the host mirror
of that model without a GPU.  Never imported by the product package."""
import ctypes as C
import types

import numpy as np
import torch
import torch.nn as nn

import tcnn_oracle as to


def level_table(cfg):
    """level table of the product library's HOST function, so that oracle and
    kernels evaluate the same resolutions (tests/test_tcnn_oracle.py)"""
    from xrdslam_amd import _lib
    L = int(cfg['n_levels'])
    sc, rs = np.zeros(L, np.float32), np.zeros(L, np.uint32)
    sz, of = np.zeros(L, np.uint32), np.zeros(L, np.uint32)
    tot = C.c_uint32(0)
    dense = 1 if cfg.get('otype') == 'Grid' and cfg.get('type') == 'Dense' \
        else 0
    rc = _lib.lib().xrd_hashgrid_levels(
        L, int(cfg['base_resolution']), float(cfg['per_level_scale']),
        int(cfg.get('log2_hashmap_size', 19)), dense, sc.ctypes.data,
        rs.ctypes.data, sz.ctypes.data, of.ctypes.data, C.byref(tot))
    assert rc == 0
    return [(float(a), int(b), int(c), int(d))
            for a, b, c, d in zip(sc, rs, sz, of)], int(tot.value)


class OracleEncoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, dtype=torch.float):
        super().__init__()
        self.cfg = dict(encoding_config)
        if self.cfg['otype'] in ('HashGrid', 'Grid'):
            self.levels, total = level_table(self.cfg)
            self.n_output_dims = 2 * len(self.levels)
            self.params = nn.Parameter((torch.rand(total * 2) * 2 - 1) * 1e-4)
        else:
            self.n_bins = int(self.cfg['n_bins'])
            self.n_output_dims = n_input_dims * self.n_bins
            self.params = nn.Parameter(torch.zeros(0))

    def forward(self, x):
        if self.cfg['otype'] in ('HashGrid', 'Grid'):
            return to.hashgrid_forward(x.float(), self.params, self.levels)
        return to.oneblob_forward(x.float(), self.n_bins)


def module():
    m = types.ModuleType('tinycudann')
    m.Encoding = OracleEncoding
    return m
