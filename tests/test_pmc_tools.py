"""CPU: the counter tooling behind the bench line's `mfma_busy_frac` (tools/
pmc_summary.py, tools/pmc_merge.py) on a synthetic rocprofv3 counter CSV: the
per-kernel means, the duration join and the derived fractions
(mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x
GRBM_GUI_ACTIVE per XCD), LDS conflict and LDS-issue-stall shares)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = ('void xrd::(anonymous namespace)::nice_map_fused_kernel<3, 3, '
          'false, true, false>(xrd_nice_scene, int)')
SHORT = 'nice_map_fused<stage=3,NT=3,dp=false,dw=true>'


def _csv(path, rows):
    with open(path, 'w') as f:
        f.write('"Kernel_Name","Counter_Name","Counter_Value"\n')
        for k, c, v in rows:
            f.write(f'"{k}","{c}",{v}\n')


def test_summary_merge_and_derived_fractions(tmp_path):
    d = tmp_path / 'pmc'
    d.mkdir()
    cyc = 400000.0                       # cycles per XCD of a launch
    passes = {
        'MFMA': {'SQ_VALU_MFMA_BUSY_CYCLES': 0.25 * 4 * 256 * cyc,
                 'SQ_BUSY_CU_CYCLES': 256 * cyc, 'SQ_WAVE_CYCLES': 1e8,
                 'GRBM_GUI_ACTIVE': 8 * cyc},
        'LDS': {'SQ_LDS_BANK_CONFLICT': 2e5, 'SQ_LDS_IDX_ACTIVE': 1e7,
                'SQ_WAIT_INST_LDS': 1e6, 'SQ_WAIT_INST_ANY': 6e7,
                'SQ_ACTIVE_INST_ANY': 4e7},
        'FETCH_SIZE': {'FETCH_SIZE': 1000.0},
        'WRITE_SIZE': {'WRITE_SIZE': 500.0}}
    for name, ctrs in passes.items():
        csv = tmp_path / f'{name}.csv'
        rows = []
        for c, v in ctrs.items():
            rows += [(KERNEL, c, v * 0.5), (KERNEL, c, v * 1.5),
                     ('some_torch_kernel', c, 7.0)]
        _csv(csv, rows)
        trace = tmp_path / f'{name}_trace.csv'
        with open(trace, 'w') as f:
            f.write('"Kernel_Name","Start_Timestamp","End_Timestamp"\n')
            f.write(f'"{KERNEL}",1000,201000\n"{KERNEL}",5000,205000\n')
        subprocess.run([sys.executable,
                        os.path.join(ROOT, 'tools', 'pmc_summary.py'),
                        str(csv), ' '.join(ctrs), str(d / f'pmc_{name}.json'),
                        str(trace)], check=True, capture_output=True)
    out = tmp_path / 'merged.json'
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_merge.py'),
                    str(out), str(d), 'test'], check=True,
                   capture_output=True)
    r = json.load(open(out))
    assert list(r['FETCH_SIZE']) == [SHORT]          # torch kernels dropped
    assert r['FETCH_SIZE'][SHORT] == {'launches': 2, 'mean': 1000.0,
                                      'max': 1500.0}
    der = r['_derived'][SHORT]
    assert abs(der['mfma_busy_frac'] - 0.25) < 1e-12
    assert abs(der['cycles_per_xcd'] - cyc) < 1e-6
    assert abs(der['clock_ghz'] - cyc / 200000.0) < 1e-9     # 200 us launches
    assert abs(der['lds_conflict_frac'] - 0.02) < 1e-12
    assert abs(der['wait_inst_lds_over_issue'] - 0.01) < 1e-12
    assert abs(der['wait_inst_any_over_issue'] - 0.6) < 1e-12
