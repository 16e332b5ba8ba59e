"""merge the per-pass summaries of tools/run_pmc.sh into one json and derive
the busy fractions the counters give:
pmc_merge.py <out.json> <dir with pmc_<PASS>.json> <note>

Derived per kernel (under "_derived"), MI355X = 256 CUs x 4 SIMDs, 8 XCDs:
  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x cycles)
                    (the gfx94x MfmaUtil formula; ROCm 7.2 ships no gfx950
                    derived-counter section), cycles = GRBM_GUI_ACTIVE per XCD
                    (the counter is summed over the 8 XCDs: / 8)
  clock_ghz       = cycles / the launch's duration in the SAME profiled pass
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_lds_frac   = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES-like total
                    (SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY as the denominator
                    of the issue-side buckets of the LDS pass)"""
import json
import os
import sys

out, d, note = sys.argv[1], sys.argv[2], sys.argv[3]
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE', 'MFMA', 'LDS'):
    p = os.path.join(d, f'pmc_{c}.json')
    if os.path.exists(p):
        res.update(json.load(open(p)))
N_XCD, N_CU, N_SIMD = 8, 256, 4
der = {}


def mean(c, k):
    return res.get(c, {}).get(k, {}).get('mean')


for k in res.get('SQ_VALU_MFMA_BUSY_CYCLES', {}):
    e = {}
    busy, gui = mean('SQ_VALU_MFMA_BUSY_CYCLES', k), mean('GRBM_GUI_ACTIVE', k)
    dur = mean('_duration_ns_SQ_VALU_MFMA_BUSY_CYCLES', k)
    if busy is not None and gui:
        cyc = gui / N_XCD
        e['cycles_per_xcd'] = cyc
        e['mfma_busy_frac'] = busy / (N_SIMD * N_CU * cyc)
        if dur:
            e['duration_us_profiled'] = dur / 1e3
            e['clock_ghz'] = cyc / dur
    cu = mean('SQ_BUSY_CU_CYCLES', k)
    if cu and busy is not None:
        # share of the cycles a CU had waves resident during which its MFMA
        # pipes were busy (SQ_BUSY_CU_CYCLES counts per CU, in quad-cycles on
        # some blocks: reported raw beside the ratio)
        e['mfma_busy_over_busy_cu'] = busy / (N_SIMD * cu)
    der[k] = e
for k in res.get('SQ_LDS_IDX_ACTIVE', {}):
    e = der.setdefault(k, {})
    act, conf = mean('SQ_LDS_IDX_ACTIVE', k), mean('SQ_LDS_BANK_CONFLICT', k)
    if act:
        e['lds_conflict_frac'] = (conf or 0.0) / act
    wl, wa, ai = (mean('SQ_WAIT_INST_LDS', k), mean('SQ_WAIT_INST_ANY', k),
                  mean('SQ_ACTIVE_INST_ANY', k))
    if wl is not None and wa is not None and ai:
        e['wait_inst_lds_over_issue'] = wl / (wa + ai)
        e['wait_inst_any_over_issue'] = wa / (wa + ai)
res['_derived'] = der
res['_note'] = note
json.dump(res, open(out, 'w'), indent=1)
print('wrote', out, {k: len(v) for k, v in res.items() if k != '_note'})
