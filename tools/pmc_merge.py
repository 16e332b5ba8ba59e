"""merge the per-counter summaries of tools/run_pmc.sh into one json:
pmc_merge.py <out.json> <dir with pmc_FETCH_SIZE.json / pmc_WRITE_SIZE.json> <note>"""
import json
import os
import sys

out, d, note = sys.argv[1], sys.argv[2], sys.argv[3]
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    res.update(json.load(open(os.path.join(d, f'pmc_{c}.json'))))
res['_note'] = note
json.dump(res, open(out, 'w'), indent=1)
print('wrote', out, {k: len(v) for k, v in res.items() if k != '_note'})
