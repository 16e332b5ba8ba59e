"""summarise a rocprofv3 kernel_stats.csv: name, calls, avg us, pct — and the
share of launches / kernel time spent in this repo's own kernels (xrd::) vs
torch's (at::, copies, rocBLAS, ...).  usage: prof_summary.py <csv> [rows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25


def short(name):
    m = re.search(r'(nice_(?:fwd|bwd|bwd_fused|map_fused)_kernel)<([^>]*)>',
                  name)
    if m:
        return f'{m.group(1)}<{m.group(2).replace(" ", "")}>'
    s = name.replace('void xrd::(anonymous namespace)::', '') \
        .replace('xrd::(anonymous namespace)::', '') \
        .replace('void at::native::', 'at::')
    return s[:60]


own = {'calls': 0, 'ns': 0.0}
other = {'calls': 0, 'ns': 0.0}
for r in rows:
    tgt = own if 'xrd::' in r['Name'] else other
    tgt['calls'] += int(r['Calls'])
    tgt['ns'] += float(r['TotalDurationNs'])
tot_c, tot_ns = own['calls'] + other['calls'], own['ns'] + other['ns']
print(f"own kernels (xrd::): {own['calls']} launches "
      f"({100.0 * own['calls'] / max(tot_c, 1):.1f} %), "
      f"{own['ns'] / 1e6:.1f} ms ({100.0 * own['ns'] / max(tot_ns, 1):.1f} % "
      f"of kernel time); torch / library kernels: {other['calls']} launches, "
      f"{other['ns'] / 1e6:.1f} ms "
      f"({100.0 * other['ns'] / max(tot_ns, 1):.1f} %)")
for r in rows[:n]:
    print(f"{short(r['Name']):62s} calls={int(r['Calls']):6d} "
          f"avg_us={float(r['AverageNs']) / 1e3:9.1f} "
          f"total_ms={float(r['TotalDurationNs']) / 1e6:9.2f} "
          f"pct={float(r['Percentage']):5.1f}")
