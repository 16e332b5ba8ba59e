"""summarise a rocprofv3 kernel_stats.csv: name, calls, avg us, pct"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    name = r['Name']
    m = re.search(r'nice_(fwd|bwd)_kernel<(\d+), (\d+)(?:, (\w+), (\w+))?>', name)
    short = name[:60]
    m2 = re.search(r'nice_bwd_fused_kernel<(\d+), (\d+), (\w+), (\w+)>', name)
    if m2:
        short = (f"nice_bwd_fused<stage={m2.group(1)},NT={m2.group(2)},"
                 f"dp={m2.group(3)},dw={m2.group(4)}>")
    elif m:
        short = f"nice_{m.group(1)}<stage={m.group(2)},NT={m.group(3)},dp={m.group(4)},dw={m.group(5)}>"
    short = short.replace('void xrd::(anonymous namespace)::', '').replace('void at::native::', 'at::')
    print(f"{short:62s} calls={int(r['Calls']):6d} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} pct={float(r['Percentage']):5.1f}")
