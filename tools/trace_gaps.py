"""GPU idle time of a rocprofv3 --kernel-trace run: union of the kernel
intervals over the LAST ``frac`` of the trace (the timed region of bench.py),
and which kernels the queue waited in front of.  CAVEAT: rocprofv3's kernel
tracing serialises dispatches (concurrent streams show 0 % overlap) and adds
~10 us in front of every graph launch: read the idle time as an upper bound.
usage: trace_gaps.py <kernel_trace.csv> [frac=0.5]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                 r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '0'))))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - frac)
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy, cur_end, gaps = 0, rows[0][0], collections.Counter()
gap_n = collections.Counter()
sum_k = 0
for s, e, name, _ in rows:
    sum_k += e - s
    if s > cur_end:
        short = name.replace('void ', '').replace(
            'xrd::(anonymous namespace)::', '').split('(')[0][:50]
        gaps[short] += s - cur_end
        gap_n[short] += 1
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
print(f'kernels {len(rows)}, span {span/1e6:.2f} ms, union busy '
      f'{busy/1e6:.2f} ms ({100.0*busy/span:.1f} %), idle '
      f'{(span-busy)/1e6:.2f} ms, sum of kernel durations {sum_k/1e6:.2f} ms '
      f'(overlap of concurrent streams {100.0*(sum_k-busy)/max(sum_k,1):.1f} %)')
print('idle time by the kernel that ended the gap (ms, gaps, mean us):')
for k, v in gaps.most_common(14):
    print(f'  {v/1e6:8.3f} {gap_n[k]:6d} {v/gap_n[k]/1e3:8.1f}  {k}')
