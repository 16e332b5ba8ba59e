"""summarise rocprofv3 --pmc counter_collection.csv: per kernel mean counter"""
import csv, re, sys, collections
path, counter = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if r.get('Counter_Name') != counter:
        continue
    name = r['Kernel_Name']
    m = re.search(r'nice_(fwd|bwd)_kernel<(\d+), (\d+)(?:, (\w+), (\w+))?>', name)
    short = name[:50]
    if m:
        short = f"nice_{m.group(1)}<dec/stage={m.group(2)},NT={m.group(3)},dp={m.group(4)},dw={m.group(5)}>"
    elif 'nice_dw_kernel' in name:
        short = 'nice_dw_kernel'
    elif 'adam_cells' in name:
        short = 'adam_cells_kernel'
    else:
        continue
    acc[short].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(f'{k:55s} launches={len(v):4d} mean_{counter}={sum(v)/len(v):14.1f}')
