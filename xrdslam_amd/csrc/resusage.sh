#!/bin/bash
# usage: resusage.sh file.hip  -> compact per-kernel register/scratch table
f=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I$(dirname $0)/../../include -I$(dirname $0) -x hip -c $f -o /tmp/_res.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m:
        cur={'name':m.group(1)};rows.append(cur);continue
    for key in ['VGPRs','AGPRs','SGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]','VGPR Spill']:
        m=re.search(r'    '+key+r': (\d+)',line)
        if m and cur is not None: cur[key.split(' ')[0]+('Spill' if 'Spill' in key else '')]=m.group(1)
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'\(.*','',n).replace('xrd::(anonymous namespace)::','')
    print(n, {k:v for k,v in r.items() if k!='name'})
"
