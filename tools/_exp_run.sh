#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for v in base; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o t -- python $GRAFT_REPO_ROOT/tools/coslam_kernel_timing.py --rays 1024 2560 --reps 10 > /tmp/out_$v.txt 2>&1
  echo "== $v"; python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/p_$v/t_kernel_stats.csv 60 | grep -i "coslam\|hash" 
done
