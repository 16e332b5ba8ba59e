#!/bin/bash
# usage: tools/run_profile.sh <tag> <bench args...>   (on the GPU box)
# kernel-trace + stats of bench.py; only the small summary CSVs are kept.
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python $GRAFT_REPO_ROOT/${XRD_PROF_SCRIPT:-bench.py} "$@" > $out/bench_stdout.txt 2> $out/bench_stderr.txt
cp /tmp/prof_$tag/bench_kernel_stats.csv $out/ 2>/dev/null
python $GRAFT_REPO_ROOT/tools/prof_summary.py $out/bench_kernel_stats.csv 40 > $out/kernel_summary.txt
tail -1 $out/bench_stdout.txt
head -14 $out/kernel_summary.txt
