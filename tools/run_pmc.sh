#!/bin/bash
# usage (on the GPU box): tools/run_pmc.sh <tag> [bench.py arguments, e.g. --algo vox-fusion]
# two separate counter passes (FETCH_SIZE, WRITE_SIZE) over a short bench run;
# --pmc is combined with --kernel-trace only.  Only the per-kernel summaries are
# kept (the raw counter CSVs are tens of MB).
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps ${XRD_PMC_STEPS:-5} --warmup ${XRD_PMC_WARMUP:-5} --no-cpu-baseline --no-graphs --no-others --no-side-runs "$@" \
    > $out/pmc_${c}_stdout.txt 2> $out/pmc_${c}_stderr.txt
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $c $out/pmc_$c.json > $out/pmc_$c.txt
  head -20 $out/pmc_$c.txt
done
