#!/bin/bash
# usage (on the GPU box): tools/run_pmc.sh <tag> [bench.py arguments, e.g. --algo vox-fusion]
# separate counter passes over a short bench run; --pmc is combined with
# --kernel-trace only.  Passes (MI355X_MICROARCH.md "rocprofv3 PMC slots":
# FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2 — one pass each; the SQ
# block has 8 slots):
#   FETCH_SIZE | WRITE_SIZE                       HBM traffic of a launch
#   MFMA: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
#   LDS:  SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
# XRD_PMC_PASSES selects (default: all four).  Only the per-kernel summaries
# are kept (the raw counter CSVs are tens of MB).
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
passes=${XRD_PMC_PASSES:-FETCH_SIZE WRITE_SIZE MFMA LDS}
for c in $passes; do
  case $c in
    MFMA) ctrs="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" ;;
    LDS) ctrs="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" ;;
    *) ctrs=$c ;;
  esac
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps ${XRD_PMC_STEPS:-5} --warmup ${XRD_PMC_WARMUP:-5} --no-cpu-baseline --no-graphs --no-others --no-side-runs "$@" \
    > $out/pmc_${c}_stdout.txt 2> $out/pmc_${c}_stderr.txt
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  k=$(find /tmp/pmc_$c -name '*kernel_trace.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f "$ctrs" $out/pmc_$c.json $k > $out/pmc_$c.txt
  head -20 $out/pmc_$c.txt
done
