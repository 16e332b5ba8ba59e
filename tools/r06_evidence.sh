#!/bin/bash
# round-6 evidence on the GPU box (outputs: gpurun_out/<tag>/): the -m gpu
# suite, the default bench line (all five algorithms), its rocprofv3 kernel
# stats, PMC passes per algorithm (FETCH_SIZE / WRITE_SIZE traffic + the SQ
# MFMA-busy and LDS passes of tools/run_pmc.sh), phase stamps of the headline
# kernel, smoke().
# XRD_EV_PMC="nice vox pointslam splatam" selects the PMC passes (default all).
tag=${1:-r06_final}
pmc=${XRD_EV_PMC:-nice vox pointslam splatam}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
t0=$SECONDS
# counter passes first: the bench line below reads its `traffic` from profiles/
for a in $pmc; do
  case $a in
    nice) timeout 600 bash tools/run_pmc.sh $tag/pmc_nice > $out/pmc_nice_tail.txt 2>&1 ;;
    vox) XRD_PMC_STEPS=5 XRD_PMC_WARMUP=3 timeout 600 bash tools/run_pmc.sh $tag/pmc_vox --algo vox-fusion > $out/pmc_vox_tail.txt 2>&1 ;;
    pointslam) XRD_PMC_STEPS=2 XRD_PMC_WARMUP=1 timeout 800 bash tools/run_pmc.sh $tag/pmc_pointslam --algo point-slam --first-iters 60 --no-steady-state > $out/pmc_ps_tail.txt 2>&1 ;;
    splatam) XRD_PMC_STEPS=2 XRD_PMC_WARMUP=1 timeout 800 bash tools/run_pmc.sh $tag/pmc_splatam --algo splaTAM > $out/pmc_spl_tail.txt 2>&1 ;;
  esac
  case $a in nice) j=r06_pmc.json;; *) j=r06_pmc_$a.json;; esac
  python tools/pmc_merge.py profiles/$j $out/pmc_$a "rocprofv3 --pmc passes (tools/run_pmc.sh: FETCH_SIZE, WRITE_SIZE, MFMA, LDS), evidence run $tag" > /dev/null \
    && cp profiles/$j $out/$j
  echo "pmc $a done $((SECONDS-t0))s"
done
export XRD_PARITY_REPORT=$out/parity_margins.txt
timeout 900 python -m pytest tests -m gpu -q > $out/gpu_tests.txt 2>&1
tail -3 $out/gpu_tests.txt; echo "tests done $((SECONDS-t0))s"
timeout 600 python bench.py > $out/bench_stdout.txt 2> $out/bench_stderr.txt
tail -1 $out/bench_stdout.txt > $out/bench_line.json
echo "bench done $((SECONDS-t0))s"; cut -c1-200 $out/bench_line.json
XRD_PROF_SCRIPT=bench.py timeout 500 bash tools/run_profile.sh $tag/prof --no-cpu-baseline --no-side-runs > $out/prof_tail.txt 2>&1
echo "prof done $((SECONDS-t0))s"; head -3 $out/prof/kernel_summary.txt
timeout 100 python tools/nice_bwd_timing.py 1000 200 > $out/nice_map_timing.txt 2>&1
timeout 100 python tools/pc_graph_timing.py 24508 > $out/pointslam_group_timing.txt 2>&1
timeout 100 python tools/nice_track_timing.py 200 > $out/nice_track_timing.txt 2>&1
[ -f tools/scratch/libxrdslam_hip_stamps.so ] && timeout 100 python tools/nice_map_stamps.py run 1000 > $out/nice_map_phases.txt 2>&1
timeout 400 python tools/kernel_counts.py > $out/kernel_counts.txt 2> $out/kernel_counts_err.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1
tail -1 $out/smoke.txt
echo "all done $((SECONDS-t0))s"
