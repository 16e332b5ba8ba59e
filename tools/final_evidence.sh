#!/bin/bash
# round-end evidence on the GPU box: bench lines of every algorithm, the
# rocprofv3 kernel summary of the default line, smoke(), the -m gpu suite.
# usage: bash tools/final_evidence.sh <tag>      (outputs: gpurun_out/<tag>/)
tag=${1:-final}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
t0=$SECONDS
timeout ${2:-330} python -m pytest tests -m gpu -q > $out/gpu_tests.txt 2>&1
tail -3 $out/gpu_tests.txt
echo "tests done $((SECONDS-t0))s"
timeout 200 python bench.py > $out/nice_stdout.txt 2> $out/nice_stderr.txt
tail -1 $out/nice_stdout.txt > $out/bench_line.json
echo "nice done $((SECONDS-t0))s"; cut -c1-300 $out/bench_line.json
timeout 150 bash tools/run_profile.sh $tag/prof --steps 30 --warmup 5 --no-cpu-baseline > $out/prof_tail.txt 2>&1
echo "prof done $((SECONDS-t0))s"
timeout 200 python bench.py --algo point-slam > $out/ps_stdout.txt 2> $out/ps_stderr.txt
tail -1 $out/ps_stdout.txt > $out/bench_pointslam_line.json
echo "point-slam done $((SECONDS-t0))s"; cut -c1-200 $out/bench_pointslam_line.json
timeout 120 python bench.py --algo vox-fusion > $out/vox_stdout.txt 2> $out/vox_stderr.txt
tail -1 $out/vox_stdout.txt > $out/bench_voxfusion_line.json
echo "vox done $((SECONDS-t0))s"; cut -c1-200 $out/bench_voxfusion_line.json
timeout 150 python bench.py --algo splaTAM > $out/splatam_stdout.txt 2> $out/splatam_stderr.txt
tail -1 $out/splatam_stdout.txt > $out/bench_splatam_line.json
echo "splatam done $((SECONDS-t0))s"; cut -c1-200 $out/bench_splatam_line.json
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1
tail -1 $out/smoke.txt
echo "all done $((SECONDS-t0))s"
