#!/bin/bash
# round 3, call d: SVD in registers, sync sort with idle wavefronts parked / select-based median; trace of the main stream
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03d
mkdir -p $OUT
cd $ROOT
timeout 120 tools/esort_bench 2>&1 | grep -E "n= 100 kind=0|n= 600 kind=0|n=1000|n=2048 kind=0|OK|FAIL" | tee $OUT/esort_bench.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'], 'split_us', d['roofline']['avg_launch_us'])"; }
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/bench_$r.json 2> $OUT/bench_$r.err; line $OUT/bench_$r.json new
done
ERASOR_HIP_SORT_STAMPS=1 timeout 200 python bench.py --no-cpu-baseline --steps 6 2>&1 >/dev/null | grep "slowest" | grep -v "level-0\|esort slowest" | tail -6
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -24 > $OUT/breakdown_seq05.txt
grep -E "rgpf|bin_vox|wall" $OUT/breakdown_seq05.txt
bash tools/gpu_trace.sh r03d 2>&1 | tail -16
