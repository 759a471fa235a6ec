#!/bin/bash
# round 3, call c: level-synchronous exact sort (DPP scan, pair-carrying stop lists) in R-GPF / per-bin voxelisation
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03c
mkdir -p $OUT
cd $ROOT
timeout 120 tools/esort_bench 2>&1 | tee $OUT/esort_bench.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'], 'split_us', d['roofline']['avg_launch_us'])"; }
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/bench_$r.json 2> $OUT/bench_$r.err; line $OUT/bench_$r.json new
done
ERASOR_HIP_SORT_STAMPS=1 timeout 200 python bench.py --no-cpu-baseline --steps 6 2>&1 >/dev/null | grep "slowest" | tail -6
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -24 > $OUT/breakdown_seq05.txt
grep -E "rgpf|bin_vox|wall" $OUT/breakdown_seq05.txt
