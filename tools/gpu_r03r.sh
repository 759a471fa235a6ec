#!/bin/bash
# round 3, call r: Scan Ratio Test's first pass inside the map-side bin statistics (k_bin_stats_srt), DPP column scan
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03r
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
for r in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json srt_ahead
  ERASOR_HIP_NO_SRT_AHEAD=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/old_$r.json 2> /dev/null; line $OUT/old_$r.json srt_in_one_wg
done
bash tools/gpu_trace.sh r03r 2>&1 | tail -13
