#!/bin/bash
# round 3, call g: chunk scan without LDS staging, output layout folded into SRT + write-back
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03g
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/fold_$r.json 2> $OUT/fold_$r.err; line $OUT/fold_$r.json fold
  ERASOR_HIP_NO_FOLD=1 timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/nofold_$r.json 2> $OUT/nofold_$r.err; line $OUT/nofold_$r.json nofold
done
bash tools/gpu_trace.sh r03g 2>&1 | tail -15
timeout 300 python bench.py --no-cpu-baseline --steps 30 --workload large_scale_05 > $OUT/ls05.json 2> $OUT/ls05.err; line $OUT/ls05.json ls05
GPU_MAX_HW_QUEUES=8 timeout 400 python bench.py --mode seq-per-gpu --interleave async --steps 20 --warmup 3 --no-cpu-baseline > $OUT/seq_async.json 2> $OUT/seq_async.err; line $OUT/seq_async.json seq_async_q8
