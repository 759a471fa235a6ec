#!/bin/bash
# round 3, call az: no barrier packet for a join whose event is complete already when the step is enqueued
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03az
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "golden or split_ahead or prefetched or interleaved or grows" > $OUT/pytest_some.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_some.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
for r in 1 2 3 4; do
  timeout 200 $B > $OUT/skip_$r.json 2> /dev/null; line $OUT/skip_$r.json skip_complete_joins
  ERASOR_HIP_ALWAYS_WAIT=1 timeout 200 $B > $OUT/always_$r.json 2> /dev/null; line $OUT/always_$r.json always_wait
done
