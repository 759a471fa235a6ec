#!/bin/bash
# round 3, call af: the per-bin stages' level-synchronous sort out of line (own register allocation, no spill reloads inside its level loop)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03af
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "golden or config4 or split_ahead or grows" > $OUT/pytest_some.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_some.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2 3 4; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json new
  cp variants/prev.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
for r in 1 2; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B --workload large_scale_05 > $OUT/ls05_new_$r.json 2> /dev/null; line $OUT/ls05_new_$r.json ls05_new
  cp variants/prev.so erasor_amd/liberasor_hip.so; timeout 200 $B --workload large_scale_05 > $OUT/ls05_prev_$r.json 2> /dev/null; line $OUT/ls05_prev_$r.json ls05_previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
bash tools/gpu_trace.sh r03af 2>&1 | tail -14 | grep -E "revert|span"
