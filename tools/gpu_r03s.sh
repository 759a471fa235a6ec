#!/bin/bash
# round 3, call s: R-POD key decided on float32 estimates with an exact float64 fallback near boundaries; k_step_end reads first
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03s
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2 3; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json fast_key
  cp variants/prev.so erasor_amd/liberasor_hip.so
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-workloads > $OUT/bench_parity.json 2> $OUT/bench_parity.err; python -c "
import json; d=json.loads(open('$OUT/bench_parity.json').read().strip().split('\n')[-1]); print('parity', d['parity_checked_steps'], d['final_map_checked'], d['ms_per_step'])"
timeout 300 python bench.py --steps 12 --warmup 3 --no-extra-workloads --workload large_scale_05 > $OUT/ls05_parity.json 2> $OUT/ls05_parity.err; python -c "
import json; d=json.loads(open('$OUT/ls05_parity.json').read().strip().split('\n')[-1]); print('ls05 parity', d['parity_checked_steps'], d['final_map_checked'], d['ms_per_step'])"
bash tools/gpu_trace.sh r03s 2>&1 | tail -13
