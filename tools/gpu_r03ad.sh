#!/bin/bash
# round 3, call ad: k_revert_bins (common path only) + k_revert_bins_rare on a side stream / serial
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ad
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "golden or config4 or split_ahead or grows" > $OUT/pytest_some.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_some.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2 3; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/side_$r.json 2> /dev/null; line $OUT/side_$r.json rare_on_side_stream
  ERASOR_HIP_RARE_SERIAL=1 timeout 200 $B > $OUT/serial_$r.json 2> /dev/null; line $OUT/serial_$r.json rare_serial
  cp variants/prev.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
for r in 1 2; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B --workload large_scale_05 > $OUT/ls05_side_$r.json 2> /dev/null; line $OUT/ls05_side_$r.json ls05_side
  ERASOR_HIP_RARE_SERIAL=1 timeout 200 $B --workload large_scale_05 > $OUT/ls05_serial_$r.json 2> /dev/null; line $OUT/ls05_serial_$r.json ls05_serial
  cp variants/prev.so erasor_amd/liberasor_hip.so; timeout 200 $B --workload large_scale_05 > $OUT/ls05_prev_$r.json 2> /dev/null; line $OUT/ls05_prev_$r.json ls05_previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
bash tools/gpu_trace.sh r03ad 2>&1 | tail -14
