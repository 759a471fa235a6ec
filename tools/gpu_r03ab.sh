#!/bin/bash
# round 3, call ab: the per-bin launch without the rare paths in its hot workgroups (two kernels / one launch with an out-of-line rare path)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ab
mkdir -p $OUT
cd $ROOT
timeout 1100 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2 3; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/two_$r.json 2> /dev/null; line $OUT/two_$r.json two_kernels
  cp variants/one_launch.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/one_$r.json 2> /dev/null; line $OUT/one_$r.json one_launch
  cp variants/prev.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B --workload large_scale_05 > $OUT/ls05_two.json 2> /dev/null; line $OUT/ls05_two.json ls05_two_kernels
cp variants/one_launch.so erasor_amd/liberasor_hip.so; timeout 200 $B --workload large_scale_05 > $OUT/ls05_one.json 2> /dev/null; line $OUT/ls05_one.json ls05_one_launch
cp variants/prev.so erasor_amd/liberasor_hip.so; timeout 200 $B --workload large_scale_05 > $OUT/ls05_prev.json 2> /dev/null; line $OUT/ls05_prev.json ls05_previous
cp /tmp/keep.so erasor_amd/liberasor_hip.so
bash tools/gpu_trace.sh r03ab 2>&1 | tail -15
