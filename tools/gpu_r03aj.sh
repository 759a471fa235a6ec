#!/bin/bash
# round 3, call aj: DPP reductions in the per-bin code and k_bbox, two divisions fewer per Jacobi rotation
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03aj
mkdir -p $OUT
cd $ROOT
timeout 1100 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2 3 4; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json new
  cp variants/prev.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
ERASOR_HIP_SORT_STAMPS=1 timeout 200 $B 2>&1 >/dev/null | grep -i "slowest reverted\|10 ns ticks" | head -4
