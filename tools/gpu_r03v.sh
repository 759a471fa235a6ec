#!/bin/bash
# round 3, call v: wide sort levels in two launches (virtual median move, children routed by the swap kernel), DPP reductions
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03v
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2 3; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json two_launch_levels
  cp variants/prev.so erasor_amd/liberasor_hip.so
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --python-loop > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -26 | grep -E "q_esort|q_nn|voi_gather|bin_stats|wall"
python tools/save_map_probe.py 2>&1 | tail -2
ERASOR_HIP_HOST_TIMING=1 timeout 200 python bench.py --no-cpu-baseline --steps 8 --python-loop 2>&1 >/dev/null | grep "step host\] enqueue" | tail -4
