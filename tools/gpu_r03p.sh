#!/bin/bash
# round 3, call p: the finisher of the scan sort on the level-synchronous sort, finisher grid sized by input
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03p
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2 3; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json final_sync
  cp variants/prev.so erasor_amd/liberasor_hip.so
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -26 | grep -E "q_esort|q_nn|wall"
python tools/save_map_probe.py 2>&1 | tail -3
