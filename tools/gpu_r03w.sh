#!/bin/bash
# round 3, call w: label search's first shell without the triple loop (slab table), binvox centroid sums with loads in flight
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03w
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2 3; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json new
  cp variants/prev.so erasor_amd/liberasor_hip.so
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -26 | grep -E "q_nn|q_centroids|rgpf|bin_vox|wall"
bash tools/gpu_trace.sh r03w 2>&1 | tail -12 | grep -E "revert|gather|srt|span"
python tools/save_map_probe.py 2>&1 | tail -2
