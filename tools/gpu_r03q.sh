#!/bin/bash
# round 3, call q: R-GPF + per-bin voxelisation in one launch (k_revert_bins)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03q
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
for r in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/fused_$r.json 2> /dev/null; line $OUT/fused_$r.json fused
  ERASOR_HIP_NO_FUSE=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/nofuse_$r.json 2> /dev/null; line $OUT/nofuse_$r.json two_launches
done
timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload large_scale_05 > $OUT/ls05_fused.json 2> /dev/null; line $OUT/ls05_fused.json ls05_fused
ERASOR_HIP_NO_FUSE=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload large_scale_05 > $OUT/ls05_nofuse.json 2> /dev/null; line $OUT/ls05_nofuse.json ls05_two_launches
bash tools/gpu_trace.sh r03q 2>&1 | tail -13
