#!/bin/bash
# round 3, call h: block_excl_scan on DPP (k_srt4, k_layout4, scans), write-back prologue on one wavefront
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03h
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/fold_$r.json 2> $OUT/fold_$r.err; line $OUT/fold_$r.json fold
  ERASOR_HIP_NO_FOLD=1 timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/nofold_$r.json 2> $OUT/nofold_$r.err; line $OUT/nofold_$r.json nofold
done
bash tools/gpu_trace.sh r03h 2>&1 | tail -14
