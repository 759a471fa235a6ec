#!/bin/bash
# round 3, call o: how much do the look-ahead chains cost the main chain?  (A/B by environment / arguments)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03o
mkdir -p $OUT
cd $ROOT
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/base_$r.json 2> /dev/null; line $OUT/base_$r.json base
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --lookahead 1 > $OUT/la1_$r.json 2> /dev/null; line $OUT/la1_$r.json lookahead1
  ERASOR_HIP_FINAL_GRID=256 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/fg256_$r.json 2> /dev/null; line $OUT/fg256_$r.json final_grid_256
  ERASOR_HIP_FINAL_GRID=128 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/fg128_$r.json 2> /dev/null; line $OUT/fg128_$r.json final_grid_128
  ERASOR_HIP_REV_GRID=16 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/rg16_$r.json 2> /dev/null; line $OUT/rg16_$r.json rev_grid_16
  ERASOR_HIP_REV_GRID=32 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/rg32_$r.json 2> /dev/null; line $OUT/rg32_$r.json rev_grid_32
done
