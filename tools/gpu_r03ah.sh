#!/bin/bash
# round 3, call ah: three nodes announced ahead (four query sides)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ah
mkdir -p $OUT
cd $ROOT
timeout 1100 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
for w in seq05 ouster128 large_scale_05 seq05_yaml; do
  for r in 1 2; do
    for la in 2 3; do
      timeout 200 $B --workload $w --lookahead $la > $OUT/${w}_la${la}_$r.json 2> /dev/null; line $OUT/${w}_la${la}_$r.json ${w}_ahead_$la
    done
  done
done
