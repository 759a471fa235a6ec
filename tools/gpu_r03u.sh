#!/bin/bash
# round 3, call u: whole-workgroup partitions (k_esort_mid / _level, bins > 2048 keys) with DPP scans; alternative launch paths test
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03u
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json new
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload large_scale_05 > $OUT/ls05_$r.json 2> /dev/null; line $OUT/ls05_$r.json ls05
done
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -26 | grep -E "q_esort|rgpf|bin_vox|wall"
