#!/bin/bash
# round 3, call m: the default bench line with the other workloads embedded, the union exchange, profiles of both workloads
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03m
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
/usr/bin/time -v timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; grep "Elapsed (wall" $OUT/bench_default.err
python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().split('\n')[-1])
print('default', d['value'], d['ms_per_step'], 'parity', d['parity_checked_steps'], d['final_map_checked'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['launches'])
for e in d.get('other_workloads', []): print('  ', e.get('workload'), e.get('value'), e.get('ms_per_step'), e.get('roofline', {}).get('frac'), e.get('roofline', {}).get('step_frac'), e.get('wall_s'), e.get('error'))"
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --union-eval 8 > $OUT/union.json 2> $OUT/union.err; python -c "
import json; d=json.loads(open('$OUT/union.json').read().strip().split('\n')[-1]); print(d['union_exchange'])"
bash tools/collect_profiles.sh r03m_seq05 > $OUT/collect_seq05.log 2>&1; tail -2 $OUT/collect_seq05.log
bash tools/collect_profiles.sh r03m_ls05 --workload large_scale_05 > $OUT/collect_ls05.log 2>&1; tail -2 $OUT/collect_ls05.log
