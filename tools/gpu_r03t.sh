#!/bin/bash
# round 3: profiles of the closing build, both workloads
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03t
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash tools/collect_profiles.sh r03t_seq05 > $OUT/collect_seq05.log 2>&1; tail -1 $OUT/collect_seq05.log
bash tools/collect_profiles.sh r03t_ls05 --workload large_scale_05 > $OUT/collect_ls05.log 2>&1; tail -1 $OUT/collect_ls05.log
python -c "
import json
for t in ('seq05','ls05'):
    d=json.loads(open('$ROOT/gpurun_out/profiles_r03t_'+t+'/bench.json').read().strip().split('\n')[-1])
    print(t, d['value'], d['ms_per_step'], d['parity_checked_steps'], d['final_map_checked'], d['roofline']['frac'], d['roofline']['step_frac'], [ (e['workload'], e.get('ms_per_step')) for e in d.get('other_workloads', [])])"
