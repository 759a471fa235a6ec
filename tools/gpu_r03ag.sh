#!/bin/bash
# round 3, closing call 2: the whole GPU suite + smoke on the build with the out-of-line per-bin sort, then the evidence for profiles/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ag
mkdir -p $OUT
cd $ROOT
timeout 1100 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/collect_profiles.sh r03ag_seq05 > $OUT/collect_seq05.log 2>&1; tail -2 $OUT/collect_seq05.log
bash tools/collect_profiles.sh r03ag_ls05 --workload large_scale_05 > $OUT/collect_ls05.log 2>&1; tail -2 $OUT/collect_ls05.log
python -c "
import json
for t in ('seq05','ls05'):
    d=json.loads(open('gpurun_out/profiles_r03ag_%s/bench.json'%t).read().strip().split('\n')[-1])
    print(t, d['value'], d['ms_per_step'], d.get('ms_per_step_without_lookahead'), d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline'].get('value'), d.get('parity_checked_steps'), d.get('final_map_checked'))
"
