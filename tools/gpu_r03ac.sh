#!/bin/bash
# round 3, call ac: k_revert_bins with a half / half grid (common path | rare paths out of line); tests, A/B, then the evidence for profiles/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ac
mkdir -p $OUT
cd $ROOT
timeout 1100 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json new
  cp variants/prev.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B --workload large_scale_05 > $OUT/ls05_new_$r.json 2> /dev/null; line $OUT/ls05_new_$r.json ls05_new
  cp variants/prev.so erasor_amd/liberasor_hip.so; timeout 200 $B --workload large_scale_05 > $OUT/ls05_prev_$r.json 2> /dev/null; line $OUT/ls05_prev_$r.json ls05_previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
bash tools/collect_profiles.sh r03ac_seq05 > $OUT/collect_seq05.log 2>&1; tail -2 $OUT/collect_seq05.log
bash tools/collect_profiles.sh r03ac_ls05 --workload large_scale_05 > $OUT/collect_ls05.log 2>&1; tail -2 $OUT/collect_ls05.log
python -c "
import json
for t in ('seq05','ls05'):
    d=json.loads(open('gpurun_out/profiles_r03ac_%s/bench.json'%t).read().strip().split('\n')[-1])
    print(t, d['value'], d['ms_per_step'], d.get('ms_per_step_without_lookahead'), d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline'].get('value'), d.get('parity_checked_steps'), d.get('final_map_checked'))
"
bash tools/gpu_trace.sh r03ac 2>&1 | tail -14
