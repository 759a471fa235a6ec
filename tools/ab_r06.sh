#!/bin/bash
# round 6 A/B on ONE box: the GPU suite, then the headline workload with the query chains on their own / shared by 2, 3, 4 scans,
# with and without overlapped steps.  usage: tools/ab_r06.sh TAG [notest]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06a}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -z "$2" ]; then
  timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log
fi
B="timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --repeats 5"
run() { echo "== $1 | $2"; env $1 $B $2 2>$OUT/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['ms_per_step_all'], 'chain', d['main_chain_us'], 'period', d['steady_state_ms_per_step'], d['overlapped_steps'], d['shared_chain_launches'])" || tail -5 $OUT/err.txt; }
for wl in "" "--workload large_scale_05" "--workload ouster128" "--workload seq05_yaml"; do
  run "ERASOR_HIP_OVERLAP=0" "--chain-batch 1 --lookahead 3 $wl"
  run "ERASOR_HIP_OVERLAP=0" "--chain-batch 2 --lookahead 6 $wl"
  run "ERASOR_HIP_OVERLAP=1" "--chain-batch 1 --lookahead 3 $wl"
  run "ERASOR_HIP_OVERLAP=1" "--chain-batch 2 --lookahead 6 $wl"
  run "ERASOR_HIP_OVERLAP=1" "--chain-batch 3 --lookahead 7 --chain-lead 3 $wl"
  run "ERASOR_HIP_OVERLAP=1" "--chain-batch 4 --lookahead 7 --chain-lead 3 $wl"
  run "ERASOR_HIP_OVERLAP=0" "--chain-batch 3 --lookahead 7 --chain-lead 3 $wl"
done
