#!/bin/bash
# one table: auto vs forced modes on the bench workloads
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06g}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "alternative_launch or handle_decides or random_operation" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --repeats 7"
run() { echo "== $1 | $2"; env $1 $B $2 2>$OUT/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['ms_per_step_all'], d['overlapped_steps'], d['shared_chain_launches'])" || tail -5 $OUT/err.txt; }
for wl in "" "--workload seq05_yaml" "--workload large_scale_05" "--workload large_scale_05 --large-scale-mode on" "--workload ouster128"; do
run "ERASOR_HIP_OVERLAP=0" "$wl"
run "ERASOR_HIP_OVERLAP=1" "$wl"
run "X=1" "$wl"
done
