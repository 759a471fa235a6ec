#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06d}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_trace
ERASOR_HIP_OVERLAP=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_trace -- python $ROOT/bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-extra-workloads --no-pr-rr --no-callback-bench --chain-batch 3 --lookahead 7 > $OUT/bench_traced.json 2>/dev/null
cd $ROOT
python tools/trace_window.py /tmp/rp_trace 20 3 1 > $OUT/window_ov1.txt 2>&1
python tools/trace_window.py /tmp/rp_trace 20 2 0 > $OUT/window_ov1_all.txt 2>&1
cat $OUT/window_ov1.txt | cut -c1-150
B="timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --repeats 5"
run() { echo "== $1 | $2"; env $1 $B $2 2>$OUT/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['ms_per_step_all'], 'chain', d['main_chain_us'], d['overlapped_steps'], d['shared_chain_launches'])" || tail -5 $OUT/err.txt; }
run "ERASOR_HIP_OVERLAP=0" "--chain-batch 1 --lookahead 3"
run "ERASOR_HIP_OVERLAP=1" "--chain-batch 3 --lookahead 7"
run "ERASOR_HIP_OVERLAP=1" "--chain-batch 2 --lookahead 6"
