#!/bin/bash
# the launches of a few steady-state overlapped steps, all queues (tools/trace_window.py), under rocprofv3 --kernel-trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-tracewin}
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_trace -- python $ROOT/bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr "$@" > $OUT/bench_traced.json 2>/dev/null
cd $ROOT
python tools/trace_window.py /tmp/rp_trace 50 4 1 > $OUT/trace_window.txt 2>&1
cat $OUT/trace_window.txt
