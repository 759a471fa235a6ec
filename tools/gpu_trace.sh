#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-trace}
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_trace -- python $ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-workloads "$@" > $OUT/bench_traced.json 2>/dev/null
cd $ROOT
python tools/trace_main.py /tmp/rp_trace 9 > $OUT/trace_main.txt 2>&1; python tools/trace_main.py /tmp/rp_trace 10 >> $OUT/trace_main.txt 2>&1
python tools/trace_query.py /tmp/rp_trace -4 > $OUT/trace_query.txt 2>&1; python tools/trace_query.py /tmp/rp_trace -5 >> $OUT/trace_query.txt 2>&1
cat $OUT/trace_main.txt $OUT/trace_query.txt
