#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03an
mkdir -p $OUT
cd $ROOT
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
timeout 200 $B --python-loop > $OUT/pyloop.json 2>/dev/null; line $OUT/pyloop.json python_loop
timeout 200 $B > $OUT/native.json 2>/dev/null; line $OUT/native.json native_loop
python tools/export_cpp_bench.py /tmp/cppbench 40 > /dev/null 2>&1
timeout 200 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 2>$OUT/cpp.err | tail -1 | cut -c1-420
ERASOR_HIP_HOST_TIMING=1 timeout 200 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 8 4 2>&1 | grep -E "step host|enqueue" | tail -6
tail -3 $OUT/cpp.err
