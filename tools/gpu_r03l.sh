#!/bin/bash
# round 3, call l: host scans through pinned staging (async H2D), roofline events on every fourth split launch
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03l
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'], 'split', d['roofline']['avg_launch_us'], d['roofline']['launches'], d['roofline']['frac'])"; }
for r in 1 2 3; do
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/b_$r.json 2> /dev/null; line $OUT/b_$r.json driver_args
done
timeout 300 python tools/export_cpp_bench.py /tmp/cppbench 36 > $OUT/export.log 2>&1
timeout 300 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 > $OUT/cpp_bench.json 2> $OUT/cpp_bench.err; echo "cpp bench rc=$?"; cut -c1-400 $OUT/cpp_bench.json
