#!/bin/bash
# round 3, call i: C++ drop-in bench (host clouds in / rejected clouds out), ROS adapter with a one-message hold, union eval
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03i
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 300 python tools/export_cpp_bench.py /tmp/cppbench 36 > $OUT/export.log 2>&1; tail -1 $OUT/export.log
timeout 300 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 > $OUT/cpp_bench.json 2> $OUT/cpp_bench.err; echo "cpp bench rc=$?"; cat $OUT/cpp_bench.json; tail -2 $OUT/cpp_bench.err
timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 4 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('python bench', d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"
