#!/bin/bash
# round 3, call j: node loop in native code (erasor_hip_run_nodes), shim look-ahead buffers fixed, C++ drop-in bench again
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03j
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'], d['config']['node_loop'][:6], d.get('parity_checked_steps'))"; }
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/native_$r.json 2> $OUT/native_$r.err; line $OUT/native_$r.json native
  timeout 200 python bench.py --no-cpu-baseline --steps 30 --python-loop > $OUT/python_$r.json 2> $OUT/python_$r.err; line $OUT/python_$r.json python
done
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; line $OUT/bench_default.json default_with_cpu
timeout 300 python tools/export_cpp_bench.py /tmp/cppbench 36 > $OUT/export.log 2>&1
timeout 300 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 > $OUT/cpp_bench.json 2> $OUT/cpp_bench.err; echo "cpp bench rc=$?"; cut -c1-420 $OUT/cpp_bench.json
