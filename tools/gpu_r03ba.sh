#!/bin/bash
# round 3, call ba: the shim's host path (repack buffers reused, announced cloud recognised by size + sampled points)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ba
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_shim.py -m gpu -x -q > $OUT/pytest_shim.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_shim.log
python tools/export_cpp_bench.py /tmp/cppbench 60 > /dev/null 2>&1
for r in 1 2 3; do
  timeout 100 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 50 6 2>/dev/null | tail -1 | tee $OUT/cpp_bench_$r.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cpp', d['ms_per_callback'], d['ms_per_callback_next_node_announced'], d['of_which_announce_next'], d['ms_per_step_device_resident_two_ahead'])"
done
