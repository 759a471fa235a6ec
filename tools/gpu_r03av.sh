#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python tools/export_cpp_bench.py /tmp/cppbench 60 > /dev/null 2>&1
for m in 1 2 0; do
  export ERASOR_HIP_PRIO_MODE=$m
  for r in 1 2 3 4 5 6; do
    timeout 100 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 50 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio mode $m: cpp', d['ms_per_callback'], d['ms_per_callback_next_node_announced'], d['ms_per_step_device_resident_two_ahead'])"
  done
  for r in 1 2 3; do
    timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio mode $m: seq05', d['value'], d['ms_per_step'], d['ms_per_step_without_lookahead'])"
  done
done
