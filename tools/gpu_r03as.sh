#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python tools/export_cpp_bench.py /tmp/cppbench 40 > /dev/null 2>&1
for q in 4 16 32; do
  export GPU_MAX_HW_QUEUES=$q
  for r in 1 2 3 4 5; do
    timeout 100 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $q:', d['ms_per_callback'], d['ms_per_callback_next_node_announced'], d['ms_per_step_device_resident_two_ahead'])"
  done
done
