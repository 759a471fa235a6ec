#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python tools/export_cpp_bench.py /tmp/cppbench 130 > /dev/null 2>&1
for r in 1 2 3 4 5 6; do
  timeout 200 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 100 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=100 W=20:', d['ms_per_callback'], d['ms_per_callback_next_node_announced'], d['ms_per_step_device_resident_two_ahead'])"
done
