#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ap
mkdir -p $OUT
cd $ROOT
python tools/export_cpp_bench.py /tmp/cppbench 40 > /dev/null 2>&1
for q in default 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "== GPU_MAX_HW_QUEUES=$q"
  timeout 200 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cpp', d['ms_per_callback'], d['ms_per_callback_next_node_announced'], d['ms_per_step_device_resident_two_ahead'])"
  timeout 200 python bench.py --mode seq-per-gpu --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq-per-gpu', d['value'], d['ms_per_step'])"
  timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq05', d['value'], d['ms_per_step'])"
done
