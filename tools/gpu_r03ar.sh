#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python tools/export_cpp_bench.py /tmp/cppbench 40 > /dev/null 2>&1
python -c "
import ctypes
h=ctypes.CDLL('libamdhip64.so'); lo=ctypes.c_int(); hi=ctypes.c_int(); h.hipDeviceGetStreamPriorityRange(ctypes.byref(lo),ctypes.byref(hi)); print('priority range least', lo.value, 'greatest', hi.value)"
for r in 1 2 3 4; do
  timeout 200 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('same prio ', d['ms_per_callback'], d['ms_per_callback_next_node_announced'], d['ms_per_step_device_resident_two_ahead'])"
  ERASOR_HIP_QSTREAM_MIXED_PRIO=1 timeout 200 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed prio', d['ms_per_callback'], d['ms_per_callback_next_node_announced'], d['ms_per_step_device_resident_two_ahead'])"
done
for r in 1 2; do
ERASOR_HIP_QSTREAM_MIXED_PRIO=1 timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq05 mixed', d['value'], d['ms_per_step'])"
timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq05 same', d['value'], d['ms_per_step'])"
done
