#!/bin/bash
# round 3, call aq: 16 hardware queues by default (library constructor / bench.py): several handles in one process
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03aq
mkdir -p $OUT
cd $ROOT
python tools/export_cpp_bench.py /tmp/cppbench 40 > /dev/null 2>&1
for r in 1 2 3; do
  timeout 200 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cpp', d['ms_per_callback'], d['ms_per_callback_next_node_announced'], d['ms_per_step_device_resident_two_ahead'])"
done
for il in off async threads; do
  timeout 300 python bench.py --mode seq-per-gpu --interleave $il --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq-per-gpu interleave $il', d['value'], d['ms_per_step'])"
done
for n in 2 3; do
  timeout 300 python bench.py --mode seq-per-gpu --interleave async --seqs $n --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq-per-gpu async, $n sequences', d['value'], d['ms_per_step'])"
done
timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seq05', d['value'], d['ms_per_step'])"
