#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03k
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/export_cpp_bench.py /tmp/cppbench 20 > $OUT/export.log 2>&1
ERASOR_HIP_HOST_TIMING=1 timeout 300 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 12 4 > $OUT/cpp_bench.json 2> $OUT/cpp_bench.err; echo "cpp bench rc=$?"; cut -c1-480 $OUT/cpp_bench.json
grep -n "step host\|query chain\|map chain" $OUT/cpp_bench.err | sed -n 60,100p
# A/B: is it the split's event bracket (profiling(2)) that costs the Python bench ~18 us per step?
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/prof2_$r.json 2> /dev/null; line $OUT/prof2_$r.json split_events_on
  ERASOR_BENCH_NO_SPLIT_EVENTS=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/prof0_$r.json 2> /dev/null; line $OUT/prof0_$r.json split_events_off
done
cp erasor_amd/liberasor_hip.so /tmp/keep.so
cp variants/r02.so erasor_amd/liberasor_hip.so
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --python-loop > $OUT/r02lib_$r.json 2> /dev/null; line $OUT/r02lib_$r.json round2_library
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
