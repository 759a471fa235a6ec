#!/bin/bash
# round 3, closing call 5: the whole GPU suite + smoke on the final build, the evidence for profiles/ (both workloads), the secondary modes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03bc
mkdir -p $OUT
cd $ROOT
timeout 1100 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/collect_profiles.sh r03bc_seq05 > $OUT/collect_seq05.log 2>&1; tail -2 $OUT/collect_seq05.log
bash tools/collect_profiles.sh r03bc_ls05 --workload large_scale_05 > $OUT/collect_ls05.log 2>&1; tail -2 $OUT/collect_ls05.log
python -c "
import json
for t in ('seq05','ls05'):
    d=json.loads(open('gpurun_out/profiles_r03bc_%s/bench.json'%t).read().strip().split('\n')[-1])
    print(t, d['value'], d['ms_per_step'], d.get('ms_per_step_without_lookahead'), d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline'].get('value'), d.get('parity_checked_steps'), d.get('final_map_checked'))
"
timeout 200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; python -c "
import json; d=json.loads(open('$OUT/bench_driver_args.json').read().strip().split('\n')[-1]); print('driver args', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['rocprofv3_kernel_avg_us'])"
timeout 200 python bench.py --mode seq-per-gpu --no-cpu-baseline --steps 100 --warmup 10 > $OUT/seq_per_gpu.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/seq_per_gpu.json').read().strip().split('\n')[-1]); print('seq-per-gpu', d['value'], d['ms_per_step'])"
python tools/export_cpp_bench.py /tmp/cppbench 60 > /dev/null 2>&1 && timeout 200 erasor_amd/erasor_offline_demo --bench /tmp/cppbench 50 6 2>/dev/null | tail -1 > $OUT/cpp_bench.json; cat $OUT/cpp_bench.json
