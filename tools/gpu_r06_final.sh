#!/bin/bash
# the closing pass of round 6 on ONE box: the GPU suite, the rocprofv3 / PMC evidence of this very build (tools/collect_profiles.sh, both bench
# workloads) -- copied into profiles/ of the box's tree so that the bench line that follows resolves its roofline fields from THIS build --,
# the driver's command, 100-step passes plain / overlapped
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r06final}
mkdir -p $OUT
cd $ROOT
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $OUT/pytest_gpu.log | tail -2
tools/collect_profiles.sh r06 > $OUT/collect_r06.log 2>&1
tools/collect_profiles.sh r06_ls05 --workload large_scale_05 > $OUT/collect_r06_ls05.log 2>&1
cp gpurun_out/kernel_stats_latest*.csv gpurun_out/pmc_latest*.json gpurun_out/latest_meta*.json profiles/ 2>/dev/null
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --repeats 5 --steps 100"
for m in 0 1; do echo "== steps 100, ERASOR_HIP_OVERLAP=$m"; ERASOR_HIP_OVERLAP=$m $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['ms_per_step_all'])"; done
