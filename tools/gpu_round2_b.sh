#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k config4 > $OUT/pytest_config4.log 2>&1; tail -3 $OUT/pytest_config4.log
for q in 4 8; do
  for la in 1 2; do
    echo "GPU_MAX_HW_QUEUES=$q lookahead=$la"
    GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --lookahead $la --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['ms_per_step_without_lookahead'], d['roofline']['avg_launch_us'])"
  done
done
ERASOR_HIP_SORT_STAMPS=1 timeout 300 python bench.py --no-cpu-baseline --steps 5 --workload large_scale_05 2>&1 | grep -v "^{" | tail -12
ERASOR_HIP_SORT_STAMPS=1 timeout 300 python bench.py --no-cpu-baseline --steps 5 2>&1 | grep -v "^{" | tail -12
ERASOR_HIP_HOST_TIMING=1 timeout 300 python bench.py --no-cpu-baseline --steps 5 2>&1 | grep -v "^{" | tail -12
