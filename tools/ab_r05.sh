B="timeout 120 python bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --repeats 5"
run() { echo "== $1 $2"; env $1 $B $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['ms_per_step_all'], d['overlapped_steps'])"; }
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nodes_announced or full_size or step_parity or prefetched or rows_in or large_scale or config4 or alternative" 2>&1 | tail -3
run "X=1" "--workload large_scale_05"
run "ERASOR_HIP_NO_AHEAD_SCATTER=1" "--workload large_scale_05"
run "ERASOR_HIP_OVERLAP=0" "--workload large_scale_05"
run "ERASOR_HIP_OVERLAP=1"
run "ERASOR_HIP_OVERLAP=1 ERASOR_HIP_NO_AHEAD_SCATTER=1"
run "X=1" "--workload seq05_yaml"
