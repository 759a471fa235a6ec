B="timeout 120 python bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --repeats 5"
run() { echo "== $1 $2"; env $1 $B $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['ms_per_step_all'], d['overlapped_steps'])"; }
run "ERASOR_HIP_OVERLAP=1"
run "ERASOR_HIP_OVERLAP=1 ERASOR_HIP_CHAIN_SPLIT=1"
run "ERASOR_HIP_OVERLAP=0"
run "ERASOR_HIP_OVERLAP=1 ERASOR_HIP_CHAIN_SPLIT=1" "--workload large_scale_05"
run "ERASOR_HIP_OVERLAP=1" "--workload large_scale_05"
run "ERASOR_HIP_OVERLAP=1 ERASOR_HIP_CHAIN_SPLIT=1" "--workload seq05_yaml"
