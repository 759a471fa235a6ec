B="timeout 120 python bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --repeats 5"
run() { echo "== $1 $2"; env $1 $B $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['ms_per_step_all'], d['overlapped_steps'])"; }
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nodes_announced or step_parity or large_scale or config4 or full_size_bench" 2>&1 | tail -2
run "X=1" "--workload large_scale_05"
run "ERASOR_HIP_OVERLAP=1"
ERASOR_HIP_OVERLAP=1 ERASOR_HIP_CHAIN_STAMPS=1 timeout 100 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extra-workloads --repeats 1 2>&1 >/dev/null | grep "^\[stamps" | sed -n 30,33p | cut -c1-330
