B="timeout 120 python bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --repeats 7"
run() { echo "== $1 $2"; env $1 $B $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['ms_per_step_all'], d['main_chain_us'])"; }
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hooks.py -m gpu -x -q -k "step_parity or exact_sort or one_bin or full_size_parity or config4 or voxelize" 2>&1 | tail -2
run "X=1"
run "X=1" "--workload large_scale_05"
ERASOR_HIP_SORT_STAMPS=1 timeout 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-workloads --repeats 1 2>&1 >/dev/null | grep -E "last per-bin|slowest reverted" | sed -n 20,27p | cut -c1-200
