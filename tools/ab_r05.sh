B="timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --mode seq-per-gpu --seqs 2"
run() { echo "== $1 $2"; env $1 $B $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], 'scans/s', d['ms_per_step'], 'ms per scan')"; }
run "ERASOR_HIP_OVERLAP=0" "--interleave off"
run "ERASOR_HIP_OVERLAP=0" "--interleave async"
run "ERASOR_HIP_OVERLAP=0 ERASOR_HIP_QSTREAMS=1" "--interleave async"
run "ERASOR_HIP_OVERLAP=0 ERASOR_HIP_QSTREAMS=1" "--interleave threads"
run "ERASOR_HIP_OVERLAP=0 ERASOR_HIP_QSTREAMS=1 GPU_MAX_HW_QUEUES=4" "--interleave async"
run "ERASOR_HIP_OVERLAP=0 ERASOR_HIP_QSTREAMS=2 GPU_MAX_HW_QUEUES=4" "--interleave async"
