#!/bin/bash
# query chains in sets of N (bench.py --chain-batch / --chain-lead = erasor_hip_chain_batch) on ONE box: the driver's statistic and 100-step passes
#   tools/sweep_chain_sets.sh "<workload args>" "<batch lead>" ...      e.g.  tools/sweep_chain_sets.sh "--workload large_scale_05" "3 3" "2 3"
cd ${GRAFT_REPO_ROOT:-/root/repo}
A="--no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr"
W="$1"; shift
for rep in 1 2; do for cb in "$@"; do set -- $cb
for cfg in "--steps 20 --warmup 5 --repeats 7" "--steps 100 --warmup 5 --repeats 5"; do
timeout 300 python bench.py $A $W $cfg --chain-batch $1 --chain-lead $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$W batch $1 lead $2 [$cfg]', d['ms_per_step'], d['overlapped_steps']['taken'])"
done; done; done
