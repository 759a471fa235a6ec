cd ${GRAFT_REPO_ROOT:-/root/repo}
A="--no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --steps 20 --warmup 5 --repeats 7"
for rep in 1 2 3; do for la in 7 6 5 4; do for lead in 3 2; do
timeout 300 python bench.py $A --lookahead $la --chain-lead $lead 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('lookahead $la lead $lead', d['ms_per_step'], d['overlapped_steps']['taken'], d['ms_per_step_all'][1:4])"
done; done; done
