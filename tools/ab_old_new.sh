#!/bin/bash
# old build (a copy of HEAD's sources + library under _ab_old/, made here before the call) against the working tree on ONE box, in the
# order old, new, new, old: the driver's statistic (seven 20-step passes) and 100-step passes, headline workload and config 4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
A="--no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr"
run() { timeout 300 python $1/bench.py $A $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$3', d['ms_per_step'], d['ms_per_step_all'], 'taken', d['overlapped_steps']['taken'], 'auto', [round(x,1) for x in (d['overlapped_steps']['auto']['plain_period_us'], d['overlapped_steps']['auto']['overlapped_period_us'])])"; }
for wl in ${AB_WORKLOADS:-"" "--workload large_scale_05"}; do
  for cfg in "--steps 20 --warmup 5 --repeats 7" "--steps 100 --warmup 5 --repeats 5"; do
    echo "== $wl $cfg"
    for rep in 1 2; do
      run _ab_old "$wl $cfg" old; run . "$wl $cfg" new; run . "$wl $cfg" new; run _ab_old "$wl $cfg" old
    done
  done
done
