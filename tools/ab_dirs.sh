#!/bin/bash
# several builds side by side on ONE box (each a directory with bench.py + erasor_amd/ like _ab_old/): the driver's statistic and 100-step
# passes, the directories in rotation, twice.   tools/ab_dirs.sh <dir> <dir> ...   (AB_WORKLOADS as in tools/ab_old_new.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
A="--no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr"
run() { timeout 300 python $1/bench.py $A $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-10s' % '$1', d['ms_per_step'], d['ms_per_step_all'][:3], 'auto', [round(x,1) for x in (d['overlapped_steps']['auto']['plain_period_us'], d['overlapped_steps']['auto']['overlapped_period_us'])])"; }
for wl in ${AB_WORKLOADS:-"--workload=seq05" "--workload=large_scale_05"}; do
  for cfg in "--steps 20 --warmup 5 --repeats 7" "--steps 100 --warmup 5 --repeats 5"; do
    echo "== $wl $cfg"
    for rep in 1 2; do for d in "$@"; do run $d "$wl $cfg"; done; done
  done
done
