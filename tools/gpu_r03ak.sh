#!/bin/bash
# round 3, call ak: the query's bucketing + bin statistics in one launch (k_qb_one), sort queues opened by the key kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ak
mkdir -p $OUT
cd $ROOT
timeout 1100 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2 3 4; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json new
  ERASOR_HIP_NO_QB_ONE=1 timeout 200 $B > $OUT/noqb_$r.json 2> /dev/null; line $OUT/noqb_$r.json new_without_qb_one
  cp variants/prev.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
for w in ouster128 large_scale_05; do timeout 200 $B --workload $w > $OUT/$w.json 2> /dev/null; line $OUT/$w.json $w; done
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --profile-all 2>&1 >/dev/null | tail -26 | grep -E "q_bucket|bin_stats|q_keys|q_esort |wall"
