#!/bin/bash
# round 3, call ay: one / two more wide levels of the scan sort (ESORT_WIDE_SLACK 5 / 6: fewer stragglers for k_esort_mid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ay
mkdir -p $OUT
cd $ROOT
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2 3; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/base_$r.json 2> /dev/null; line $OUT/base_$r.json slack_4
  for v in slack5 slack6; do
    cp variants/$v.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/${v}_$r.json 2> /dev/null; line $OUT/${v}_$r.json $v
  done
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
