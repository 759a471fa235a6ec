#!/bin/bash
# A/B several builds of liberasor_hip.so on the same box: tools/ab_many.sh "<bench args>" a.so b.so ...
ARGS=$1; shift
cp erasor_amd/liberasor_hip.so /tmp/lib_keep.so
for r in 1 2; do
  for v in "$@"; do
    cp $v erasor_amd/liberasor_hip.so
    echo -n "$v: "
    timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step; no look-ahead', d['ms_per_step_without_lookahead'], '; split us', d['roofline']['avg_launch_us'], d['roofline']['frac'])"
  done
done
cp /tmp/lib_keep.so erasor_amd/liberasor_hip.so
