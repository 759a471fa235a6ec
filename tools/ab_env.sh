#!/bin/bash
# A/B environment settings on the same box: tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...   ("-" = no setting)
ARGS=$1; shift
for r in 1 2; do
  for e in "$@"; do
    echo -n "$e: "
    if [ "$e" = "-" ]; then timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $ARGS 2>/dev/null | tail -1 > /tmp/ab.json
    else env $e timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $ARGS 2>/dev/null | tail -1 > /tmp/ab.json; fi
    python -c "import sys,json; d=json.loads(open('/tmp/ab.json').read()); print(d['ms_per_step'], 'ms/step; no look-ahead', d['ms_per_step_without_lookahead'], '; split us', d['roofline']['avg_launch_us'])"
  done
done
