#!/bin/bash
# A/B environment settings on the same box: tools/ab_env.sh "VAR=a" "VAR=b" ... (each run: bench.py 30 steps)
for r in 1 2; do
  for e in "$@"; do
    echo -n "$e: "
    env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step', d['roofline']['avg_launch_us'], 'us split')"
  done
done
