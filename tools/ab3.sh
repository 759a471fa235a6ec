#!/bin/bash
# A/B/C...: tools/ab3.sh lib1.so lib2.so ...  (two rounds, bench.py 30 steps each)
cp erasor_amd/liberasor_hip.so /tmp/lib_keep.so
for r in 1 2; do
  for v in "$@"; do
    cp $v erasor_amd/liberasor_hip.so
    echo -n "$v: "
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step')"
  done
done
cp /tmp/lib_keep.so erasor_amd/liberasor_hip.so
