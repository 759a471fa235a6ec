#!/bin/bash
# A/B two builds of liberasor_hip.so on the same box: tools/ab.sh variants/a.so variants/b.so [rounds]
A=$1; B=$2; R=${3:-3}
cp erasor_amd/liberasor_hip.so /tmp/lib_keep.so
for r in $(seq $R); do
  for v in $A $B; do
    cp $v erasor_amd/liberasor_hip.so
    echo -n "$v: "
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step', d['roofline']['avg_launch_us'], 'us split')"
  done
done
cp /tmp/lib_keep.so erasor_amd/liberasor_hip.so
