#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03z
mkdir -p $OUT
cd $ROOT
timeout 100 python tools/dbg_golden.py 2>&1 | tail -3
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'], 'roof', d['roofline']['avg_launch_us'], d['roofline']['frac'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so
  ERASOR_HIP_NO_SPLIT_PARTS=1 timeout 200 $B > $OUT/noparts_$r.json 2> /dev/null; line $OUT/noparts_$r.json no_parts
  for g in 16 32 64 128; do
    ERASOR_HIP_OSPLIT_GRID=$g timeout 200 $B > $OUT/g${g}_$r.json 2> /dev/null; line $OUT/g${g}_$r.json grid_$g
  done
  ERASOR_HIP_OSPLIT_GRID=32 ERASOR_HIP_OSPLIT_AT=2 timeout 200 $B > $OUT/g32s_$r.json 2> /dev/null; line $OUT/g32s_$r.json grid_32_at_srt
  cp variants/dbg_oldO.so erasor_amd/liberasor_hip.so
  ERASOR_HIP_NO_SPLIT_PARTS=1 timeout 200 $B > $OUT/oldO_$r.json 2> /dev/null; line $OUT/oldO_$r.json no_parts_oldOgather
  cp variants/prev.so erasor_amd/liberasor_hip.so
  timeout 200 $B > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
