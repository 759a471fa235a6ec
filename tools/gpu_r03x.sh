#!/bin/bash
# round 3, call x: the next step's VoI split in two parts (outskirts part forked behind the gather, on a side stream), the gather's
# outskirts chunks compacted across tiles
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03x
mkdir -p $OUT
cd $ROOT
timeout 1100 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'], 'roof', d['roofline']['avg_launch_us'], d['roofline']['frac'])"; }
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so
  timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 > $OUT/new_$r.json 2> /dev/null; line $OUT/new_$r.json fork_at_gather
  ERASOR_HIP_OSPLIT_AT=1 timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 > $OUT/at1_$r.json 2> /dev/null; line $OUT/at1_$r.json fork_at_bucketing
  ERASOR_HIP_OSPLIT_AT=2 timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 > $OUT/at2_$r.json 2> /dev/null; line $OUT/at2_$r.json fork_at_srt
  ERASOR_HIP_NO_SPLIT_PARTS=1 timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 > $OUT/noparts_$r.json 2> /dev/null; line $OUT/noparts_$r.json no_parts
  cp variants/prev.so erasor_amd/liberasor_hip.so
  timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 > $OUT/prev_$r.json 2> /dev/null; line $OUT/prev_$r.json previous
done
cp /tmp/keep.so erasor_amd/liberasor_hip.so
timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --workload large_scale_05 --steps 20 --warmup 5 > $OUT/ls05_new.json 2> /dev/null; line $OUT/ls05_new.json ls05_new
ERASOR_HIP_NO_SPLIT_PARTS=1 timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --workload large_scale_05 --steps 20 --warmup 5 > $OUT/ls05_noparts.json 2> /dev/null; line $OUT/ls05_noparts.json ls05_noparts
bash tools/gpu_trace.sh r03x 2>&1 | tail -16
