#!/bin/bash
# round 3, call bb: chunk records of the outskirts (bounding box + valid count): the VoI split skips chunks outside the circle
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03bb
mkdir -p $OUT
cd $ROOT
timeout 1100 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); r=d['roofline']; print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'], 'split', r['avg_launch_us'], 'us', r['bytes_per_launch'], 'B frac', r['frac'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
for r in 1 2 3; do
  timeout 200 $B > $OUT/rec_$r.json 2> /dev/null; line $OUT/rec_$r.json records
  ERASOR_HIP_NO_OMETA=1 timeout 200 $B > $OUT/norec_$r.json 2> /dev/null; line $OUT/norec_$r.json no_records
done
for r in 1 2; do
  timeout 200 $B --workload large_scale_05 > $OUT/ls05_rec_$r.json 2> /dev/null; line $OUT/ls05_rec_$r.json ls05_records
  ERASOR_HIP_NO_OMETA=1 timeout 200 $B --workload large_scale_05 > $OUT/ls05_norec_$r.json 2> /dev/null; line $OUT/ls05_norec_$r.json ls05_no_records
done
