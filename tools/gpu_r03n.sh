#!/bin/bash
# round 3, call n: fewer wide sort levels (host launches), two sequences interleaved on one GPU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03n
mkdir -p $OUT
cd $ROOT
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for r in 1 2; do
  for v in keep slack3 slack2; do
    if [ $v = keep ]; then cp /tmp/keep.so erasor_amd/liberasor_hip.so; else cp variants/$v.so erasor_amd/liberasor_hip.so; fi
    timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/${v}_$r.json 2> /dev/null; line $OUT/${v}_$r.json $v
  done
done
cp variants/slack2.so erasor_amd/liberasor_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact_std_sort or full_size or voxelize or whole_map" > $OUT/pytest_slack2.log 2>&1; echo "pytest(slack2) rc=$?"; tail -2 $OUT/pytest_slack2.log
cp /tmp/keep.so erasor_amd/liberasor_hip.so
for q in 8; do for n in 2 3 5; do for il in off async; do
  GPU_MAX_HW_QUEUES=$q timeout 400 python bench.py --mode seq-per-gpu --seqs $n --interleave $il --steps 20 --warmup 3 --no-cpu-baseline > $OUT/seq${n}_$il.json 2> $OUT/seq${n}_$il.err
  line $OUT/seq${n}_$il.json "seqs=$n interleave=$il q$q"
done; done; done
