#!/bin/bash
# round 3, call e: async API, bench self-verification, config 3 on one GPU: sequences one after the other / interleaved / threads
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03e
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().split('\n')[-1])
print('default', d['value'], d['ms_per_step'], 'parity steps', d['parity_checked_steps'], d['final_map_checked'], '|', d['parity'])
print(' roofline', d['roofline']['bound'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['dominant_kernel'])
print(' cpu', d['cpu_baseline']['value'], d['cpu_port']['value'])"
tail -3 $OUT/bench_default.err
for q in 4 8; do for il in off async threads; do
  GPU_MAX_HW_QUEUES=$q timeout 400 python bench.py --mode seq-per-gpu --interleave $il --steps 20 --warmup 3 --no-cpu-baseline > $OUT/seq_${il}_q$q.json 2> $OUT/seq_${il}_q$q.err
  python -c "
import json; d=json.loads(open('$OUT/seq_${il}_q$q.json').read().strip().split('\n')[-1])
print('seq-per-gpu queues $q interleave $il:', d['value'], 'scans/s', d['ms_per_step'], 'ms/step')" || tail -3 $OUT/seq_${il}_q$q.err
done; done
