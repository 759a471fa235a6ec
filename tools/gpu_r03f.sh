#!/bin/bash
# round 3, call f: the query chain as two hipGraphs per side (recorded by LAUNCH, patched per scan): host time, single sequence, config 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03f
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/graph_$r.json 2> $OUT/graph_$r.err; line $OUT/graph_$r.json graph
  ERASOR_HIP_NO_GRAPH=1 timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/nograph_$r.json 2> $OUT/nograph_$r.err; line $OUT/nograph_$r.json nograph
done
ERASOR_HIP_HOST_TIMING=1 timeout 200 python bench.py --no-cpu-baseline --steps 8 2> $OUT/host_timing.txt > /dev/null
grep "step host" $OUT/host_timing.txt | sed -n 20,30p
grep -A6 "step host" $OUT/host_timing.txt | sed -n 100,114p
for q in 4 8; do for g in 0 1; do
  if [ $g = 0 ]; then export ERASOR_HIP_NO_GRAPH=1; else unset ERASOR_HIP_NO_GRAPH; fi
  GPU_MAX_HW_QUEUES=$q timeout 400 python bench.py --mode seq-per-gpu --interleave async --steps 20 --warmup 3 --no-cpu-baseline > $OUT/seq_async_q${q}_g$g.json 2> $OUT/seq_async_q${q}_g$g.err
  python -c "
import json; d=json.loads(open('$OUT/seq_async_q${q}_g$g.json').read().strip().split('\n')[-1])
print('seq-per-gpu async queues $q graph $g:', d['value'], 'scans/s', d['ms_per_step'], 'ms/step')" || tail -3 $OUT/seq_async_q${q}_g$g.err
done; done
unset ERASOR_HIP_NO_GRAPH
timeout 300 python bench.py --steps 20 --warmup 5 --workload large_scale_05 > $OUT/bench_ls05.json 2> $OUT/bench_ls05.err; echo "ls05 rc=$?"
python -c "
import json; d=json.loads(open('$OUT/bench_ls05.json').read().strip().split('\n')[-1])
print('ls05', d['value'], d['ms_per_step'], 'parity', d['parity_checked_steps'], d['final_map_checked'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['step_frac'])"
