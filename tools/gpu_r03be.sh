#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03be
mkdir -p $OUT
cd $ROOT
( time timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$OUT/bench_driver.json').read().strip().split('\n')[-1])
print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['dtype'][:20], d['vs_baseline'], d['scaling'], d['higher_is_better'])
r=d['roofline']; print({k:r[k] for k in ('bound','achieved','peak','unit','frac','traffic','rocprofv3_kernel_avg_us','avg_launch_us','step_frac')}); print(r['dominant_kernel'])
print(d['cpu_baseline']); print(d['parity_checked_steps'], d['final_map_checked'])
print([(w['workload'], w['value']) for w in d['other_workloads']])
"
