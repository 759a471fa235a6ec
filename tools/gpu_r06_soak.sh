#!/bin/bash
# round 6 soak: the seeded random walk over caller behaviour (deep look-ahead, chains held back / dropped / claimed early) and long
# sequences with every step replayed on the oracle by bench.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r06soak}
mkdir -p $OUT
cd $ROOT
( time ERASOR_FUZZ_SEEDS=${2:-240} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k random_operation -p no:cacheprovider ) > $OUT/fuzz.log 2>&1; tail -4 $OUT/fuzz.log
( time ERASOR_FUZZ_SEEDS=120 ERASOR_HIP_OVERLAP= timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k random_operation -p no:cacheprovider ) > $OUT/fuzz_auto.log 2>&1; tail -3 $OUT/fuzz_auto.log
ERASOR_HIP_OVERLAP=1 timeout 900 python bench.py --steps 100 --warmup 5 --repeats 7 --no-extra-workloads --no-pr-rr --no-callback-bench --cpu-seconds 200 > $OUT/long_seq05.json 2> $OUT/long_seq05.err
python -c "
import json; d=json.loads(open('$OUT/long_seq05.json').read().strip().split(chr(10))[-1]); print('seq05 forced overlap, 705 steps:', d['ms_per_step'], d['parity_checked_steps'], d['final_map_checked'], d['overlapped_steps'], d['shared_chain_launches'])"
timeout 900 python bench.py --workload large_scale_05 --steps 40 --warmup 5 --repeats 3 --no-extra-workloads --no-pr-rr --no-callback-bench --cpu-seconds 300 > $OUT/long_ls05.json 2> $OUT/long_ls05.err
python -c "
import json; d=json.loads(open('$OUT/long_ls05.json').read().strip().split(chr(10))[-1]); print('ls05 auto, 125 steps:', d['ms_per_step'], d['parity_checked_steps'], d['final_map_checked'], d['overlapped_steps'], d['shared_chain_launches'])"
