#!/bin/bash
# first GPU pass of round 2: GPU tests (incl. the reference-made vectors), smoke, the bench workloads
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_seq05.json 2> $OUT/bench_seq05.err; echo "bench rc=$?"
timeout 600 python bench.py --workload large_scale_05 --steps 10 --cpu-steps 2 > $OUT/bench_ls05.json 2> $OUT/bench_ls05.err; echo "bench ls rc=$?"
timeout 600 python bench.py --workload large_scale_05 --large-scale-mode on --steps 10 --no-cpu-baseline > $OUT/bench_ls05_submap.json 2> $OUT/bench_ls05_submap.err; echo "bench ls submap rc=$?"
timeout 600 python bench.py --workload ouster128 --steps 10 --cpu-steps 2 > $OUT/bench_ouster.json 2> $OUT/bench_ouster.err; echo "bench ouster rc=$?"
timeout 600 python bench.py --workload seq05_yaml --steps 10 --no-cpu-baseline > $OUT/bench_seq05_yaml.json 2> $OUT/bench_seq05_yaml.err; echo "bench yaml rc=$?"
ERASOR_BENCH_BACKEND=gloo ERASOR_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "bench 2rank rc=$?"
timeout 900 python bench.py --mode seq-per-gpu --steps 5 --eval --street-length 400 --streets 2 > $OUT/bench_seqpergpu.json 2> $OUT/bench_seqpergpu.err; echo "bench seqpergpu rc=$?"
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; for f in $OUT/bench_*.json; do echo $f; head -c 600 $f; echo; done
