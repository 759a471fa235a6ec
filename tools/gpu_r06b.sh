#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06b}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log
B="timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --repeats 1 --steps 30"
ERASOR_HIP_OVERLAP=1 ERASOR_HIP_CHAIN_STAMPS=1 $B --chain-batch 3 --lookahead 7 2> $OUT/stamps_ov1_b3.txt > /dev/null
grep "stamps," $OUT/stamps_ov1_b3.txt | tail -14
ERASOR_HIP_OVERLAP=0 ERASOR_HIP_CHAIN_STAMPS=1 $B --chain-batch 3 --lookahead 7 2> $OUT/stamps_ov0_b3.txt > /dev/null
grep "stamps," $OUT/stamps_ov0_b3.txt | tail -8
