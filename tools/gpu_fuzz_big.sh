#!/bin/bash
# a long soak of the seeded random walk over caller behaviour (tests/test_gpu_parity.py -k random_operation): overlap forced, then the handle deciding
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-fuzzbig}
N=${2:-1000}
mkdir -p $OUT
cd $ROOT
( time ERASOR_FUZZ_SEEDS=$N ERASOR_HIP_OVERLAP=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k random_operation -p no:cacheprovider ) > $OUT/fuzz.log 2>&1; grep -E "passed|failed" $OUT/fuzz.log | tail -1
( time ERASOR_FUZZ_SEEDS=$((N/2)) ERASOR_HIP_OVERLAP= timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k random_operation -p no:cacheprovider ) > $OUT/fuzz_auto.log 2>&1; grep -E "passed|failed" $OUT/fuzz_auto.log | tail -1
( time ERASOR_FUZZ_SEEDS=$((N/4)) ERASOR_HIP_OVERLAP=0 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k random_operation -p no:cacheprovider ) > $OUT/fuzz_plain.log 2>&1; grep -E "passed|failed" $OUT/fuzz_plain.log | tail -1
