#!/bin/bash
# parity suite, then A/B of two builds on the same box: tools/gpu_ab_test.sh <tag> a.so b.so ["bench args"]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-ab}
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q --timeout=120 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
bash tools/ab_many.sh "$4" $2 $3
