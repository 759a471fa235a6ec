#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02o}
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q --timeout=120 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
bash tools/ab_env.sh "" ERASOR_BENCH_NO_POSE_AHEAD=1 -
bash tools/ab_env.sh "--workload ouster128" ERASOR_BENCH_NO_POSE_AHEAD=1 -
