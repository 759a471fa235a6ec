#!/bin/bash
# round 3, call b: block_esort vs block_esort_sync on the device (one workgroup, cycles), launch / graph-launch host cost
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03b
mkdir -p $OUT
cd $ROOT
timeout 120 tools/esort_bench 2>&1 | tee $OUT/esort_bench.txt
timeout 120 tools/launch_rate 2>&1 | grep -v "exit kernel" | tee $OUT/launch_rate.txt
