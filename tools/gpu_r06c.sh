#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06c}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
for q in "" 4 8 16; do
  if [ -z "$q" ]; then env -u GPU_MAX_HW_QUEUES tools/queue_probe 8 400 64 0 > $OUT/queue_probe_unset.txt; else GPU_MAX_HW_QUEUES=$q tools/queue_probe 8 400 64 0 > $OUT/queue_probe_$q.txt; fi
done
GPU_MAX_HW_QUEUES=16 tools/queue_probe 8 400 64 1 > $OUT/queue_probe_16_idle.txt
cat $OUT/queue_probe_*.txt | cut -c1-420
