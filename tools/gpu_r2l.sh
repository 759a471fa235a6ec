#!/bin/bash
# A/B of the exact-sort partition rewrite: parity first, then timing against the previous build, then the sort's stamps
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02l}
NEW=${2:-variants/wp2.so}
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q --timeout=120 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
bash tools/ab_many.sh "" variants/g4.so $NEW
bash tools/ab_many.sh "--workload large_scale_05" variants/g4.so $NEW
ERASOR_HIP_SORT_STAMPS=1 timeout 150 python bench.py --no-cpu-baseline --steps 8 2>&1 >/dev/null | grep -E "slowest|esort" | tail -8 | cut -c1-420
