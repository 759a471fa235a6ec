#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02n}
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q --timeout=120 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
bash tools/ab_many.sh "" ${2:-variants/wp2.so} ${3:-variants/runs2.so}
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -24 > $OUT/breakdown_seq05.txt
grep -E "q_|wall" $OUT/breakdown_seq05.txt
