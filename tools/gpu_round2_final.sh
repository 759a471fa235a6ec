#!/bin/bash
# round-2 closing run: the GPU suite, the smoke, then the evidence for profiles/ (config 2)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02z
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q --timeout=120 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
bash tools/collect_profiles.sh r02z_seq05 > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log
cat gpurun_out/profiles_r02z_seq05/bench.json | cut -c1-1500
