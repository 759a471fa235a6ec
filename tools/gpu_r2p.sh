#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02p}
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q --timeout=120 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/collect_profiles.sh r02zz_seq05 > $OUT/collect.log 2>&1; tail -2 $OUT/collect.log
cut -c1-330 gpurun_out/profiles_r02zz_seq05/bench.json
