#!/bin/bash
# round-2 late A/B: gather pieces / tile size variants, pre-converted call arguments, then the per-stage breakdown
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02k
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q --timeout=120 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
bash tools/ab_many.sh "" variants/base.so variants/g1.so variants/g4.so variants/g8.so variants/g4t2k.so
bash tools/ab_many.sh "--workload large_scale_05" variants/base.so variants/g4.so variants/g4t2k.so
bash tools/ab_env.sh "" ERASOR_BENCH_NUMPY_ARGS=1 -
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -22 > $OUT/breakdown_seq05.txt
grep -E "rgpf|bin_vox|srt|voi_bucket|voi_gather|assemble|layout|chunk|bin_stats|voi_split|step_end|wall" $OUT/breakdown_seq05.txt
