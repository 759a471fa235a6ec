#!/bin/bash
# the merged early write-back (k_assemble_early) against the build under _ab_old/ on ONE box: the whole GPU suite first, then kernel averages
# and bench lines old, new, new, old
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r06ea}
mkdir -p $OUT
cd $ROOT
timeout 1700 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $OUT/pytest_gpu.log | tail -2
AB_KERNELS="k_revert_bins_srt k_assemble_late k_assemble_early k_srt4 k_assemble_map k_voi_split k_voi_gather k_chunk_scan_one k_late_gather" AB_WORKLOADS="--workload=seq05 --workload=large_scale_05" tools/ab_kernels.sh 2>&1 | tee $OUT/ab.txt
