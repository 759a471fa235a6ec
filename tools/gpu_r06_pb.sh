#!/bin/bash
# the per-bin launch after a change to it, old build (_ab_old/) against the working tree on ONE box: a slice of the parity suite, the slowest
# bin's phase stamps of both builds, the kernel's rocprofv3 averages and the bench lines in the order old, new, new, old
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r06pb}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "equal_heights or step_parity or v2 or midway" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
for which in _ab_old .; do for w in seq05 large_scale_05; do
  echo "== stamps $which $w"
  ERASOR_HIP_SORT_STAMPS=1 timeout 300 python $ROOT/$which/bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --steps 20 --warmup 5 --repeats 1 --workload $w 2>&1 >/dev/null | grep "last per-bin workgroup\|slowest reverted bin\|slowest R-GPF bin, 10 ns\|^.reverted bins" | tail -8
done; done
AB_KERNELS="k_revert_bins_srt k_assemble_late" AB_WORKLOADS="--workload=seq05 --workload=large_scale_05" tools/ab_kernels.sh 2>&1 | tee $OUT/ab.txt
