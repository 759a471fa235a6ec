#!/bin/bash
# quick GPU check: parity tests (-x), then the headline workloads (clean timing), then the slowest-bin stamps
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-quick}
SKIPTEST=${2:-}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -z "$SKIPTEST" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
fi
for wl in seq05 large_scale_05; do
  timeout 300 python bench.py --no-cpu-baseline --steps 30 --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  python -c "import sys,json; d=json.loads(open('$OUT/bench_$wl.json').read().strip().split('\n')[-1]); print('$wl', d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'], 'split_us', d['roofline']['avg_launch_us'], 'step_frac', d['roofline']['step_frac'])"
done
for wl in seq05 large_scale_05; do
  ERASOR_HIP_SORT_STAMPS=1 timeout 300 python bench.py --no-cpu-baseline --steps 6 --workload $wl 2>&1 >/dev/null | grep "slowest" | tail -4
done
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -22 > $OUT/breakdown_seq05.txt
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all --workload large_scale_05 2>&1 >/dev/null | tail -22 > $OUT/breakdown_ls05.txt
grep -E "rgpf|bin_vox|srt|voi_bucket|voi_gather|assemble|layout|chunk|bin_stats|voi_split|step_end|wall" $OUT/breakdown_seq05.txt $OUT/breakdown_ls05.txt
