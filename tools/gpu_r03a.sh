#!/bin/bash
# round 3, first GPU call: where round 2's build stands, PRESORT timed at last (A/B, twice), host enqueue time per step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03a
mkdir -p $OUT
cd $ROOT
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'], 'split_us', d['roofline']['avg_launch_us'])"; }
for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/base_$r.json 2> $OUT/base_$r.err; line $OUT/base_$r.json base
  ERASOR_HIP_PRESORT=1 timeout 200 python bench.py --no-cpu-baseline --steps 30 > $OUT/presort_$r.json 2> $OUT/presort_$r.err; line $OUT/presort_$r.json presort
done
ERASOR_HIP_HOST_TIMING=1 timeout 200 python bench.py --no-cpu-baseline --steps 8 2> $OUT/host_timing.txt > /dev/null
grep "step host" $OUT/host_timing.txt | head -14
ERASOR_HIP_SORT_STAMPS=1 timeout 200 python bench.py --no-cpu-baseline --steps 6 2>&1 >/dev/null | grep "slowest" | tail -6
timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -24 > $OUT/breakdown_seq05.txt
ERASOR_HIP_PRESORT=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --profile-all 2>&1 >/dev/null | tail -24 > $OUT/breakdown_seq05_presort.txt
grep -E "rgpf|bin_vox|wall" $OUT/breakdown_seq05.txt $OUT/breakdown_seq05_presort.txt
