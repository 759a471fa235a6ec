#!/bin/bash
# quick look at the per-bin launch after a change to it: a slice of the parity suite, the kernel's duration under rocprofv3 on both bench workloads,
# the slowest bin's stamps, the driver's bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r06zs}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "equal_heights or step_parity" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
B="timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --steps 20 --warmup 5"
cd /tmp && export TMPDIR=/tmp
for w in "" "--workload large_scale_05"; do
  rm -rf /tmp/rp_zs
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_zs -- $B $w > /dev/null 2>&1
  echo "== rocprofv3 $w"; f=$(find /tmp/rp_zs -name '*kernel_stats.csv' | head -1); python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if any(k in r['Name'] for k in ('k_revert_bins_srt','k_esort_final_b','k_voi_gather','k_srt4')): print(r['Name'][:28], r['Calls'], 'avg', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])"
done
cd $ROOT
echo "== stamps"; ERASOR_HIP_SORT_STAMPS=1 $B --repeats 1 2>&1 | grep "last per-bin workgroup\|slowest reverted bin\|10 ns ticks\] key\|^.reverted bins" | tail -48
echo "== line"; $B | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['ms_per_step_all'], d.get('overlapped_steps'))"
$B --workload large_scale_05 --repeats 7 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['ms_per_step_all'])"
