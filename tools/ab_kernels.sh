#!/bin/bash
# kernel averages (rocprofv3 --kernel-trace --stats) of the build under _ab_old/ and of the working tree on ONE box, then the bench lines
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
KERNELS=${AB_KERNELS:-"k_voi_gather k_voi_split k_late_gather k_assemble_late k_chunk_scan_one k_srt4 k_revert_bins_srt"}
A="--no-cpu-baseline --no-extra-workloads --no-callback-bench --no-pr-rr --steps 20 --warmup 5"
cd /tmp && export TMPDIR=/tmp
for which in _ab_old . _ab_old .; do
  for wl in ${AB_WORKLOADS:-"--workload=seq05"}; do
    rm -rf /tmp/rp_ab
    timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_ab -- python $ROOT/$which/bench.py $A $wl > /dev/null 2>&1
    f=$(find /tmp/rp_ab -name '*kernel_stats.csv' | head -1)
    echo "== $which $wl"; python -c "
import csv
ks='$KERNELS'.split()
for r in csv.DictReader(open('$f')):
    n=r['Name'].replace('void ','').replace('ek::','').split('(')[0].split('<')[0]
    if n in ks: print('   %-22s %5s launches  avg %7.2f us  min %6.2f  max %7.2f' % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))"
  done
done
cd $ROOT
AB_WORKLOADS=${AB_WORKLOADS:-"--workload=seq05"} tools/ab_old_new.sh | awk '{ if ($1=="old"||$1=="new") print $1, $2, $(NF-2), $(NF-1), $NF; else print }'
