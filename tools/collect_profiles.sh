#!/bin/bash
# Run ON THE GPU BOX (through gpurun): collects the evidence committed under profiles/.
#   tools/collect_profiles.sh <tag> [bench.py args]     -> gpurun_out/profiles_<tag>/...
# Passes (each its own process; counters never share a run with tracing, as the pool requires):
#   1. plain bench.py (+ per-kernel HIP-event breakdown)
#   2. timeout 240 rocprofv3 --kernel-trace --stats   (the SAME command as pass 1: seven 20-step passes, so that the averages describe the same mix of steps)
#   3. timeout 240 rocprofv3 --pmc FETCH_SIZE        4. timeout 240 rocprofv3 --pmc WRITE_SIZE
TAG=${1:-run}
shift
EXTRA="$@"   # e.g. --workload large_scale_05
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="timeout 300 python $ROOT/bench.py --steps 20 --warmup 5 --no-extra-workloads --no-pr-rr --no-callback-bench $EXTRA"
$BENCH > $OUT/bench.json 2> $OUT/bench.stderr
$BENCH --no-cpu-baseline --profile-all --repeats 1 > /dev/null 2> $OUT/bench_kernel_breakdown.txt
rm -rf /tmp/rp_stats /tmp/rp_fetch /tmp/rp_write
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -- $BENCH --no-cpu-baseline > $OUT/bench_under_rocprofv3.json 2> /dev/null
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rp_fetch -- $BENCH --steps 6 --repeats 1 --no-cpu-baseline > /dev/null 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rp_write -- $BENCH --steps 6 --repeats 1 --no-cpu-baseline > /dev/null 2>&1
cd $ROOT && python tools/summarize_profiles.py /tmp/rp_stats /tmp/rp_fetch /tmp/rp_write $OUT
ls -la $OUT
# the files bench.py reads (roofline.traffic, rocprofv3_kernel_avg_us, dominant_kernel): copies of this tag, with the build they describe
SUF=""; case "$EXTRA" in *large_scale_05*) SUF="_large_scale_05";; esac
cp $OUT/rocprofv3_kernel_stats.csv $ROOT/gpurun_out/kernel_stats_latest$SUF.csv 2>/dev/null
cp $OUT/pmc_latest.json $ROOT/gpurun_out/pmc_latest$SUF.json 2>/dev/null
cp $OUT/latest_meta.json $ROOT/gpurun_out/latest_meta$SUF.json 2>/dev/null
