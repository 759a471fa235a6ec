#!/bin/bash
# round 3, call al: where the wide levels of the scan sort stop (ESORT_WIDE_MIN 2049 / 4097 / 8193: k_esort_mid takes over earlier)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03al
mkdir -p $OUT
cd $ROOT
line() { python -c "import sys,json; d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'], 'nolook', d['ms_per_step_without_lookahead'])"; }
B="python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5"
cp erasor_amd/liberasor_hip.so /tmp/keep.so
for v in wmin8193 wmin4097 wmin8193s3; do
  cp variants/$v.so erasor_amd/liberasor_hip.so
  timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact_std_sort or heapsort_fallback or long_segments or kitti_like" > $OUT/pytest_$v.log 2>&1; echo "$v pytest rc=$?"; tail -1 $OUT/pytest_$v.log
done
for r in 1 2 3; do
  cp /tmp/keep.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/base_$r.json 2> /dev/null; line $OUT/base_$r.json wide_min_2049
  for v in wmin4097 wmin8193 wmin8193s3; do
    cp variants/$v.so erasor_amd/liberasor_hip.so; timeout 200 $B > $OUT/${v}_$r.json 2> /dev/null; line $OUT/${v}_$r.json $v
  done
done
cp variants/wmin8193.so erasor_amd/liberasor_hip.so
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --profile-all 2>&1 >/dev/null | tail -26 | grep -E "q_esort|wall"
cp /tmp/keep.so erasor_amd/liberasor_hip.so
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --profile-all 2>&1 >/dev/null | tail -26 | grep -E "q_esort|wall"
