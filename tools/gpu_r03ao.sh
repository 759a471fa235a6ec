#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ao
mkdir -p $OUT
cd $ROOT
python tools/export_cpp_bench.py /tmp/cppbench 40 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_cpp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_cpp -- $ROOT/erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 > $OUT/cpp_traced.json 2>/dev/null
cd $ROOT
f=$(find /tmp/rp_cpp -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if 'ek::' in r['Name']]
steps=max(int(r['Calls']) for r in rows if 'k_step_end' in r['Name'])
print('steps',steps)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
    print('  %-30s calls %6s calls/step %5.1f avg %8.1f us total/step %7.1f us'%(r['Name'].split('(')[0].replace('void ','').replace('ek::','')[:30], r['Calls'], int(r['Calls'])/steps, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/steps/1e3))
PY
tail -c 400 $OUT/cpp_traced.json
