#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03at
mkdir -p $OUT
cd $ROOT
python tools/export_cpp_bench.py /tmp/cppbench 40 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for r in 1 2 3 4; do
  rm -rf /tmp/rp_cpp
  timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_cpp -- $ROOT/erasor_amd/erasor_offline_demo --bench /tmp/cppbench 30 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $r:', d['ms_per_callback'], d['ms_per_callback_next_node_announced'], d['ms_per_step_device_resident_two_ahead'])"
  f=$(find /tmp/rp_cpp -name "*kernel_trace.csv" | head -1)
  python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last third of the trace = pass 3
ends=[i for i,r in enumerate(rows) if "k_step_end" in r["Kernel_Name"]]
n=len(ends)
a=ends[n-20]; b=ends[n-1]
seg=rows[a:b]
q=collections.defaultdict(collections.Counter)
for r in seg:
    name=r["Kernel_Name"].split("(")[0].replace("void ","").replace("ek::","")
    q[r["Queue_Id"]][name]+=1
for k,v in q.items(): print("   queue",k, dict(v.most_common(4)), "kernels", sum(v.values()))
t=(int(rows[b]["End_Timestamp"])-int(rows[a]["End_Timestamp"]))/19/1e3
print("   traced us per step in pass 3: %.1f"%t)
PY
done
