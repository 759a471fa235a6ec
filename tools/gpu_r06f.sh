#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06f}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().split("\n")[-1])
print("ms_per_step", d["ms_per_step"], d["ms_per_step_all"], "first", d["ms_per_step_first_pass"], "value", d["value"])
print("overlap", d["overlapped_steps"], d["shared_chain_launches"], "parity steps", d["parity_checked_steps"], d["final_map_checked"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], "| refsrc", (d.get("cpu_reference_sources") or {}).get("value"))
print("nolook", d["ms_per_step_without_lookahead"], "callback", d.get("callback_path"))
for w in d.get("other_workloads", []):
    print(w.get("workload"), w.get("ms_per_step"), w.get("ms_per_step_min"), w.get("ms_per_step_max"), w.get("overlapped_steps"), str(w.get("label"))[:50])
PY
