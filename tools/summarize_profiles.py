"""Condense rocprofv3 outputs into the small files kept under profiles/ (see tools/collect_profiles.sh)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

stats_dir, fetch_dir, write_dir, out = sys.argv[1:5]


def find(d, pat):
    f = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return f[-1] if f else None


ks = find(stats_dir, "*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(out, "rocprofv3_kernel_stats.csv"))


def pmc(d, counter):
    f = find(d, "*counter_collection.csv")
    per = defaultdict(list)
    if not f:
        return {}
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != counter:
            continue
        name = row["Kernel_Name"].split("(")[0]
        per[name].append(float(row["Counter_Value"]))
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
    return {k: {"dispatches": len(v), "mean_KiB": round(sum(v) / len(v), 1), "max_KiB": round(max(v), 1), "total_KiB": round(sum(v), 1)}
            for k, v in sorted(per.items())}


res = {"FETCH_SIZE": pmc(fetch_dir, "FETCH_SIZE"), "WRITE_SIZE": pmc(write_dir, "WRITE_SIZE")}
json.dump(res, open(os.path.join(out, "pmc_fetch_write_size.json"), "w"), indent=1)

# k_voi_split traffic per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 128-B
# requests as 64 B: x2; calibrated on k_store_outskirts, a pure 16 B/pt stream, when it is in the trace)
fs = res["FETCH_SIZE"].get("ek::k_voi_split")
ws = res["WRITE_SIZE"].get("ek::k_voi_split")
if fs:
    latest = {
        "kernel": "k_voi_split",
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/collect_profiles.sh)",
        "fetch_size_KiB_per_launch": fs["mean_KiB"],
        "write_size_KiB_per_launch": ws["mean_KiB"] if ws else None,
        "gfx950_correction": "FETCH_SIZE x2 (128-B requests tallied at 64 B)",
        "traffic_bytes_per_launch": int(fs["mean_KiB"] * 1024 * 2 + (ws["mean_KiB"] * 1024 if ws else 0)),
    }
    # every kernel's corrected traffic per launch (bench.py reports the figure of the kernel that tops the GPU-time table)
    latest["per_kernel_traffic_bytes"] = {
        k.replace("void ", "").replace("ek::", "").split("<")[0]: int(v["mean_KiB"] * 1024 * 2 + (res["WRITE_SIZE"].get(k, {}).get("mean_KiB", 0.0)) * 1024)
        for k, v in res["FETCH_SIZE"].items() if "ek::" in k}
    # the WHOLE step's corrected traffic: every kernel's total over the pass divided by the steps of the pass (one k_voi_gather each)
    steps = max(res["FETCH_SIZE"].get("ek::k_voi_gather", {}).get("dispatches", 0), 1)
    tot = 0.0
    tot_steps_only = 0.0
    per_step = {}
    for k, v in res["FETCH_SIZE"].items():
        if "ek::" not in k:
            continue
        b = v["total_KiB"] * 1024 * 2 + res["WRITE_SIZE"].get(k, {}).get("total_KiB", 0.0) * 1024
        per_step[k.replace("void ", "").replace("ek::", "").split("<")[0]] = int(b / steps)
        tot += b
        if v.get("dispatches", 0) >= steps:  # (a kernel of every step; the others belong to set_map / the pass's set-up, once)
            tot_steps_only += b
    latest["step_traffic_bytes"] = int(tot / steps)
    latest["step_traffic_bytes_without_setup_kernels"] = int(tot_steps_only / steps)
    latest["step_traffic_steps"] = steps
    latest["per_kernel_traffic_bytes_per_step"] = dict(sorted(per_step.items(), key=lambda kv: -kv[1]))
    cal = res["FETCH_SIZE"].get("ek::k_store_outskirts")
    if cal:
        latest["calibration_k_store_outskirts_fetch_KiB"] = cal["max_KiB"]
    json.dump(latest, open(os.path.join(out, "pmc_latest.json"), "w"), indent=1)
# which build these passes describe: bench.py prints the committed counters only while the device sources are the same
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hh = hashlib.sha256()
for f in ("erasor_hip.hip", "kernels.hip.h", "revert_bins.hip.h", "exact_sort.hip.h", "exact_sort_core.h"):
    hh.update(open(os.path.join(root, "erasor_amd", "csrc", f), "rb").read())
json.dump({"device_source_sha16": hh.hexdigest()[:16]}, open(os.path.join(out, "latest_meta.json"), "w"))
print(json.dumps({"kernel_stats": ks, "voi_split_fetch": fs, "voi_split_write": ws}))
