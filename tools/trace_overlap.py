"""Summarise a rocprofv3 --kernel-trace CSV: per-step critical path, per-queue busy time, overlap between queues."""
import csv, glob, sys
from collections import defaultdict

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into steps at k_step_begin
steps, cur = [], []
for r in rows:
    if "k_step_begin" in r["Kernel_Name"] and cur:
        steps.append(cur)
        cur = []
    cur.append(r)
steps.append(cur)
full = [s for s in steps if any("k_step_end" in r["Kernel_Name"] for r in s) and any("k_voi_split" in r["Kernel_Name"] for r in s)]
print("steps:", len(full))
s = full[-2]
t0 = int(s[0]["Start_Timestamp"])
print("last full step: %d kernels, span %.1f us" % (len(s), (int(s[-1]["End_Timestamp"]) - t0) / 1e3))
for r in s:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("  q%-3s %8.1f -> %8.1f  (%6.1f us) grid %-9s %s" % (r["Queue_Id"], a / 1e3, b / 1e3, (b - a) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"][:60]))
