"""From a rocprofv3 --kernel-trace CSV: the main stream's kernels of one steady-state step (start offsets, durations, gaps)."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [r for r in rows if "k_step_end" in r["Kernel_Name"]]
step = int(sys.argv[2]) if len(sys.argv) > 2 else -3  # which k_step_end closes the step to print
q = ends[step]["Queue_Id"]
main = [r for r in rows if r["Queue_Id"] == q]
# last complete step on that queue: from the kernel after the previous k_step_end to the next k_step_end
idx = [i for i, r in enumerate(main) if "k_step_end" in r["Kernel_Name"]]
a, b = idx[step - 1] + 1, idx[step]
t0 = int(main[a]["Start_Timestamp"])
prev_end = int(main[a - 1]["End_Timestamp"])
print("gap since the previous step's k_step_end: %.1f us" % ((t0 - prev_end) / 1e3))
last = t0
for r in main[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  +%7.1f  gap %5.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (s - last) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:40]))
    last = e
print("step span on the main stream: %.1f us" % ((last - t0) / 1e3))
# kernels of OTHER queues that ran inside the step's window and are part of the map chain (the outskirts part of the next VoI split)
for r in rows:
    if r["Queue_Id"] != q and "k_voi_split" in r["Kernel_Name"]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 <= s <= last:
            print("  side stream: +%7.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:40]))
