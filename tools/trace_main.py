"""From a rocprofv3 --kernel-trace CSV: the main stream's kernels of one steady-state step (start offsets, durations, gaps).
A step is printed from its chunk scan to the launch in front of the next step's chunk scan (round 4: the step's end may ride in the
NEXT step's VoI split, which is launched ahead -- so that split closes the step it is printed with)."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
opens = [r for r in rows if "k_chunk_scan_one" in r["Kernel_Name"] or "k_chunk_scan_top" in r["Kernel_Name"]]
step = int(sys.argv[2]) if len(sys.argv) > 2 else -3  # which step to print
q = opens[step]["Queue_Id"]
main = [r for r in rows if r["Queue_Id"] == q]
idx = [i for i, r in enumerate(main) if "k_chunk_scan_one" in r["Kernel_Name"] or "k_chunk_scan_top" in r["Kernel_Name"]]
a, b = idx[step], idx[step + 1] - 1
t0 = int(main[a]["Start_Timestamp"])
prev_end = int(main[a - 1]["End_Timestamp"])
print("gap between the launch in front (the VoI split, launched ahead) and the chunk scan: %.1f us" % ((t0 - prev_end) / 1e3))
last = t0
for r in main[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  +%7.1f  gap %5.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (s - last) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:40]))
    last = e
print("step span on the main stream (chunk scan .. the launch that ends the step): %.1f us" % ((last - t0) / 1e3))
t1 = int(main[idx[step + 1]]["Start_Timestamp"])
print("chunk scan to chunk scan: %.1f us" % ((t1 - t0) / 1e3))
# kernels of the OTHER queues inside the step's window: how busy the query chains keep the chip meanwhile
other = [r for r in rows if r["Queue_Id"] != q and t0 <= int(r["Start_Timestamp"]) <= last]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in other)
print("other queues inside the window: %d launches, %.1f us of kernel time" % (len(other), busy / 1e3))
