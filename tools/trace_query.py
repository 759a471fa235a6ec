"""From a rocprofv3 --kernel-trace CSV: one steady-state QUERY chain (k_query_begin .. its k_bin_stats) on its stream -- start offsets,
durations, gaps -- and how long the chain took on the wall against the sum of its kernels."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
begins = [r for r in rows if "k_query_begin" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -4
b = begins[which]
q = b["Queue_Id"]
chain = [r for r in rows if r["Queue_Id"] == q and int(r["Start_Timestamp"]) >= int(b["Start_Timestamp"])]
out = []
for r in chain:
    if out and "k_query_begin" in r["Kernel_Name"]:
        break
    out.append(r)
t0 = int(out[0]["Start_Timestamp"])
last = t0
agg = {}
for r in out:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("ek::", "")[:34]
    a = agg.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += max(0, s - last) / 1e3
    last = e
print("query chain on queue %s: %d launches, wall %.1f us, kernels %.1f us, gaps %.1f us" %
      (q, len(out), (last - t0) / 1e3, sum(a[1] for a in agg.values()), sum(a[2] for a in agg.values())))
for name, a in agg.items():
    print("  %-34s x%-3d  %7.1f us   (gaps in front %5.1f us)" % (name, a[0], a[1], a[2]))
