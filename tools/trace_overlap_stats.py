"""From a rocprofv3 --kernel-trace CSV of bench.py (look-ahead pass, then the pass without look-ahead): per kernel, the average duration of
the launches that ran ALONE against those that shared the chip with a kernel of another queue -- what concurrency costs each kernel."""
import csv, glob, sys, bisect
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("ek::", "").replace("void ", "")[:30]) for r in rows]
ev.sort()
starts = [e[0] for e in ev]
stat = {}
for i, (s, e, q, name) in enumerate(ev):
    # overlapped time with kernels of other queues
    ov = 0
    j = i - 1
    while j >= 0 and ev[j][0] > s - 2_000_000:  # look back 2 ms
        if ev[j][2] != q and ev[j][1] > s:
            ov += min(e, ev[j][1]) - s
        j -= 1
    j = i + 1
    while j < len(ev) and ev[j][0] < e:
        if ev[j][2] != q:
            ov += min(e, ev[j][1]) - ev[j][0]
        j += 1
    frac = ov / max(1, e - s)
    st = stat.setdefault(name, {"alone": [], "shared": []})
    (st["shared"] if frac > 0.5 else st["alone"] if frac < 0.05 else st.setdefault("mixed", [])).append((e - s) / 1e3)
print("%-32s %8s %10s %8s %10s %7s" % ("kernel", "alone n", "avg us", "shared n", "avg us", "ratio"))
tot_a = tot_s = 0.0
for name, st in sorted(stat.items(), key=lambda kv: -sum(kv[1]["shared"] + kv[1]["alone"])):
    a, s = st["alone"], st["shared"]
    if len(a) < 3 or len(s) < 3:
        continue
    ma, ms = sum(a) / len(a), sum(s) / len(s)
    print("%-32s %8d %10.2f %8d %10.2f %7.2f" % (name, len(a), ma, len(s), ms, ms / ma))
