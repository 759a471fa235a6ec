"""From a rocprofv3 --kernel-trace CSV: every launch between two per-bin launches (k_revert_bins_srt) of the steady state, all queues --
start offset, queue, duration, kernel.  Round 5: what runs beside what when steps overlap.
  python tools/trace_window.py <dir> [first_step [n_steps [skip_query_chain]]]"""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rev = [r for r in rows if "k_revert_bins" in r["Kernel_Name"]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 9
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
skipq = len(sys.argv) > 4 and sys.argv[4] != "0"
t0, t1 = int(rev[first]["Start_Timestamp"]), int(rev[first + n]["Start_Timestamp"])
qs = {}
QCH = ("k_esort", "k_bbox", "k_voxel_keys", "k_run_", "k_centroids", "k_query_nn", "k_qb_", "k_query_begin", "k_bin_stats(")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s >= t1:
        continue
    name = r["Kernel_Name"].replace("void ", "").replace("ek::", "").split("(")[0][:44]
    full = r["Kernel_Name"].replace("ek::", "")
    if skipq and any(k in full for k in QCH) and "k_bin_stats_srt" not in full:
        continue
    q = qs.setdefault(r["Queue_Id"], len(qs))
    print("+%8.1f  q%d %s dur %6.1f  %s" % ((s - t0) / 1e3, q, "    " * q, (e - s) / 1e3, name))
print("period: %.1f us per step over %d steps" % ((t1 - t0) / 1e3 / n, n))
pers = [(int(rev[i + 1]["Start_Timestamp"]) - int(rev[i]["Start_Timestamp"])) / 1e3 for i in range(len(rev) - 1)]
print("per-bin launch to per-bin launch, all steps (us):", " ".join("%.0f" % p for p in pers))
