#!/usr/bin/env python3
"""bench.py — ERASOR per-scan hot path on MI355X.

A "step" is one pass of the hot path (query voxelise -> fetch_VoI -> R-POD -> Scan Ratio Test -> R-GPF ->
map write-back; reference OfflineMapUpdater.cpp:237-294) over one synthetic 120 k-point scan against a
~10 M-point map that is resident in HBM.  Inputs (map and scans) are in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
N > 1: launched by torch.distributed.run, one rank per GPU.  Rank 0 builds the map and RCCL-broadcasts it
(xGMI); every rank then processes its own scans against its own replica — no data-path collective, weak scaling.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the roofline and cpu_baseline definitions).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_workload(args, rank):
    """KITTI-05-shaped synthetic world; ~10 M-pt map, 120 k-pt scans, 20 rings x 108 sectors."""
    from erasor_amd import synth
    w = synth.World(seed=20210305 + 5, length=args.street_length, n_streets=args.streets, street_gap=50.0, n_moving=10, n_peds=6)
    lidar = synth.Lidar.hdl64(args.az_steps)
    return w, lidar


def make_params(args):
    import erasor_amd
    from erasor_amd import synth
    p = erasor_amd.params_default()
    synth.apply_params(p, "05")           # config/seq_05.yaml thresholds ...
    p.max_range, p.num_rings, p.num_sectors = 80.0, 20, 108  # ... on BASELINE.json config[1]'s 20 x 108 R-POD (80 m, as in seq_00/07/large_scale YAMLs)
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streets", type=int, default=5)
    ap.add_argument("--street-length", type=float, default=1000.0)
    ap.add_argument("--az-steps", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--profile-all", action="store_true", help="second pass with per-kernel HIP events (breakdown on stderr)")
    ap.add_argument("--lookahead", type=int, default=2, choices=[1, 2], help="scans announced ahead (erasor_hip_prefetch_scan)")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="do not announce the next scan (erasor_hip_prefetch_scan): every step runs its own query chain first")
    args = ap.parse_args()

    import torch
    import erasor_amd
    from erasor_amd import synth

    from erasor_amd import dist as ed
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # "nccl" is RCCL on ROCm.  ERASOR_BENCH_BACKEND / ERASOR_BENCH_ONE_DEVICE exist only to smoke-test the multi-rank
    # control flow on a single-GPU box (RCCL refuses two ranks on one device).
    dist, world_size, rank, local_rank = ed.init(os.environ.get("ERASOR_BENCH_BACKEND", "nccl"))
    if os.environ.get("ERASOR_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    erasor_amd.build()
    world, lidar = build_workload(args, rank)
    P = make_params(args)
    K, W = args.steps, args.warmup
    n_frames = K + W + 2  # scans beyond the timed ones: the last timed steps announce them like every other step

    # ---- the global map: rank 0 samples it, RCCL broadcast over xGMI to every replica ----
    t0 = time.time()
    m = world.sample_map(spacing=0.2, frames=range(0, 320, 2), step=1.0) if rank == 0 else None
    if dist is not None:
        dist.barrier()
    tb = time.time()
    d_map = ed.broadcast_map(dist, rank, dev, m)
    torch.cuda.synchronize()
    t_bcast = (time.time() - tb) if dist is not None else None
    N_map = int(d_map.shape[0])
    t_map = time.time() - t0

    # ---- this rank's scans (its shard of the scan stream) and poses; uploaded before the timed region ----
    jr = np.random.default_rng(1234 + rank)
    x0, frame_ids = ed.shard_frames(rank, world_size, n_frames)
    scans, Tb, To = [], [], []
    for k in frame_ids:
        p7 = world.pose(k, 1.0, x0=x0, jitter_rng=jr)
        scans.append(world.cast(p7, lidar, k))
        tb_ = erasor_amd.geopose2eigen(p7)
        Tb.append(tb_)
        To.append(erasor_amd.invert_rigid(tb_))
    Tl = erasor_amd.geopose2eigen([0, 0, synth.LIDAR_HEIGHT, 0, 0, 0, 1])
    d_scans = [torch.from_numpy(s).to(dev) for s in scans]
    n_scan = int(np.mean([len(s) for s in scans]))
    torch.cuda.synchronize()

    g = erasor_amd.Erasor(P, device=local_rank)
    g.set_map_device(d_map.data_ptr(), N_map)

    lookahead = not args.no_lookahead
    LA = args.lookahead  # scans announced ahead of the one being stepped

    def run(k):
        # offline sequence processing: scan k+1 is announced before step k, so that its voxelisation / binning (which do
        # not depend on the map) overlap step k's map-side stages; step k returns with ITS results on the host as before
        if lookahead and k + LA < n_frames:
            g.prefetch_device(d_scans[k + LA].data_ptr(), len(scans[k + LA]), Tl)
        return g.step_device(d_scans[k].data_ptr(), len(scans[k]), Tl, Tb[k], To[k])

    if lookahead:
        for j in range(LA):
            g.prefetch_device(d_scans[j].data_ptr(), len(scans[j]), Tl)
    for k in range(W):
        run(k)
    g.profile_reset()
    g.profiling(2)  # HIP events around voi_split only, on the handle's stream, during the timed region
    split_bytes = []
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t_start = time.perf_counter()
    last = None
    for k in range(W, W + K):
        split_bytes.append(g.voi_split_bytes())
        last = run(k)  # synchronous: returns after the step's results are on the host
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    prof = g.profile_get()
    g.profiling(0)
    elapsed = ed.max_over_ranks(dist, elapsed, dev)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = world_size * K / elapsed
    ms_per_step = elapsed / K * 1e3
    # ---- roofline of the dominant kernel (voi_split): bytes its layout must stream per launch / measured launch time ----
    # k_voi_split is launched with its own start / stop HIP events (hipExtLaunchKernelGGL) on the handle's stream during the
    # timed region: they stamp the kernel's execution window, which is also what rocprofv3 reports (profiles/).
    vs_ms, vs_n = prof.get("voi_split", (0.0, 0))
    avg_ms = vs_ms / max(vs_n, 1)
    alg_bytes = float(np.mean([b for b, _ in split_bytes]))
    entries = float(np.mean([e for _, e in split_bytes]))
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None  # HBM bytes per launch from the PMC counters: measured in a separate rocprofv3 --pmc pass (profiles/)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            traffic = int(json.load(f)["traffic_bytes_per_launch"])
    except Exception:
        pass
    rocprof_avg = None  # kernel-only average of the committed rocprofv3 --kernel-trace --stats run of this command
    try:
        import csv
        with open(os.path.join(ROOT, "profiles", "kernel_stats_latest.csv")) as f:
            for row in csv.DictReader(f):
                if "k_voi_split" in row["Name"]:
                    rocprof_avg = round(float(row["AverageNs"]) / 1e3, 2)
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": "k_voi_split", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic, "bytes_per_launch": int(alg_bytes),
                "entries_per_launch": int(entries), "avg_launch_us": round(avg_ms * 1e3, 2),
                "rocprofv3_kernel_avg_us": rocprof_avg,
                "launches": int(vs_n),
                "aos16_equiv_GBps": round(16.0 * entries / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else 0.0}

    # ---- optional per-kernel breakdown (extra K steps, all kernels bracketed) ----
    if args.profile_all:
        g.profile_reset()
        g.profiling(1)
        for k in range(W, W + K):
            # re-running the same scans against the already-updated map is fine for a time breakdown
            run(k)
        pa = g.profile_get()
        g.profiling(0)
        tot = sum(v[0] for v in pa.values())
        for name, (ms, cnt) in sorted(pa.items(), key=lambda kv: -kv[1][0]):
            print("  %-14s %8.3f ms/step  (%5.1f%%, %d launches)" % (name, ms / K, 100 * ms / max(tot, 1e-9), cnt // K), file=sys.stderr)
        print("  sum of kernels %.3f ms/step vs wall %.3f ms/step" % (tot / K, ms_per_step), file=sys.stderr)

    # ---- the same K scans again without look-ahead (every step runs its own query chain first): latency-style figure ----
    sync_ms = None
    if lookahead and world_size == 1:
        lookahead = False
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for k in range(W, W + K):
            run(k)  # against the already-updated map: same work per step, results not used
        torch.cuda.synchronize()
        sync_ms = (time.perf_counter() - ts) * 1e3 / K
        lookahead = True

    # ---- CPU baseline: the oracle (single-threaded port, like the single-threaded reference) on a bounded sample ----
    cpu = None
    if world_size == 1 and not args.no_cpu_baseline:
        import ctypes as C
        from oracle import orc  # the CPU oracle: cpu_baseline leg only
        po = orc.Params()
        C.memmove(C.byref(po), C.byref(P), C.sizeof(po))
        o = orc.Oracle(po)
        o.set_map(m)
        ns = min(args.cpu_steps, n_frames)
        tc = time.perf_counter()
        for k in range(ns):
            o.step(scans[k], Tl, Tb[k], To[k])
        tcpu = time.perf_counter() - tc
        cpu = {"value": round(ns / tcpu, 3), "unit": "scans/s", "cores": 1, "kind": "port",
               "sample": "%d steps of the same workload (same %d-pt map, same scans), oracle/erasor_oracle.cpp -O2, 1 thread" % (ns, N_map)}
        o.close()

    out = {
        "metric": "scans_per_sec", "value": round(value, 2), "unit": "scans/s", "n_gpus": world_size, "steps": K, "warmup": W,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (transforms, R-GPF) + f64 (VoI test, polar binning, scan ratio)", "data": "synthetic",
        "config": {"workload": "KITTI-05-shaped synthetic street, %d-pt map resident in HBM, ~%d-pt HDL-64-like scans, R-POD 20 rings x 108 sectors @ 80 m, "
                               "seq_05.yaml thresholds, ERASOR v3; one scan per step, 1 m/frame" % (N_map, n_scan),
                   "map_points": N_map, "scan_points": n_scan, "rings": 20, "sectors": 108, "sharding": "scan-parallel replicas, RCCL broadcast of the map",
                   "lookahead_scans": LA if lookahead else 0},
        "map_points_x_scans_per_sec": round(value * N_map, 1),
        "ms_per_step_without_lookahead": None if sync_ms is None else round(sync_ms, 4),
        "roofline": roofline, "cpu_baseline": cpu,
        "last_step": last.as_dict() if last is not None else None,
        "setup_s": {"map_build_and_upload": round(t_map, 2), "rccl_broadcast": None if t_bcast is None else round(t_bcast, 4)},
    }
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
