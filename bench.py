#!/usr/bin/env python3
"""bench.py — ERASOR per-scan hot path on MI355X.

A "step" is one pass of the hot path (query voxelise -> fetch_VoI -> R-POD -> Scan Ratio Test -> R-GPF ->
map write-back; reference OfflineMapUpdater.cpp:237-294) over one synthetic scan against a map that is
resident in HBM.  Inputs (map and scans) are in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W [--workload W] [--mode M]

--gpus N > 1 without a torchrun environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU
(backend "nccl" = RCCL).  It refuses to run with fewer than N devices.

workloads (BASELINE.json configs):
  seq05           config[1]: ~10 M-pt map, ~120 k-pt HDL-64 scans, R-POD 20 x 108 @ 80 m, seq_05.yaml thresholds  (default)
  seq05_yaml      config[1] with config/seq_05.yaml's own geometry: 15 x 60 @ 60 m
  large_scale_05  config[3]: config/large_scale_05.yaml, ~40 M-pt un-voxelised map (> the 256 MiB L3: fetch_VoI is truly HBM-bound)
  ouster128       config[4]: config/your_own_env_ouster.yaml, 256 k-pt Ouster-128 scans, 20 x 60 @ 20 m
modes:
  replicas        rank 0 builds the map, ONE RCCL broadcast over xGMI, every rank processes its own shard of the scan
                  stream against its replica, no data-path collective (weak scaling)                                 (default)
  seq-per-gpu     config[2]: the five KITTI-shaped sequences 00/01/02/05/07 (config/seq_0X.yaml parameters), dealt
                  round-robin over the ranks; every rank builds its own worlds, no map exchange
Both modes end with one RCCL all_gather of per-rank result counters.

Prints ONE JSON line on rank 0 (see DESIGN.md §6 for the roofline and cpu_baseline definitions).
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

# (several handles in one process -- seq-per-gpu mode -- need more hardware queues than HIP's default of 4, or a handle's two query streams can
# share one and its chains serialise: gpurun_out/r03ap.  Read when the HIP runtime starts, i.e. before torch or the library touch the device;
# liberasor_hip.so sets the same default when it is loaded)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (params preset, R-POD override, lidar, map spacing, streets, description)
    "seq05": dict(seq="05", rpod=(80.0, 20, 108), lidar="hdl64", spacing=0.2, l2b_z=True,
                  desc="KITTI-05-shaped synthetic street, R-POD 20 rings x 108 sectors @ 80 m, seq_05.yaml thresholds"),
    "seq05_yaml": dict(seq="05", rpod=None, lidar="hdl64", spacing=0.2, l2b_z=True,
                       desc="KITTI-05-shaped synthetic street, config/seq_05.yaml verbatim (15 x 60 @ 60 m)"),
    "large_scale_05": dict(seq="large_scale_05", rpod=None, lidar="hdl64", spacing=0.1, l2b_z=True,
                           desc="config/large_scale_05.yaml (20 x 108 @ 80 m), dense un-voxelised map"),
    "ouster128": dict(seq="ouster", rpod=None, lidar="ouster128", spacing=0.2, l2b_z=False,
                      desc="config/your_own_env_ouster.yaml (20 x 60 @ 20 m, lidar2body = identity), Ouster-128 256 k-pt scans"),
}
SEQS = ["00", "01", "02", "05", "07"]  # BASELINE config[2]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=7,
                    help="the K-step timed pass is repeated this many times inside ONE invocation, the sequence simply continuing (each pass "
                         "bracketed by barrier + device synchronisation like the contract's one pass); ms_per_step / value are the MEDIAN pass, "
                         "ms_per_step_min / _max the spread.  One 20-step pass is a 4 ms sample on boxes that differ by +-8 %% (VERDICT r04).  "
                         "Applies where one sequence per rank is stepped by the native node loop (the default mode); otherwise 1")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="seq05")
    ap.add_argument("--mode", choices=["replicas", "seq-per-gpu"], default="replicas")
    ap.add_argument("--large-scale-mode", choices=["off", "on"], default="off",
                    help="large_scale_05 only: is_large_scale with submap_size 160 (launch/run_erasor_in_large_scale.launch)")
    ap.add_argument("--streets", type=int, default=5)
    ap.add_argument("--street-length", type=float, default=1000.0)
    ap.add_argument("--az-steps", type=int, default=0, help="0 = the sensor's own (2000 HDL-64, 2048 Ouster-128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="stop the CPU leg after this much CPU work")
    ap.add_argument("--verify-prefix-seconds", type=float, default=0.0,
                    help="with --no-cpu-baseline: replay the FIRST steps the GPU ran (warm-up first) on the oracle for this much CPU time and compare every "
                         "erasor_step_result -- the bounded check of the extra passes of the default run (the headline and one config-4 pass replay every step)")
    ap.add_argument("--no-verify-all", dest="verify_all", action="store_false",
                    help="let --cpu-seconds also bound the oracle's replay of the GPU's steps (default: every step the GPU ran -- warm-up and all "
                         "timed passes -- is replayed and compared, then the final map: ~0.08 s per step on the 10 M-point map)")
    ap.add_argument("--profile-all", action="store_true", help="second pass with per-kernel HIP events (breakdown on stderr)")
    ap.add_argument("--lookahead", type=int, default=7, choices=[1, 2, 3, 4, 5, 6, 7],
                    help="nodes announced ahead (erasor_hip_run_nodes / erasor_hip_prefetch_node).  Round 6: six -- the query chains of announced nodes\n"
                         "beyond the --chain-lead-th in line share their launches --chain-batch at a time (erasor_hip_chain_batch), and a set of two\n"
                         "of three takes about three steps to get through its queue")
    ap.add_argument("--chain-batch", type=int, default=2, choices=[1, 2, 3, 4],
                    help="query chains of announced nodes that share one set of launches (1: every chain on its own, round 5's behaviour)")
    ap.add_argument("--chain-lead", type=int, default=3, choices=[1, 2, 3, 4, 5, 6],
                    help="a chain is held back for a shared set only while this many chains are in their queues in front of it")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="do not announce the next scan (erasor_hip_prefetch_scan): every step runs its own query chain first")
    ap.add_argument("--seqs", type=int, default=len(SEQS), help="seq-per-gpu: use only the first N of the five sequences (e.g. 2: what one GPU of config 3's four gets)")
    ap.add_argument("--placement", choices=["queue", "static"], default="queue",
                    help="seq-per-gpu: which rank runs which sequence -- 'queue': the next sequence goes to whichever rank is idle (BASELINE config 3: "
                         "the fifth sequence onto the first free GPU; every rank prepares all sequences, a shared counter hands them out); "
                         "'static': dealt round-robin in advance")
    ap.add_argument("--interleave", choices=["async", "threads", "pairs", "off"], default="off",
                    help="seq-per-gpu with several sequences on one rank: off = one sequence after the other (default: measured fastest on one "
                         "MI355X -- the chains of two sequences slow each other down more than the overlap gains: 2 sequences 3864 scans/s one "
                         "after the other vs 3145 interleaved, 5 sequences 3766 vs 2877, gpurun_out/r03n); async = one host thread keeps every "
                         "sequence's step in flight (erasor_hip_step_async / _wait); threads = one host thread per sequence, blocking steps; pairs (round 6) = "
                         "threads, two sequences at a time, every handle with ONE query stream (ERASOR_HIP_QSTREAMS=1): four busy queues on the "
                         "process's four compute pipes -- the layout in which sharing a GPU pays (1.5 x for two sequences)")
    ap.add_argument("--interleave-group", type=int, default=0,
                    help="--interleave threads: how many sequences run side by side (0: all of the rank's).  Round 6: a process has FOUR compute pipes;\n"
                         "with ERASOR_HIP_QSTREAMS=1 a handle keeps two queues busy (main + one query stream), so TWO sequences side by side have a\n"
                         "pipe per queue: 2 is the group that pays (7950-8010 scans/s against 5310 one after the other)")
    ap.add_argument("--python-loop", action="store_true",
                    help="drive the timed steps from a Python loop (prefetch + step per node) instead of ONE erasor_hip_run_nodes call "
                         "(the offline driver's node loop in native code: the same calls, without ~20 us of interpreter time between two steps)")
    ap.add_argument("--no-pr-rr", action="store_true", help="skip PR / RR of the final map (erasor_amd.evalmap; ~10-20 s of host time for a 10 M-point map)")
    ap.add_argument("--no-callback-bench", action="store_true", help="skip the C++ drop-in path benchmark (erasor_offline_demo --bench)")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="default run (1 GPU, seq05, replicas) only: do not append the short passes over BASELINE configs 4 / 2-yaml / 5")
    ap.add_argument("--union-eval", type=int, default=0, metavar="N",
                    help="replicas mode: after the timed pass, SURVEY 8(e)(ii)'s result exchange -- every rank applies N scans of its shard to the "
                         "INITIAL map each (Jacobi-style), ONE all_gather of the removed initial-map indices, PR / RR of the united map next to the "
                         "sequential fold's (erasor_amd/dist.py)")
    ap.add_argument("--eval", action="store_true", help="seq-per-gpu: PR/RR of every sequence's final map (erasor_amd.evalmap)")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` must be an N-rank RCCL run by itself"""
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not os.environ.get("ERASOR_BENCH_ONE_DEVICE"):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible — refusing to measure fewer ranks than asked" % (args.gpus, n_dev))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def device_source_hash():
    """sha256 over the device sources: profiles/ records it when the rocprofv3 passes are collected (tools/collect_profiles.sh);
    a different hash here means a kernel has changed since, and the committed counters no longer describe this build"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "erasor_amd", "csrc")
    for f in ("erasor_hip.hip", "kernels.hip.h", "revert_bins.hip.h", "exact_sort.hip.h", "exact_sort_core.h"):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_profile(prof_tag):
    """what the separate rocprofv3 passes of THIS build measured (profiles/*_latest*): k_voi_split's PMC traffic and average
    duration, and the kernel that tops the GPU-time table.  None for everything when the device sources have changed since."""
    import csv
    out = {"traffic": None, "voi_split_avg_us": None, "dominant": None, "critical": None, "stale": None, "step_traffic": None, "step_traffic_steps_only": None}
    try:
        with open(os.path.join(ROOT, "profiles", "latest_meta%s.json" % prof_tag)) as f:
            meta = json.load(f)
        out["stale"] = meta.get("device_source_sha16") != device_source_hash()
    except Exception:
        out["stale"] = True
    if out["stale"]:
        return out
    per_kernel = {}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest%s.json" % prof_tag)) as f:
            j = json.load(f)
        out["traffic"] = int(j["traffic_bytes_per_launch"])
        out["step_traffic"] = j.get("step_traffic_bytes")
        out["step_traffic_steps_only"] = j.get("step_traffic_bytes_without_setup_kernels")
        per_kernel = j.get("per_kernel_traffic_bytes", {})
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "profiles", "kernel_stats_latest%s.csv" % prof_tag)) as f:
            rows = [r for r in csv.DictReader(f) if r["Name"].startswith("ek::") or "ek::" in r["Name"]]
        for r in rows:
            if "k_voi_split" in r["Name"]:
                out["voi_split_avg_us"] = round(float(r["AverageNs"]) / 1e3, 2)
        steps = max([int(r["Calls"]) for r in rows if "k_voi_gather" in r["Name"]] + [1])  # (one gather per step)
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        name = top["Name"].split("(")[0].replace("void ", "").replace("ek::", "")
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        dom = {"name": name, "avg_us": round(float(top["AverageNs"]) / 1e3, 2), "launches_per_scan": round(int(top["Calls"]) / steps, 1),
               "share_of_gpu_time": round(float(top["TotalDurationNs"]) / tot, 4), "source": "from_profiles (rocprofv3 --kernel-trace --stats)"}
        tb = per_kernel.get(name)
        if tb:
            dom["traffic_bytes_per_launch"] = int(tb)
            dom["achieved_GBps"] = round(tb / (float(top["AverageNs"]) * 1e-9) / 1e9, 1)
            dom["frac_of_hbm_peak"] = round(dom["achieved_GBps"] / PEAK_HBM_GBPS, 4)
        out["dominant"] = dom
        # the kernel that owns the step's dependency chain: the per-bin launch (R-GPF + per-bin voxelisation, one workgroup per reverted bin)
        for r in rows:
            if "k_revert_bins_srt" in r["Name"]:
                cname = "k_revert_bins_srt"
                crit = {"name": cname, "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2),
                        "launches_per_scan": round(int(r["Calls"]) / steps, 2), "share_of_gpu_time": round(float(r["TotalDurationNs"]) / tot, 4),
                        "bound": "latency (one 1024-thread workgroup per reverted bin: exact std::sort emulation, sequential float32 sums, one-lane SVD)",
                        "source": "from_profiles (rocprofv3 --kernel-trace --stats)"}
                tb = per_kernel.get(cname)
                if tb:
                    crit["traffic_bytes_per_launch"] = int(tb)
                    crit["achieved_GBps"] = round(tb / (float(r["AverageNs"]) * 1e-9) / 1e9, 1)
                    crit["frac_of_hbm_peak"] = round(crit["achieved_GBps"] / PEAK_HBM_GBPS, 4)
                out["critical"] = crit
    except Exception:
        pass
    return out


def host_identity():
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count()}


def make_params(wl, args, seq=None):
    import erasor_amd
    from erasor_amd import synth
    p = erasor_amd.params_default()
    synth.apply_params(p, seq or wl["seq"])
    if seq is None and wl["rpod"]:
        p.max_range, p.num_rings, p.num_sectors = wl["rpod"]
    if seq is None and wl["seq"] == "large_scale_05" and args.large_scale_mode == "on":
        p.is_large_scale, p.submap_size = 1, 160.0
    return p


def make_lidar(wl, args):
    from erasor_amd import synth
    if wl["lidar"] == "ouster128":
        return synth.Lidar.ouster128(args.az_steps or 2048)
    return synth.Lidar.hdl64(args.az_steps or 2000)


_CAST = None


def _cast_one(k):
    world, lidar, poses = _CAST
    return world.cast(poses[k], lidar, k)


def cast_scans(world, lidar, poses):
    """world.cast for every pose, on up to 16 host cores (fork: the children inherit the world, run numpy only and return the scan)"""
    global _CAST
    n = len(poses)
    workers = min(16, os.cpu_count() or 1, n)
    if workers <= 1 or n < 8 or os.environ.get("ERASOR_BENCH_SERIAL_CAST"):
        return [world.cast(poses[k], lidar, k) for k in range(n)]
    import multiprocessing as mp
    _CAST = (world, lidar, poses)
    try:
        with mp.get_context("fork").Pool(workers) as pool:
            return pool.map(_cast_one, range(n), chunksize=max(1, n // (4 * workers)))
    finally:
        _CAST = None


class Sequence:
    """one map + its scan stream, resident on the device, with the look-ahead driver loop"""

    def __init__(self, args, P, world, lidar, d_map, n_map, x0, n_frames, dev, device_index, l2b_z, seed):
        import torch
        import erasor_amd
        from erasor_amd import synth
        self.args, self.P, self.n_frames = args, P, n_frames
        jr = np.random.default_rng(seed)
        self.scans, self.Tb, self.To, self.poses = [], [], [], []
        for k in range(n_frames):
            self.poses.append(world.pose(k, 1.0, x0=x0, jitter_rng=jr))
        # the synthetic scans are ray-cast on the host (numpy, ~0.25 s each): with the repeated timed pass a run needs ~150 of them,
        # cast side by side on the host cores (set-up time only; the children never touch the device)
        self.scans = cast_scans(world, lidar, self.poses)
        for k in range(n_frames):
            p7 = self.poses[k]
            if not l2b_z:  # sensors whose lidar2body is the identity: express the scan in the body frame (z up from the ground)
                s = self.scans[k].copy()
                s[:, 2] += synth.LIDAR_HEIGHT
                self.scans[k] = s
            tb = erasor_amd.geopose2eigen(p7)
            self.Tb.append(tb)
            self.To.append(erasor_amd.invert_rigid(tb))
        self.Tl = erasor_amd.geopose2eigen([0, 0, synth.LIDAR_HEIGHT if l2b_z else 0.0, 0, 0, 0, 1])
        self.d_scans = [torch.from_numpy(s).to(dev) for s in self.scans]
        # what a step call needs, converted once (a C++ caller has no such cost; in Python the conversions of the three
        # matrices and the tensor look-ups were ~15 us of main-stream idle time between two steps)
        cm = (lambda t: t) if os.environ.get("ERASOR_BENCH_NUMPY_ARGS") else erasor_amd.c_mat  # (A/B switch)
        self.c_Tl = cm(self.Tl)
        self.c_Tb = [cm(t) for t in self.Tb]
        self.c_To = [cm(t) for t in self.To]
        self.d_ptr = [t.data_ptr() for t in self.d_scans]
        self.n_pts = [len(s) for s in self.scans]
        self.n_scan = int(np.mean([len(s) for s in self.scans]))
        self.g = erasor_amd.Erasor(P, device=device_index)
        self.g.set_map_device(d_map.data_ptr(), n_map)
        self.g.chain_batch(args.chain_batch, args.chain_lead)
        self.lookahead = not args.no_lookahead
        self.LA = args.lookahead
        self.primed = False
        # nodes are announced with their pose (scan + odometry, like erasor::node): the next step's VoI split is launched ahead
        self.with_pose = not os.environ.get("ERASOR_BENCH_NO_POSE_AHEAD")
        # the whole sequence as arrays for erasor_hip_run_nodes (the node loop in native code)
        import ctypes as C
        self.na = erasor_amd.Erasor.node_arrays(self.d_ptr, self.n_pts, self.Tb, self.To)
        self.announced = C.c_size_t(0)

    def run_block(self, first, count):
        """nodes [first, first + count) in ONE native call: each announced LA ahead with its pose, stepped one after the other"""
        P, N, Tb, To = self.na
        self.primed = True
        return self.g.run_nodes(P, N, self.c_Tl, Tb, To, first, count, self.LA if self.lookahead else 0, self.announced)

    def prime(self):
        if self.lookahead and not self.primed:
            for j in range(min(self.LA, self.n_frames)):
                self.g.prefetch_device(self.d_ptr[j], self.n_pts[j], self.c_Tl, self.c_Tb[j] if self.with_pose else None, self.c_To[j] if self.with_pose else None)
        self.primed = True

    def run(self, k):
        # offline sequence processing: scan k+LA is announced before step k, so that its voxelisation / binning (which do
        # not depend on the map) overlap step k's map-side stages; step k returns with ITS results on the host as before
        g = self.g
        if self.lookahead and k + self.LA < self.n_frames:
            g.prefetch_device(self.d_ptr[k + self.LA], self.n_pts[k + self.LA], self.c_Tl, self.c_Tb[k + self.LA] if self.with_pose else None,
                              self.c_To[k + self.LA] if self.with_pose else None)
        return g.step_device(self.d_ptr[k], self.n_pts[k], self.c_Tl, self.c_Tb[k], self.c_To[k])

    def run_async(self, k):
        """the same step in two halves (erasor_hip_step_async / erasor_hip_step_wait): announce, enqueue, return"""
        g = self.g
        if self.lookahead and k + self.LA < self.n_frames:
            g.prefetch_device(self.d_ptr[k + self.LA], self.n_pts[k + self.LA], self.c_Tl, self.c_Tb[k + self.LA] if self.with_pose else None,
                              self.c_To[k + self.LA] if self.with_pose else None)
        g.step_async(self.d_ptr[k], self.n_pts[k], self.c_Tl, self.c_Tb[k], self.c_To[k], device=True)

    def wait(self):
        return self.g.step_wait()


def cpu_baseline(args, P, m, seq, l2b7, gpu_results=None, gpu_final=None):
    """the CPU path on this box's host cores, same workload, bounded sample.  Preferred: oracle/_ref = the reference's own
    sources (kind "reference", with its own two wall-clock spans); the oracle port is timed beside it -- over the very steps
    the GPU has just run (warm-up + timed), and every step's erasor_step_result, the last step's dynamic-point mask and the
    final map are compared with what the GPU produced: a mismatch fails the run."""
    import ctypes as C
    from oracle import orc, ref  # checker / baseline only — never on the product path
    po = orc.Params()
    C.memmove(C.byref(po), C.byref(P), C.sizeof(po))
    out = {}
    o = orc.Oracle(po)
    o.set_map(m)
    ns, tcpu = 0, 0.0
    n_want = len(gpu_results) if gpu_results else min(args.cpu_steps, seq.n_frames)
    parity = {"parity_checked_steps": 0}
    for k in range(min(n_want, seq.n_frames)):
        tc = time.perf_counter()
        ro = o.step(seq.scans[k], seq.Tl, seq.Tb[k], seq.To[k])
        tcpu += time.perf_counter() - tc
        ns += 1
        if gpu_results:
            do, dg = ro.as_dict(), gpu_results[k].as_dict()
            bad = [f for f in do if f not in ("n_ambiguous", "n_sort_fallback") and do[f] != dg[f]]
            if bad or dg["n_ambiguous"]:
                raise SystemExit("bench.py: PARITY FAILURE at step %d (GPU vs oracle): %s" % (k, {f: (dg[f], do[f]) for f in bad} or "n_ambiguous != 0"))
            parity["parity_checked_steps"] = ns
        if tcpu > args.cpu_seconds / 2 and not (gpu_results and args.verify_all):
            break
    if gpu_results:
        parity["parity"] = "erasor_step_result of %d GPU steps (warm-up + timed pass, bench call pattern) == the oracle's" % ns
        if gpu_final is not None and ns == len(gpu_results):
            gm, grej = gpu_final
            om, orej = o.get_map(), o.get_rejected_indices()
            same_map = gm.shape == om.shape and bool((gm.view(np.uint32) == om.view(np.uint32)).all())
            same_rej = grej.shape == orej.shape and bool((grej == orej).all())
            if not (same_map and same_rej):
                raise SystemExit("bench.py: PARITY FAILURE after the timed pass: final map identical %s, last dynamic-point mask identical %s" % (same_map, same_rej))
            parity["parity"] += "; the %d-point map the timed pass leaves and its last dynamic-point mask (%d indices) are bit-identical" % (len(gm), len(grej))
            parity["final_map_checked"] = True
        else:
            parity["final_map_checked"] = False
    o.close()
    port = {"value": round(ns / tcpu, 3), "unit": "scans/s", "cores": 1, "kind": "port",
            "sample": "%d steps of the same workload (same %d-pt map, same scans), oracle/erasor_oracle.cpp -O2 (copy-free restatement), 1 thread" % (ns, len(m))}
    if not ref.available():
        return port, None, parity
    r = ref.RefUpdater(po, m, l2b7)
    ns, tcpu, sv, se = 0, 0.0, 0.0, 0.0
    for k in range(min(args.cpu_steps, seq.n_frames)):
        tc = time.perf_counter()
        r.step(seq.scans[k], seq.poses[k])
        tcpu += time.perf_counter() - tc
        a, b = r.spans()
        sv, se, ns = sv + a, se + b, ns + 1
        if tcpu > args.cpu_seconds / 2:
            break
    r.close()
    refd = {"value": round(ns / tcpu, 3), "unit": "scans/s", "cores": 1, "kind": "reference sources over stand-in headers (NOT a reference build)",
            "kind_note": "the reference's build needs ROS, PCL, Eigen and tf, none of which this image has: by the task's rules it is unbuildable here and "
                         "this figure does not count as the reference's -- it times the reference's own source text (its full-map copies included) "
                         "compiled against the oracle's restated third-party arithmetic; cpu_baseline is the oracle port",
            "sample": "%d callback_node steps of the same workload (same %d-pt map, same scans) through oracle/_ref = the reference's "
                      "unmodified erasor.cpp / erasor_utils.cpp / OfflineMapUpdater.cpp (-O2, 1 thread, like the single-threaded node); "
                      "PCL/Eigen/ROS underneath are the stand-ins of oracle/stubs (publishing is a no-op)" % (ns, len(m)),
            "ms_per_scan": round(tcpu / ns * 1e3, 1),
            "reference_spans_ms": {"Extracting VoI": round(sv / ns * 1e3, 1), "ERASOR": round(se / ns * 1e3, 1)}}
    return port, refd, parity


def verify_prefix(args, P, m, seq, gpu_results):
    """the oracle over the first steps the GPU ran, bounded by --verify-prefix-seconds of CPU work: every erasor_step_result compared"""
    import ctypes as C
    from oracle import orc  # checker only -- never on the product path
    po = orc.Params()
    C.memmove(C.byref(po), C.byref(P), C.sizeof(po))
    o = orc.Oracle(po)
    o.set_map(m)
    ns, tcpu = 0, 0.0
    for k in range(min(len(gpu_results), seq.n_frames)):
        tc = time.perf_counter()
        ro = o.step(seq.scans[k], seq.Tl, seq.Tb[k], seq.To[k])
        tcpu += time.perf_counter() - tc
        do, dg = ro.as_dict(), gpu_results[k].as_dict()
        bad = [f for f in do if f not in ("n_ambiguous", "n_sort_fallback") and do[f] != dg[f]]
        if bad or dg["n_ambiguous"]:
            raise SystemExit("bench.py: PARITY FAILURE at step %d (GPU vs oracle): %s" % (k, {f: (dg[f], do[f]) for f in bad} or "n_ambiguous != 0"))
        ns += 1
        if tcpu > args.verify_prefix_seconds:
            break
    o.close()
    return {"parity_checked_steps": ns, "final_map_checked": False,
            "parity": "erasor_step_result of the first %d of %d GPU steps (warm-up first, bench call pattern) == the oracle's (%.1f s of CPU: a bounded check)"
                      % (ns, len(gpu_results), tcpu)}


def _ref_sequence_worker(job):
    """one reference updater (oracle/_ref) on one sequence, in its own process: (steps, seconds)"""
    import ctypes as C
    from oracle import orc, ref  # baseline only
    pbytes, m, scans, poses, l2b7, n_steps = job
    po = orc.Params()
    C.memmove(C.byref(po), pbytes, C.sizeof(po))
    r = ref.RefUpdater(po, m, l2b7)
    t0 = time.perf_counter()
    for k in range(n_steps):
        r.step(scans[k], poses[k])
    dt = time.perf_counter() - t0
    r.close()
    return n_steps, dt


def cpu_sequence_parallel(args, seqs, maps, l2b7):
    """config 3's fair multi-core comparison: one single-threaded reference updater per sequence, min(5, nproc)
    of them side by side on the host cores; aggregate scans/s over the slowest worker's wall time"""
    import ctypes as C
    import multiprocessing as mp
    from oracle import ref
    if not ref.available():
        return None
    jobs = []
    n_steps = max(1, min(args.cpu_steps, 3))
    for sid, s in seqs:
        pb = bytes(C.string_at(C.addressof(s.P), C.sizeof(s.P)))
        jobs.append((pb, maps[sid], s.scans[:n_steps], s.poses[:n_steps], l2b7, n_steps))
    cores = min(len(jobs), os.cpu_count() or 1)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_ref_sequence_worker, jobs)
    wall = time.perf_counter() - t0
    steps = sum(r[0] for r in res)
    slowest = max(r[1] for r in res)
    return {"value": round(steps / slowest, 3), "unit": "scans/s", "cores": cores, "kind": "reference sources over stand-in headers (NOT a reference build)",
            "sample": "%d sequences x %d callback_node steps, one single-threaded updater of the reference's sources (oracle/_ref) per sequence on %d host "
                      "cores side by side; aggregate steps / slowest worker's stepping time (wall incl. start-up %.1f s)"
                      % (len(jobs), n_steps, cores, wall)}


def main():
    args = parse_args()
    if args.interleave == "pairs":
        # round 6: two sequences side by side, each handle with ONE query stream (main + query = two busy compute queues; the process has
        # four compute pipes): the layout in which independent sequences DO share a GPU -- 7900-8060 scans/s for two against 5400 one
        # after the other.  (Read by the library when a handle is created: set before the first one.)
        os.environ.setdefault("ERASOR_HIP_QSTREAMS", "1")
        args.interleave, args.interleave_group = "threads", 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)  # does not return
    import torch
    import erasor_amd
    from erasor_amd import synth
    from erasor_amd import dist as ed

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # "nccl" is RCCL on ROCm.  ERASOR_BENCH_BACKEND / ERASOR_BENCH_ONE_DEVICE exist only to smoke-test the multi-rank
    # control flow on a single-GPU box (RCCL refuses two ranks on one device).
    backend = os.environ.get("ERASOR_BENCH_BACKEND", "nccl")
    dist, world_size, rank, local_rank = ed.init(backend)
    if world_size != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world_size))
    if os.environ.get("ERASOR_BENCH_ONE_DEVICE"):
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU %d (%d visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # (a stale library is rebuilt by ONE rank; the others wait, then find it up to date)
    if rank == 0:
        erasor_amd.build()
    if dist is not None:
        dist.barrier()
    erasor_amd.build()
    wl = WORKLOADS[args.workload]
    K, W = args.steps, args.warmup
    # the timed pass is repeated (the sequence continues) where ONE sequence per rank is stepped by the native node loop
    R = max(1, args.repeats) if (args.mode == "replicas" and not args.python_loop and not os.environ.get("ERASOR_BENCH_NO_POSE_AHEAD")) else 1
    n_frames = K * R + W + args.lookahead  # scans beyond the timed ones: the last timed steps announce them like every other step
    lidar = make_lidar(wl, args)
    l2b7 = [0, 0, synth.LIDAR_HEIGHT if wl["l2b_z"] else 0.0, 0, 0, 0, 1]

    t0 = time.time()
    t_bcast = None
    m = None
    seqs = []  # the Sequence objects this rank owns (replicas: one; seq-per-gpu: its share of the five)
    if args.mode == "replicas":
        world = synth.World(seed=20210305 + 5, length=args.street_length, n_streets=args.streets, street_gap=50.0, n_moving=10, n_peds=6)
        P = make_params(wl, args)
        # ---- the global map: rank 0 samples it, ONE RCCL broadcast over xGMI to every replica ----
        m = world.sample_map(spacing=wl["spacing"], frames=range(0, 320, 2), step=1.0) if rank == 0 else None
        if dist is not None:
            dist.barrier()
        tb = time.time()
        d_map = ed.broadcast_map(dist, rank, dev, m)
        torch.cuda.synchronize()
        t_bcast = (time.time() - tb) if dist is not None else None
        x0, _ = ed.shard_frames(rank, world_size, n_frames)
        seqs.append(("replica%d" % rank, Sequence(args, P, world, lidar, d_map, int(d_map.shape[0]), x0, n_frames, dev, local_rank,
                                                  wl["l2b_z"], 1234 + rank)))
        maps = {"replica%d" % rank: m}
    else:
        maps = {}
        n_seqs = min(max(args.seqs, 1), len(SEQS))
        queued = args.placement == "queue" and world_size > 1 and args.interleave == "off"
        for i in (range(n_seqs) if queued else ed.deal_round_robin(n_seqs, rank, world_size)):
            sid = SEQS[i]
            world = synth.World(seed=20210305 + int(sid), length=args.street_length, n_streets=args.streets, street_gap=50.0,
                                n_moving=10, n_peds=6)
            P = make_params(wl, args, seq=sid)
            mm = world.sample_map(spacing=wl["spacing"], frames=range(0, 320, 2), step=1.0)
            d_map = torch.from_numpy(mm).to(dev)
            seqs.append((sid, Sequence(args, P, world, lidar, d_map, len(mm), 300.0, n_frames, dev, local_rank, True, 4321 + int(sid))))
            maps[sid] = mm
    torch.cuda.synchronize()
    t_map = time.time() - t0

    step_results = []  # erasor_step_result of every step of the first sequence, warm-up included (compared with the oracle's below)
    native_loop = not args.python_loop and len(seqs) == 1 and not os.environ.get("ERASOR_BENCH_NO_POSE_AHEAD")
    for si, (_, s) in enumerate(seqs):
        if native_loop:
            step_results.extend(s.run_block(0, W))
            continue
        s.prime()
        for k in range(W):
            r_ = s.run(k)
            if si == 0:
                step_results.append(r_)
    first = seqs[0][1] if seqs else None
    if first is not None:
        first.g.profile_reset()
        first.g.chain_timing(reset=True)
        if not os.environ.get("ERASOR_BENCH_NO_SPLIT_EVENTS"):  # (A/B switch: what the roofline's own measurement costs the step)
            # start / stop HIP events attached to every FOURTH k_voi_split launch of the timed region, on the handle's stream (the
            # bracket costs the step it observes ~9 us: measured 0.276 vs 0.267 ms per scan with every launch bracketed / none)
            first.g.profiling(3)
    split_bytes = []
    if not native_loop:
        R = 1
    pass_elapsed = []  # this rank's wall time of every K-step pass
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t_start = time.perf_counter()
    last = None
    totals = np.zeros(6, np.int64)  # steps, map_rejected, reverted_bins, final map size, static, dynamic
    interleave = args.interleave if len(seqs) > 1 else "off"
    if native_loop:
        # ONE call for the K timed nodes (erasor_hip_run_nodes = the offline driver's node loop, main_in_your_env.cpp:92-123) -- R times,
        # each pass between its own barrier + device synchronisation; the sequence continues from pass to pass
        s = seqs[0][1]
        for rep in range(R):
            if rep:
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                t_start = time.perf_counter()
            split_bytes.append(s.g.voi_split_bytes())
            rs = s.run_block(W + rep * K, K)
            if rep + 1 < R:
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                pass_elapsed.append(time.perf_counter() - t_start)
            split_bytes.append(s.g.voi_split_bytes())
            step_results.extend(rs)
            last = rs[-1]
            totals[0] += K
            totals[1] += sum(r.n_map_rejected for r in rs)
            totals[2] += sum(r.n_reverted_bins for r in rs)
        totals[3] += last.n_map_out
        totals[4] += last.n_static
        totals[5] += last.n_dynamic
    elif interleave == "async":
        # independent sequences are independent updaters: ONE host thread keeps a step of every sequence in flight
        for k in range(W, W + K):
            for si, (_, s) in enumerate(seqs):
                if si == 0:
                    split_bytes.append(s.g.voi_split_bytes())
                s.run_async(k)
            for si, (_, s) in enumerate(seqs):
                last = s.wait()
                if si == 0:
                    step_results.append(last)
                totals[0] += 1
                totals[1] += last.n_map_rejected
                totals[2] += last.n_reverted_bins
        for _, s in seqs:
            r_ = s.g.last_result()
            totals[3] += r_.n_map_out
            totals[4] += r_.n_static
            totals[5] += r_.n_dynamic
    elif interleave == "threads":
        import threading
        per_seq = [None] * len(seqs)

        def drive(i, s):
            t = np.zeros(6, np.int64)
            r_ = None
            for k in range(W, W + K):
                r_ = s.run(k)  # (ctypes releases the GIL for the call: the host-side enqueue of the sequences runs in parallel)
                t[0] += 1
                t[1] += r_.n_map_rejected
                t[2] += r_.n_reverted_bins
            t[3], t[4], t[5] = r_.n_map_out, r_.n_static, r_.n_dynamic
            per_seq[i] = (t, r_)

        grp = args.interleave_group if args.interleave_group > 0 else len(seqs)
        for g0 in range(0, len(seqs), grp):  # (handles were created in this order: the members of a group hold neighbouring compute pipes)
            ths = [threading.Thread(target=drive, args=(i, s)) for i, (_, s) in list(enumerate(seqs))[g0:g0 + grp]]
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join()
        for t, r_ in per_seq:
            totals += t
            last = r_
    else:
        # seq-per-gpu with --placement queue and several ranks: every rank has prepared ALL sequences; a shared counter (ed.JobQueue) hands
        # the next one to whichever rank is idle.  Otherwise: the sequences this rank owns, one after the other.
        jq = ed.JobQueue(dist, len(seqs)) if (args.mode == "seq-per-gpu" and args.placement == "queue" and world_size > 1) else None
        order = iter(range(len(seqs)))
        ran = []
        while True:
            si = jq.next() if jq is not None else next(order, None)
            if si is None:
                break
            s = seqs[si][1]
            ran.append(seqs[si][0])
            for k in range(W, W + K):
                if si == 0:
                    split_bytes.append(s.g.voi_split_bytes())
                last = s.run(k)  # synchronous: returns after the step's results are on the host
                if si == 0:
                    step_results.append(last)
                totals[0] += 1
                totals[1] += last.n_map_rejected
                totals[2] += last.n_reverted_bins
            totals[3] += last.n_map_out
            totals[4] += last.n_static
            totals[5] += last.n_dynamic
        if jq is not None:
            seqs = [sq for sq in seqs if sq[0] in ran] or seqs[:1]  # (what this rank ran: the evaluation / reporting below is about those)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed_local = time.perf_counter() - t_start
    pass_elapsed.append(elapsed_local)  # (the last -- or only -- pass)
    gpu_final = None
    if world_size == 1 and (not args.no_cpu_baseline or not args.no_pr_rr) and args.mode == "replicas" and first is not None:
        gpu_final = (first.g.get_map(), first.g.get_rejected_indices())  # (after the clock has stopped; compared with the oracle's below)
    prof = first.g.profile_get() if first is not None else {}
    chain_us = first.g.chain_timing() if first is not None else (0.0, 0.0, 0, 0.0)
    if first is not None:
        first.g.profiling(0)
    # per pass: the MAX over ranks; the figure reported is the median pass (min / max beside it)
    pass_s = [ed.max_over_ranks(dist, e_, dev) for e_ in pass_elapsed]
    elapsed = float(np.median(pass_s))
    elapsed_local = float(np.median(pass_elapsed))
    # ---- result exchange: one RCCL all_gather of the per-rank counters ----
    gathered = ed.gather_counts(dist, world_size, list(totals) + [int(elapsed_local * 1e6)], dev)
    rccl_ranks = len(gathered)

    evals = None
    if args.eval and args.mode == "seq-per-gpu":
        from erasor_amd import evalmap
        mine = []
        for sid, s in seqs:
            est = s.g.voxelize_preserving_labels(s.g.get_map(), 0.2)
            gt = s.g.voxelize_preserving_labels(maps[sid], 0.2)
            ev = evalmap.evaluate_clouds(gt, est, 0.2)
            mine.append([int(sid), int(round(ev["PR"] * 1e2)), int(round(ev["RR"] * 1e2))])
        while len(mine) < (len(SEQS) + world_size - 1) // world_size:
            mine.append([-1, 0, 0])
        evals = [e for per in ed.gather_counts(dist, world_size, [v for row in mine for v in row], dev)
                 for e in np.array(per).reshape(-1, 3).tolist() if e[0] >= 0]

    # ---- what scan-parallel replicas can exchange -- the union of their removals (a DEVIATION from the sequential fold)
    union_eval = None
    if args.union_eval > 0 and args.mode == "replicas":
        from erasor_amd import evalmap
        s0 = first
        nU = min(args.union_eval, s0.n_frames)
        removed, _ = ed.jacobi_removed_indices(s0.g, lambda: s0.g.set_map_device(d_map.data_ptr(), int(d_map.shape[0])), s0.scans[:nU], s0.c_Tl,
                                               s0.c_Tb[:nU], s0.c_To[:nU], device_scans=s0.d_ptr[:nU])
        per_rank = ed.allgather_indices(dist, world_size, removed, dev)  # ONE RCCL all_gather of the removed initial-map indices
        if rank == 0:
            united, union = ed.united_static_map(m, per_rank)
            # the sequential fold over rank 0's shard (what the reference does with these scans), same handle
            s0.g.set_map_device(d_map.data_ptr(), int(d_map.shape[0]))
            for k in range(nU):
                s0.g.step_device(s0.d_ptr[k], s0.n_pts[k], s0.c_Tl, s0.c_Tb[k], s0.c_To[k])
            seq_map = s0.g.get_map()
            gt = s0.g.voxelize_preserving_labels(m, 0.2)
            ev_u = evalmap.evaluate_clouds(gt, s0.g.voxelize_preserving_labels(united, 0.2), 0.2)
            ev_s = evalmap.evaluate_clouds(gt, s0.g.voxelize_preserving_labels(seq_map, 0.2), 0.2)
            union_eval = {"scans_per_rank": nU, "ranks": world_size, "removed_per_rank": [int(len(a)) for a in per_rank], "union": int(len(union)),
                          "united_map_points": int(len(united)),
                          "united_PR_RR": [round(ev_u["PR"], 3), round(ev_u["RR"], 3)],
                          "sequential_rank0_shard_PR_RR": [round(ev_s["PR"], 3), round(ev_s["RR"], 3)],
                          "note": "Jacobi-style DEVIATION from the reference's sequential fold (OfflineMapUpdater.cpp:290 -> :393): every scan sees "
                                  "the initial map, nothing a scan adds enters the united map; reported beside the sequential result, not instead of it"}
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    total_steps = int(sum(g_[0] for g_ in gathered)) // len(pass_s)  # scans all ranks processed in ONE pass
    value = total_steps / elapsed
    ms_per_step = elapsed / max(K * len(seqs), 1) * 1e3  # this rank's wall per step
    g = first.g
    P = first.P
    N_map = int(first.g.map_size())
    n_scan = first.n_scan
    # ---- roofline of the dominant kernel (voi_split): bytes its layout must stream per launch / measured launch time ----
    # k_voi_split is launched with its own start / stop HIP events (hipExtLaunchKernelGGL) on the handle's stream during the
    # timed region: they stamp the kernel's execution window, which is also what rocprofv3 reports (profiles/).
    vs_ms, vs_n = prof.get("voi_split", (0.0, 0))
    avg_ms = vs_ms / max(vs_n, 1)
    alg_bytes = float(np.mean([b for b, _ in split_bytes])) if split_bytes else 0.0
    entries = float(np.mean([e for _, e in split_bytes])) if split_bytes else 0.0
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # traffic / rocprofv3 average: NOT measured in this run — read from the committed separate rocprofv3 passes (profiles/), and
    # only while they describe THIS build (tools/collect_profiles.sh records a hash of the device sources)
    prof_tag = "" if args.workload == "seq05" else "_" + args.workload
    cp = committed_profile(prof_tag)
    traffic, rocprof_avg = cp["traffic"], cp["voi_split_avg_us"]
    # SURVEY §8(d) step-level figure: ALG_BYTES = 16*N_map + 16*n_scan + 16*N_voi_out (write-back of the updated VoI region)
    n_voi_out = int(last.n_static_estimate + last.n_complement)
    step_alg = 16.0 * N_map + 16.0 * n_scan + 16.0 * n_voi_out
    step_gbps = step_alg / (ms_per_step * 1e-3) / 1e9
    in_l3 = alg_bytes < 256 * 2**20
    src_note = ("STALE: the device sources changed since profiles/ was collected (tools/collect_profiles.sh) -> null" if cp["stale"]
                else "from_profiles (separate rocprofv3 passes of this very build, profiles/*_latest%s.*)" % prof_tag)
    roofline = {"bound": "l3+hbm" if in_l3 else "hbm", "kernel": "k_voi_split",
                "kernel_role": "hbm_kernel: the step's one pass over the map store (fetch_VoI membership) -- NOT the kernel that dominates GPU "
                               "time, see dominant_kernel and step_frac.  Since round 3 the pass skips the outskirts chunks whose bounding box "
                               "lies outside the VoI circle (see bytes_note): it reads a sixth of what it streamed and is latency-bound as a "
                               "launch; no kernel of the step is HBM-bound any more (largest consumers by PMC traffic: profiles/pmc_latest*.json)",
                "bound_note": ("the %.0f MB this launch reads fit the 256 MiB Infinity Cache: the rate is an L3+HBM figure; "
                               "ERASOR_HIP_NO_OMETA=1 --workload large_scale_05 (378 MB per launch, every chunk read) is the HBM-only streaming "
                               "measurement" % (alg_bytes / 1e6)) if in_l3
                              else "%.0f MB per launch, beyond the 256 MiB Infinity Cache: HBM-bound" % (alg_bytes / 1e6),
                "achieved": round(achieved, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": round(achieved / PEAK_HBM_GBPS, 4), "traffic": traffic,
                "step_traffic": cp["step_traffic"],  # every kernel of the PMC pass / its steps, corrected like `traffic` (profiles/pmc_latest*.json)
                "step_traffic_steps_only": cp["step_traffic_steps_only"],  # ... without the kernels that run once per pass (set_map)
                "traffic_source": src_note, "dominant_kernel": cp["dominant"], "critical_kernel": cp["critical"],
                "bytes_per_launch": int(alg_bytes),
                "bytes_note": "what a launch has to READ: float4 of the VoI-resident part + {x,y} pairs (8 B) of the outskirts chunks whose bounding "
                              "box meets the VoI circle + a 32-byte record per outskirts chunk + masks; chunks outside the circle are skipped by "
                              "their record (round 3) -- full_stream_bytes is what the same pass streamed before, full_stream_equiv_GBps that figure "
                              "over this launch's time (NOT an HBM rate: most of those bytes are never touched); SURVEY §8(d)'s 16 B/pt assumed "
                              "an AoS map (aos16_equiv_GBps is the rate in that currency)",
                "full_stream_bytes": int(8 * (N_map - int(last.n_voi)) + 16 * int(last.n_voi)),
                "full_stream_equiv_GBps": round((8 * (N_map - int(last.n_voi)) + 16 * int(last.n_voi)) / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else 0.0,
                "bound_after_skipping": "latency: per wavefront a record load, the decision, then 16 loads -- 10 k wavefronts, ~12 us for ~15 MB",
                "entries_per_launch": int(entries), "avg_launch_us": round(avg_ms * 1e3, 2),
                "rocprofv3_kernel_avg_us": rocprof_avg, "rocprofv3_source": src_note,
                "launches": int(vs_n), "launches_note": "every fourth k_voi_split launch of the timed region carries the start / stop events",
                "aos16_equiv_GBps": round(16.0 * entries / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else 0.0,
                "working_set_vs_L3": "%.0f MB streamed per launch vs 256 MiB Infinity Cache" % (alg_bytes / 1e6),
                "step_alg_bytes": int(step_alg), "step_achieved": round(step_gbps, 1), "step_frac": round(step_gbps / PEAK_HBM_GBPS, 4),
                "step_note": "SURVEY §8(d): (16*N_map + 16*n_scan + 16*N_voi_out) / ms_per_step — the whole step against the HBM peak.  CAVEAT "
                             "(VERDICT r03): 16*N_map is NOT read any more (the VoI pass skips the outskirts chunks whose bounding box lies outside "
                             "the circle), so this fraction flatters a step that is bound by latency, not bandwidth: read needed_bytes / "
                             "time_target_us and main_chain_us instead",
                # what the step has to move WITH the chunk records: the VoI pass as it is, the scan, the VoI-resident region read once and
                # written back once -- and the time 60 % of the HBM peak would need for that, beside north_star's own target (the §8(d) bytes)
                "needed_bytes": int(alg_bytes + 16.0 * n_scan + 16.0 * int(last.n_voi) + 16.0 * n_voi_out),
                "time_target_us": round((alg_bytes + 16.0 * n_scan + 16.0 * int(last.n_voi) + 16.0 * n_voi_out) / (0.6 * PEAK_HBM_GBPS * 1e9) * 1e6, 2),
                "north_star_time_target_us": round(step_alg / (0.6 * PEAK_HBM_GBPS * 1e9) * 1e6, 1),
                "time_note": "time_target_us = needed_bytes at 60 % of the HBM peak; north_star_time_target_us = SURVEY §8(d)'s bytes at the same "
                             "rate (what north_star asks of a step); the step takes ms_per_step: a dependency chain of ~10 launches "
                             "(main_chain_us on the device's clock), every one bound by latency -- exact std::sort emulation, sequential "
                             "float32 sums, a single-lane SVD"}

    # ---- optional per-kernel breakdown (extra K steps, all kernels bracketed) ----
    if args.profile_all:
        g.profile_reset()
        g.profiling(1)
        for k in range(W, W + K):
            first.run(k)  # re-running the same scans against the already-updated map is fine for a time breakdown
        pa = g.profile_get()
        g.profiling(0)
        tot = sum(v[0] for v in pa.values())
        for name, (ms, cnt) in sorted(pa.items(), key=lambda kv: -kv[1][0]):
            print("  %-14s %8.3f ms/step  (%5.1f%%, %d launches)" % (name, ms / K, 100 * ms / max(tot, 1e-9), cnt // K), file=sys.stderr)
        print("  sum of kernels %.3f ms/step vs wall %.3f ms/step" % (tot / K, ms_per_step), file=sys.stderr)

    # ---- the same K scans again without look-ahead (every step runs its own query chain first): what a ROS callback sees ----
    sync_ms = None
    if first.lookahead and world_size == 1:
        first.lookahead = False
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for k in range(W, W + K):
            first.run(k)  # against the already-updated map: same work per step, results not used
        torch.cuda.synchronize()
        sync_ms = (time.perf_counter() - ts) * 1e3 / K
        first.lookahead = True

    # ---- CPU baseline: the reference's own sources (oracle/_ref) and the oracle port, one thread, bounded sample ----
    cpu = cpu_refsrc = None
    parity = {"parity_checked_steps": 0, "parity": "not checked in this run (no CPU leg)"}
    if world_size == 1 and not args.no_cpu_baseline and args.mode == "replicas":
        cpu, cpu_refsrc, parity = cpu_baseline(args, P, m, first, l2b7, step_results, gpu_final)
    elif world_size == 1 and not args.no_cpu_baseline:
        cpu = cpu_sequence_parallel(args, seqs, maps, l2b7)
    elif world_size == 1 and args.mode == "replicas" and args.verify_prefix_seconds > 0 and step_results:
        parity = verify_prefix(args, P, m, first, step_results)

    # ---- PR / RR of the map the timed pass has left (BASELINE.json: "PR/RR vs ref"): the reference's protocol (scripts/analysis_runner.py
    # :74-105 restated in erasor_amd/evalmap.py, pinned to the original's outputs by tests/test_evalmap.py) against the initial map as
    # labelled ground truth.  The oracle's map is bit-identical (final_map_checked), so its PR / RR are these very numbers.
    pr_rr = None
    if world_size == 1 and args.mode == "replicas" and not args.no_pr_rr and m is not None and gpu_final is not None:
        from erasor_amd import evalmap
        t_ev = time.time()
        est = g.voxelize_preserving_labels(gpu_final[0], 0.2)
        gt = g.voxelize_preserving_labels(m, 0.2)
        ev = evalmap.evaluate_clouds(gt, est, 0.2)
        pr_rr = {"PR": round(ev["PR"], 3), "RR": round(ev["RR"], 3), "F1": round(ev["F1"], 4), "gt_static": ev["gt_static"], "gt_dynamic": ev["gt_dynamic"],
                 "protocol": "scripts/analysis_runner.py:74-105 (1-NN within voxel*sqrt(3)/2 of every ground-truth point), both maps voxelised at 0.2 m "
                             "by voxelize_preserving_labels (save_static_map, OfflineMapUpdater.cpp:174-196); ground truth = the labelled initial map",
                 "vs_reference": ("identical: the %d-point map compared here is bit-identical to the CPU path's (final_map_checked)" % len(gpu_final[0]))
                 if parity.get("final_map_checked") else "the CPU path's map was not compared in this run",
                 "steps": int(K * len(pass_s) + W), "wall_s": round(time.time() - t_ev, 1)}

    # ---- the drop-in path in C++: erasor::OfflineMapUpdater::callback_node with host pcl::PointXYZI clouds in and the
    # rejected clouds out, timed by erasor_offline_demo --bench on an export of this very workload (no Python in its loop)
    callback = None
    demo = os.path.join(ROOT, "erasor_amd", "erasor_offline_demo")
    if (world_size == 1 and args.mode == "replicas" and args.workload == "seq05" and not args.no_callback_bench and not args.no_cpu_baseline
            and os.path.exists(demo)):
        import subprocess
        import tempfile
        t_cb = time.time()
        try:
            with tempfile.TemporaryDirectory(prefix="erasor_cppbench_") as d:
                subprocess.run([sys.executable, os.path.join(ROOT, "tools", "export_cpp_bench.py"), d, str(K + W + 8)], check=True, capture_output=True, timeout=300)
                r = subprocess.run([demo, "--bench", d, str(K), str(W)], capture_output=True, text=True, timeout=300)
                j = json.loads(r.stdout.strip().split("\n")[-1])
            callback = {"callback_ms": j["ms_per_callback"], "callback_next_announced_ms": j["ms_per_callback_next_node_announced"],
                        "of_which_announce_next_ms": j["of_which_announce_next"],
                        "callback_next_announced_deferred_ms": j.get("ms_per_callback_next_node_announced_deferred"),
                        "callback_nodes_announced_deep_ms": j.get("ms_per_callback_nodes_announced_deep"), "deep_lookahead": j.get("deep_lookahead"),
                        "c_abi_device_resident_two_ahead_ms": j["ms_per_step_device_resident_two_ahead"], "passes_agree": j.get("passes_agree"),
                        "note": j["callback_note"] + "; one node ahead is all a callback can know, so the query chain of the next node (~0.25 ms alone) "
                                "bounds this path, not the main chain", "wall_s": round(time.time() - t_cb, 1)}
        except Exception as e:  # (never takes the headline down)
            callback = {"error": str(e)[:300]}

    out = {
        "metric": "scans_per_sec", "value": round(value, 2), "unit": "scans/s", "n_gpus": world_size, "steps": K, "warmup": W,
        "ms_per_step": round(ms_per_step, 4), "repeats": len(pass_s),
        "ms_per_step_min": round(min(pass_s) / max(K * len(seqs), 1) * 1e3, 4), "ms_per_step_max": round(max(pass_s) / max(K * len(seqs), 1) * 1e3, 4),
        "ms_per_step_all": [round(e_ / max(K * len(seqs), 1) * 1e3, 4) for e_ in pass_s],
        # (rounds 1-4 reported ONE pass -- this is that pass, the first K steps behind the warm-up; cross-round comparisons of the
        # headline should say which of the two statistics they use)
        "ms_per_step_first_pass": round(pass_s[0] / max(K * len(seqs), 1) * 1e3, 4),
        "repeats_note": "the K-step timed pass (barrier + device synchronisation on both sides, MAX over ranks) run `repeats` times in this one "
                        "invocation, the sequence continuing from pass to pass; ms_per_step / value are the median pass",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (transforms, R-GPF) + f64 (VoI test, polar binning, scan ratio)", "data": "synthetic",
        "config": {"workload": "%s: %s; %d-pt map resident in HBM, ~%d-pt scans, ERASOR v%d; one scan per step, 1 m/frame"
                               % (args.workload, wl["desc"], N_map, n_scan, P.version),
                   "map_points": N_map, "scan_points": n_scan, "rings": int(P.num_rings), "sectors": int(P.num_sectors),
                   "max_range": float(P.max_range), "is_large_scale": int(P.is_large_scale), "mode": args.mode,
                   "sharding": ("scan-parallel replicas, one RCCL broadcast of the map, no data-path collective" if args.mode == "replicas"
                                else "one KITTI-shaped sequence per GPU (00/01/02/05/07 dealt round-robin), no map exchange"),
                   "lookahead_scans": first.LA if first.lookahead else 0,
                   "chain_batch": {"n_scans": args.chain_batch, "lead": args.chain_lead},
                   "sequences_on_this_rank": len(seqs), "interleave": interleave,
                   "node_loop": "native (erasor_hip_run_nodes: one call for the timed nodes)" if native_loop else "python (prefetch + step per node)"},
        "map_points_x_scans_per_sec": round(value * N_map, 1),
        "ms_per_step_without_lookahead": None if sync_ms is None else round(sync_ms, 4),
        # the main stream's dependency chain in the timed pass, on the device's own clock (chunk scan .. the step's end), and the time
        # the stream spends between two steps (host turnaround; the next step's VoI split, launched ahead, runs in there)
        "main_chain_us": round(chain_us[0], 1), "between_steps_us": round(chain_us[1], 1),
        "steady_state_ms_per_step": round(chain_us[3] / 1e3, 4),
        "steady_state_note": "chunk scan to chunk scan of consecutive steps on the device's clock: the period of the main stream in the timed pass "
                             "(main_chain_us contains the stream's wait for the host since the scan is launched ahead).  ms_per_step (the contract's figure: "
                             "K steps between two device synchronisations) also pays for draining the query chains of the nodes announced beyond "
                             "the last timed step, once per pass",
        # round 5: steps whose split / chunk scan / gather / bucket table were launched beside the previous step's per-bin launch, and how many
        # of them the step took (ERASOR_HIP_OVERLAP unset = the handle measures both modes and keeps the faster, 1 = always, 0 = never)
        "overlapped_steps": dict(zip(("launched_ahead", "taken"), g.overlap_counts()), mode=os.environ.get("ERASOR_HIP_OVERLAP", "auto"),
                                 auto=dict(zip(("mode_now", "plain_period_us", "overlapped_period_us"), g.overlap_auto()))),
        # round 6: sets of launches shared by the query chains of several announced nodes, and the chains that went into them
        "shared_chain_launches": dict(zip(("sets", "chains"), g.chain_batch_counts())),
        "pr_rr": pr_rr, "callback_path": callback,
        "roofline": roofline, "cpu_baseline": cpu, "cpu_reference_sources": cpu_refsrc, "host": host_identity(),
        "parity_checked_steps": parity["parity_checked_steps"], "parity": parity.get("parity"), "final_map_checked": parity.get("final_map_checked", False),
        "rccl_ranks": rccl_ranks, "backend": backend if dist is not None else None,
        "per_rank": [{"rank": i, "steps": int(v[0]), "map_rejected": int(v[1]), "reverted_bins": int(v[2]), "final_map_points": int(v[3]),
                      "static": int(v[4]), "dynamic": int(v[5]), "wall_ms": round(v[6] / 1e3, 2)} for i, v in enumerate(gathered)],
        "last_step": last.as_dict() if last is not None else None,
        "setup_s": {"map_build_and_upload": round(t_map, 2), "rccl_broadcast": None if t_bcast is None else round(t_bcast, 4),
                    "rccl_broadcast_bytes": None if t_bcast is None else 16 * N_map},
    }
    if union_eval is not None:
        out["union_exchange"] = union_eval
    if evals is not None:
        out["pr_rr"] = [{"seq": "%02d" % e[0], "PR": e[1] / 1e2, "RR": e[2] / 1e2} for e in sorted(evals)]  # (seq-per-gpu --eval: one entry per sequence)
    # ---- the other single-GPU BASELINE configs, measured in the same run (short passes, each its own process: a clean device state
    # and exactly the code path above).  config 4 is the only workload whose VoI pass streams more than the 256 MiB Infinity Cache.
    if (world_size == 1 and args.workload == "seq05" and args.mode == "replicas" and not args.no_extra_workloads and not args.no_cpu_baseline
            and not args.profile_all):
        import subprocess
        extra = []
        passes = (
            # (workload, label, extra arguments, environment, verified against the oracle?)
            ("large_scale_05", "config 4 (39 M-point dense map), is_large_scale off -- the pass that is VERIFIED against the oracle, and shorter for "
                               "it (12 timed steps: the first steps of a sequence revert more bins)", ["--cpu-seconds", "24", "--cpu-steps", "1"], {}, True),
            ("large_scale_05", "config 4, is_large_scale off, timed like the headline (20 + 5 steps)", [], {}, False),
            ("large_scale_05", "config 4, --large-scale-mode on (submap 160, OfflineMapUpdater.cpp:332-379)", ["--large-scale-mode", "on"], {}, False),
            ("large_scale_05", "config 4 with the chunk records OFF (ERASOR_HIP_NO_OMETA=1): the VoI pass streams the whole map store -- "
                               "the HBM-bound measurement of k_voi_split", [], {"ERASOR_HIP_NO_OMETA": "1"}, False),
            ("seq05_yaml", "config 2, config/seq_05.yaml verbatim", [], {}, False),
            ("ouster128", "config 5 shape, 1 GPU", [], {}, False),
        )
        for wname, label, xargs, xenv, verified in passes:
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", wname, "--steps", "12" if verified else "20", "--warmup",
                   "3" if verified else "5", "--no-extra-workloads", "--no-pr-rr", "--no-callback-bench", "--repeats", "1" if verified else "7"] + xargs
            if not verified:
                cmd += ["--no-cpu-baseline", "--verify-prefix-seconds", "4"]  # (the first steps of the pass against the oracle, 4 s of CPU)
            t_sub = time.time()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=dict(os.environ, **xenv))
                d = json.loads(r.stdout.strip().split("\n")[-1])
                rf = d["roofline"]
                extra.append({"workload": wname, "baseline_config": label, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                              "ms_per_step_without_lookahead": d["ms_per_step_without_lookahead"], "main_chain_us": d.get("main_chain_us"),
                              "repeats": d.get("repeats"), "ms_per_step_min": d.get("ms_per_step_min"), "ms_per_step_max": d.get("ms_per_step_max"),
                              "overlapped_steps": d.get("overlapped_steps"),
                              "between_steps_us": d.get("between_steps_us"), "steps": d["steps"], "warmup": d["warmup"],
                              "map_points": d["config"]["map_points"], "scan_points": d["config"]["scan_points"],
                              "is_large_scale": d["config"].get("is_large_scale"), "lookahead_scans": d["config"].get("lookahead_scans"),
                              "environment": xenv,
                              "parity_checked_steps": d.get("parity_checked_steps"), "final_map_checked": d.get("final_map_checked"), "parity": d.get("parity"),
                              "cpu_baseline": d.get("cpu_baseline"), "cpu_reference_sources": d.get("cpu_reference_sources"),
                              "roofline": {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_launch", "avg_launch_us",
                                                                  "launches", "step_alg_bytes", "step_achieved", "step_frac", "needed_bytes",
                                                                  "time_target_us")},
                              "wall_s": round(time.time() - t_sub, 1)})
            except Exception as e:  # (an extra pass must never take the headline line down with it)
                extra.append({"workload": wname, "baseline_config": label, "error": (str(e) + " | " + (r.stderr[-300:] if "r" in dir() else ""))[:500]})
        out["other_workloads"] = extra
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
