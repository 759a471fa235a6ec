"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).

ERASOR's scans are a sequential fold over ONE map (scan k+1 reads the map scan k wrote, reference
OfflineMapUpdater.cpp:290 -> :393), so the only sharding without changing semantics is across
independent replicas / sequences.  The path therefore has exactly one exchange step: the RCCL broadcast
of the global map from rank 0 over xGMI (160 MB for a 10 M-point map), then zero per-scan communication,
then a MAX-reduce of the wall time (bench) or a gather of per-rank results.

SURVEY 8(e)(ii), the result exchange of scan-parallel replicas: when N replicas work on scans of ONE sequence at the same
time they cannot fold the map the way the reference does (scan k+1 would have to see what scan k removed).  What they can
do is the Jacobi-style variant -- every scan is applied to the SAME initial map, each yields the initial-map indices of the
points it rejects (erasor_hip_get_rejected_indices on a freshly set map), the ranks exchange their index lists with ONE
all_gather and the union is removed from the initial map (`jacobi_removed_indices`, `allgather_indices`, `united_static_map`).
That is a DEVIATION from the reference's sequential fold -- nothing a scan adds (voxelised reverted bins, OMU.cpp:281-290)
enters the united map, and a scan never sees another scan's removals -- so it is reported beside the sequential result
(PR / RR of both, bench.py --union-eval), never instead of it.
"""
import os

import numpy as np


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend):
    """init_process_group from the torchrun environment; returns (dist module or None, world, rank, local_rank)"""
    world, rank, local_rank = env_world()
    if world <= 1:
        return None, 1, 0, local_rank
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist, world, rank, local_rank


def broadcast_map(dist, rank, device, map_np):
    """rank 0 holds map_np (N,4) float32; every rank gets a device tensor with the same bytes.
    Two collectives: the size (8 bytes), then the XYZI payload (16 B/point)."""
    import torch
    n = torch.tensor([0 if map_np is None else int(map_np.shape[0])], dtype=torch.int64, device=device)
    if dist is not None:
        dist.broadcast(n, src=0)
    N = int(n.item())
    t = torch.empty((N, 4), dtype=torch.float32, device=device)
    if rank == 0:
        t.copy_(torch.from_numpy(np.ascontiguousarray(map_np, dtype=np.float32)))
    if dist is not None:
        dist.broadcast(t, src=0)
    return t


def shard_frames(rank, world, frames_per_rank, stride=37):
    """The scan stream of replica `rank`: its own start offset along the trajectory (metres) and frame ids.
    Weak scaling: every rank processes `frames_per_rank` scans; shards never overlap in (offset, frame)."""
    return 300.0 + float(stride) * rank, list(range(frames_per_rank))


def deal_round_robin(n_items, rank, world):
    """indices of the items rank `rank` owns when n_items independent jobs (sequences) are dealt over `world` ranks"""
    return list(range(rank, n_items, world))


class JobQueue:
    """Independent jobs (sequences) handed out in order to whichever RANK asks next -- BASELINE config 3's "the fifth sequence onto
    the first free GPU"  instead of a static deal.  One shared counter in the process group's key-value store (an atomic
    add at rank 0's TCPStore: a few bytes per job, no collective, no data-path traffic); a single process counts locally.
    next() -> job index, or None when the jobs are gone.  Every rank must create its queues in the same order (the n-th queue of every
    rank is the same queue)."""
    _serial = 0

    def __init__(self, dist, n_jobs, store=None):
        self.n, self.dist, self._local = int(n_jobs), dist, 0
        JobQueue._serial += 1
        self.key = "erasor_job_queue_%d" % JobQueue._serial
        self.store = store
        if dist is not None and store is None:
            from torch.distributed import distributed_c10d
            self.store = distributed_c10d._get_default_store()

    def next(self):
        if self.dist is None:
            j = self._local
            self._local += 1
        else:
            j = int(self.store.add(self.key, 1)) - 1
        return j if j < self.n else None


def max_over_ranks(dist, value, device):
    import torch
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(dist, world, values, device):
    """all-gather a small int64 vector per rank (per-replica result sizes / metrics)"""
    import torch
    v = torch.tensor(list(values), dtype=torch.int64, device=device)
    if dist is None:
        return [v.tolist()]
    out = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(out, v)
    return [o.tolist() for o in out]


# ---- union of the replicas' removals (Jacobi-style deviation, see the module docstring) ----------------------
def jacobi_removed_indices(g, set_initial_map, scans, T_l2b, T_b2o, T_o2b, device_scans=None):
    """Every scan of `scans` against the SAME initial map: `set_initial_map()` puts the initial map back into handle `g`
    before each step (the step's pre-step indices are then initial-map indices).  Returns the sorted union of the
    rejected initial-map indices (uint64) and the number of steps run."""
    removed = []
    for k in range(len(scans)):
        set_initial_map()
        if device_scans is not None:
            g.step_device(device_scans[k], len(scans[k]), T_l2b, T_b2o[k], T_o2b[k])
        else:
            g.step(scans[k], T_l2b, T_b2o[k], T_o2b[k])
        removed.append(g.get_rejected_indices())
    if not removed:
        return np.zeros(0, np.uint64), 0
    return np.unique(np.concatenate(removed)).astype(np.uint64), len(scans)


def allgather_indices(dist, world, idx, device):
    """ONE exchange of variable-length index lists: sizes first (8 bytes per rank), then the lists padded to the longest.
    Returns the per-rank arrays (on every rank)."""
    import torch
    idx = np.ascontiguousarray(idx, np.int64)
    if dist is None:
        return [idx.copy()]
    n = torch.tensor([len(idx)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    mine = torch.full((cap,), -1, dtype=torch.int64, device=device)
    if len(idx):
        mine[: len(idx)] = torch.from_numpy(idx).to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [o[:sizes[r]].cpu().numpy() for r, o in enumerate(out)]


def united_static_map(initial_map, per_rank_indices):
    """initial map minus the union of every rank's removed indices; returns (united map, union)"""
    union = np.unique(np.concatenate([np.asarray(a, np.int64) for a in per_rank_indices])) if per_rank_indices else np.zeros(0, np.int64)
    keep = np.ones(len(initial_map), bool)
    keep[union] = False
    return np.ascontiguousarray(initial_map[keep]), union
