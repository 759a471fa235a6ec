"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).

ERASOR's scans are a sequential fold over ONE map (scan k+1 reads the map scan k wrote, reference
OfflineMapUpdater.cpp:290 -> :393), so the only sharding without changing semantics is across
independent replicas / sequences.  The path therefore has exactly one exchange step: the RCCL broadcast
of the global map from rank 0 over xGMI (160 MB for a 10 M-point map), then zero per-scan communication,
then a MAX-reduce of the wall time (bench) or a gather of per-rank results.
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
