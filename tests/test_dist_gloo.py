"""world_size-2 CPU test (gloo) of the N>1 path bench.py uses: map broadcast from rank 0, per-rank scan shards,
MAX-over-ranks timing, result gather.  The compute itself needs a GPU (no CPU fallback) and is covered by -m gpu."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, zlib
    import numpy as np
    sys.path.insert(0, %r)
    import torch
    from erasor_amd import dist as ed
    dist, world, rank, local_rank = ed.init("gloo")
    assert world == 2 and dist is not None
    dev = torch.device("cpu")
    m = None
    if rank == 0:
        m = np.random.default_rng(5).normal(size=(12345, 4)).astype(np.float32)
    t = ed.broadcast_map(dist, rank, dev, m)
    want = np.random.default_rng(5).normal(size=(12345, 4)).astype(np.float32)
    assert t.shape == (12345, 4) and np.array_equal(t.numpy().view(np.uint32), want.view(np.uint32)), "replica differs from rank 0's map"
    x0, frames = ed.shard_frames(rank, world, 7)
    shards = ed.gather_counts(dist, world, [int(x0)] + frames, dev)
    assert shards[0] != shards[1] and shards[0][1:] == shards[1][1:] == list(range(7))   # disjoint streams, equal work
    mine = ed.deal_round_robin(5, rank, world)   # seq-per-gpu mode: five sequences dealt over the ranks
    dealt = ed.gather_counts(dist, world, mine + [-1] * (3 - len(mine)), dev)
    assert sorted(v for per in dealt for v in per if v >= 0) == [0, 1, 2, 3, 4]
    tmax = ed.max_over_ranks(dist, 1.0 + rank, dev)
    assert tmax == 2.0
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank" + str(rank) + "-ok" + chr(10))
''') % ROOT


def test_two_rank_gloo_broadcast_shard_and_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "rank0-ok" in out.stdout and "rank1-ok" in out.stdout


def test_single_process_path_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from erasor_amd import dist as ed
    import torch
    import numpy as np
    os.environ.pop("WORLD_SIZE", None)
    d, world, rank, _ = ed.init("gloo")
    assert d is None and world == 1 and rank == 0
    m = np.arange(40, dtype=np.float32).reshape(10, 4)
    t = ed.broadcast_map(None, 0, torch.device("cpu"), m)
    assert np.array_equal(t.numpy(), m)
    assert ed.max_over_ranks(None, 3.5, torch.device("cpu")) == 3.5
