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
    # config 3 with a QUEUE instead of a deal (round 4): jobs go to whichever rank asks next; rank 1 is slow, rank 0 takes most of them
    import time
    q = ed.JobQueue(dist, 5)
    took = []
    while True:
        j = q.next()
        if j is None:
            break
        took.append(j)
        time.sleep(0.02 if rank == 0 else 0.5)
    got = ed.gather_counts(dist, world, took + [-1] * (5 - len(took)), dev)
    assert sorted(v for per in got for v in per if v >= 0) == [0, 1, 2, 3, 4], got     # every job exactly once
    assert len([v for v in got[0] if v >= 0]) > len([v for v in got[1] if v >= 0]), got  # the idle rank took the next job
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


UNION_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, %r)
    sys.path.insert(0, %r + "/tests")
    import torch
    import erasor_amd
    erasor_amd.LIB_PATH = os.environ["ERASOR_TEST_SIMT_LIB"]   # the device code on the CPU stand-in, for this test process only
    erasor_amd._lib = None
    from erasor_amd import dist as ed
    import scenarios
    from oracle import orc   # checker
    dist, world, rank, local_rank = ed.init("gloo")
    assert world == 2 and dist is not None
    dev = torch.device("cpu")
    sc = scenarios.small(n_frames=6, az=120, length=60.0)
    m0 = sc["map"] if rank == 0 else None
    m = ed.broadcast_map(dist, rank, dev, m0).numpy()                       # ONE broadcast of the global map
    g = erasor_amd.Erasor(scenarios.to_product_params(sc["params"]))
    n_scans = 4
    mine = list(range(rank, n_scans, world))                                 # scan-parallel shard of ONE sequence
    scans = [np.ascontiguousarray(sc["scans"][k], np.float32) for k in mine]
    removed, n = ed.jacobi_removed_indices(g, lambda: g.set_map(m), scans, sc["T_l2b"], [sc["T_b2o"][k] for k in mine], [sc["T_o2b"][k] for k in mine])
    assert n == len(mine)
    per_rank = ed.allgather_indices(dist, world, removed, dev)               # ONE exchange of the removed initial-map indices
    united, union = ed.united_static_map(m, per_rank)
    # single-process computation of the same thing with the oracle: every scan against the initial map, union of the masks
    want = []
    for k in range(n_scans):
        o = orc.Oracle(sc["params"])
        o.set_map(sc["map"])
        o.step(sc["scans"][k], sc["T_l2b"], sc["T_b2o"][k], sc["T_o2b"][k])
        want.append(o.get_rejected_indices())
        o.close()
    want = np.unique(np.concatenate(want)).astype(np.int64)
    assert len(want) > 0, "the scenario must reject something"
    assert np.array_equal(union, want), ("union of the two ranks' removals differs from the single-process one", len(union), len(want))
    assert len(united) == len(sc["map"]) - len(want)
    assert all(np.array_equal(per_rank[r], per_rank[r]) for r in range(world)) and sum(len(a) for a in per_rank) >= len(union)
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank" + str(rank) + "-union-ok " + str(len(union)) + chr(10))
''') % (ROOT, ROOT)


def test_two_ranks_run_real_steps_and_exchange_the_union_of_their_removals(tmp_path):
    """SURVEY 8(e)(ii) on two gloo ranks: rank 0's map is broadcast, each rank runs REAL steps (the device code on the CPU
    stand-in) on its shard of one sequence's scans -- every scan against the initial map (Jacobi-style, see erasor_amd/dist.py) --,
    ONE all_gather exchanges the removed initial-map indices, and the union equals a single-process oracle computation."""
    lib = str(tmp_path / "liberasor_hip_simt.so")
    subprocess.check_call(["g++", "-x", "c++", "-O1", "-std=c++20", "-pthread", "-ffp-contract=off", "-fPIC", "-shared",
                           "-I" + os.path.join(ROOT, "tests", "cpp", "simt_emu"), "-o", lib, os.path.join(ROOT, "erasor_amd", "csrc", "erasor_hip.hip")])
    sys.path.insert(0, ROOT)
    from oracle import orc
    orc.build()
    script = tmp_path / "union_worker.py"
    script.write_text(UNION_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ERASOR_TEST_SIMT_LIB=lib)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "rank0-union-ok" in out.stdout and "rank1-union-ok" in out.stdout


def test_job_queue_in_a_single_process():
    sys.path.insert(0, ROOT)
    from erasor_amd import dist as ed
    q = ed.JobQueue(None, 3)
    assert [q.next(), q.next(), q.next(), q.next(), q.next()] == [0, 1, 2, None, None]


def test_work_queue_of_the_cpp_driver(tmp_path):
    """erasor::WorkQueue (erasor_amd/csrc/shim/erasor_shim_queue.h), what erasor_offline_demo --queue hands sequences out with"""
    exe = str(tmp_path / "wq")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "work_queue_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "WORK-QUEUE-OK" in out.stdout, out.stdout + out.stderr


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
