"""CPU, world_size 2 over gloo: the N > 1 path -- weight-blob broadcast, contiguous batch sharding, result gather.
The per-rank compute is stood in for by the CPU oracle (test infrastructure); on GPUs each rank runs MltCnn instead
(bench.py) with the same shard / broadcast / gather code."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, size, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    import mltcnn_pkg
    import oracle
    pkg = mltcnn_pkg.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    arch = pkg.synth.arch_for_size(size)
    blob = pkg.shard.broadcast_blob(pkg.weights.synthetic_blob(arch, 10) if rank == 0 else None, dist, dev)
    lo, hi = pkg.shard.shard_bounds(total, rank, world)
    org, pred = pkg.synth.make_patches(size, hi - lo, 31337, first=lo)          # rank-local shard of the global batch
    poc, qp = pkg.synth.make_scalars(total, 31337)
    logits, split = oracle.Oracle(blob).forward(org, pred, poc[lo:hi], qp[lo:hi], threads=2)
    full_s, full_l = pkg.shard.gather_results(split, logits, total, dist, dev)
    if rank == 0:
        import hashlib
        q.put((hashlib.sha256(blob).hexdigest(), full_s, full_l))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_batch_matches_single_process(pkg):
    import torch.multiprocessing as mp
    import hashlib
    import oracle
    total, size, world = 7, 32, 2  # odd on purpose: shards of 3 and 4 CUs
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, size, q)) for r in range(world)]
    for p in procs:
        p.start()
    sha, full_s, full_l = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    blob = pkg.weights.synthetic_blob(pkg.synth.arch_for_size(size), 10)
    assert sha == hashlib.sha256(blob).hexdigest(), "broadcast blob differs from rank 0's"
    org, pred = pkg.synth.make_patches(size, total, 31337)
    poc, qp = pkg.synth.make_scalars(total, 31337)
    ref_l, ref_s = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=2)
    assert np.array_equal(full_s, ref_s) and np.array_equal(full_l, ref_l)


@pytest.mark.parametrize("total,world", [(4096, 8), (7, 2), (5, 8), (0, 4)])
def test_shard_bounds_partition(pkg, total, world):
    b = [pkg.shard.shard_bounds(total, r, world) for r in range(world)]
    assert b[0][0] == 0 and b[-1][1] == total
    assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    sizes = [hi - lo for lo, hi in b]
    assert max(sizes) - min(sizes) <= 1


def _agree_worker(rank, world, port, differ, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    import mltcnn_pkg
    pkg = mltcnn_pkg.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    arith = {"exact": 0, "w2_stages": 0, "w2_units": 0, "x_stages": 0, "x_units": 0, "rounding": 0, "flat_guard": 1, "decision_guard": 1,
             "guard_margin": 3e-3, "mag_guard_thr": 31.5, "mag_guard_kind": 1, "calib_rms": 1.6e-4 + 1e-6 * rank, "guard_reruns": rank}   # (figures outside ARITH_KEYS may differ)
    if differ and rank == 1:
        arith["exact"], arith["w2_units"] = 3, 0xC   # this rank's calibration chose hi+lo weights in layer1
    try:
        pkg.shard.agree_on_arithmetic(arith, dist)
        q.put((rank, "agreed"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("differ", [False, True])
def test_ranks_agree_on_the_calibrated_arithmetic(pkg, differ):
    """rank 0's mlt_arith_info is broadcast and compared -- an 8-rank bench line cannot mix arithmetic tiers silently; a
    mismatch fails on EVERY rank (none is left hanging in the next collective)."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, differ, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if differ:
        assert all("different arithmetic tiers" in v and "rank 1 has" in v for v in got.values()), got
    else:
        assert got == {0: "agreed", 1: "agreed"}
