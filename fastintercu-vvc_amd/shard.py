"""Multi-GPU sharding of the hot path: one process per GPU, no hot-path collective.

CU inferences are independent, so a batch is cut into contiguous rank-local ranges
(GPU g gets CUs [g*B/G, (g+1)*B/G)) and every rank runs its own `MltCnn` context.  The only exchange steps are
  * init: rank 0 broadcasts the MLTW weight blob (RCCL over xGMI on GPUs, `backend="nccl"`; gloo in the CPU tests),
  * init: every rank calibrates its own copy of the weights -- the ranks then AGREE on the outcome (rank 0's mlt_arith_info is broadcast and
    compared, `agree_on_arithmetic`): an N-rank line cannot mix arithmetic tiers silently,
  * optional: all_gather of the int32 split modes / fp32 logits when one rank needs the whole batch.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous, balanced split: sizes differ by at most one CU."""
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    return lo, hi


def broadcast_blob(blob, dist, device, src: int = 0) -> bytes:
    """Every rank returns the blob held by `src` (others may pass None)."""
    import torch
    rank = dist.get_rank()
    ln = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(ln, src)
    buf = torch.empty(int(ln.item()), dtype=torch.uint8, device=device)
    if rank == src:
        buf.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    dist.broadcast(buf, src)
    return buf.cpu().numpy().tobytes()


def gather_results(split: np.ndarray, logits: np.ndarray, total: int, dist, device):
    """all_gather of ragged rank-local results -> full [total] split modes and [total, L] logits on every rank."""
    import torch
    world = dist.get_world_size()
    cap = max(shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world))
    L = logits.shape[1]
    s_pad = torch.full((cap,), -1, dtype=torch.int32, device=device)
    l_pad = torch.zeros((cap, L), dtype=torch.float32, device=device)
    s_pad[:len(split)] = torch.from_numpy(split).to(device)
    l_pad[:len(split)] = torch.from_numpy(logits).to(device)
    s_all = [torch.empty_like(s_pad) for _ in range(world)]
    l_all = [torch.empty_like(l_pad) for _ in range(world)]
    dist.all_gather(s_all, s_pad)
    dist.all_gather(l_all, l_pad)
    out_s = np.empty((total,), np.int32)
    out_l = np.empty((total, L), np.float32)
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        out_s[lo:hi] = s_all[r][:hi - lo].cpu().numpy()
        out_l[lo:hi] = l_all[r][:hi - lo].cpu().numpy()
    return out_s, out_l


# what identifies the arithmetic a rank's load-time calibration chose (mlt_arith_info): two ranks that differ in any of these would produce
# different bits for the same CU -- and different throughput -- inside one job
ARITH_KEYS = ("exact", "w2_stages", "w2_units", "x_stages", "x_units", "rounding", "flat_guard", "decision_guard", "guard_margin", "mag_guard_thr", "mag_guard_kind")


def agree_on_arithmetic(arith: dict, dist, src: int = 0) -> dict:
    """Rank `src` broadcasts the arithmetic its calibration chose (the ARITH_KEYS of MltCnn.arithmetic()); every rank compares its own with it
    and raises on a mismatch (after an all_gather of the verdicts, so that ALL ranks fail together instead of some hanging in the next
    collective).  Returns the agreed dict.  The calibration is deterministic (same blob, same kernels, same synthetic CUs), so a mismatch
    means different library builds, different GPUs or a faulty device -- in every case a line that must not be reported as one job."""
    mine = {k: (round(float(arith[k]), 9) if isinstance(arith[k], float) else int(arith[k])) for k in ARITH_KEYS}
    box = [mine if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src)
    ref = box[0]
    ok = mine == ref
    verdicts = [None] * dist.get_world_size()
    dist.all_gather_object(verdicts, (dist.get_rank(), ok, mine))
    bad = [(r, m) for r, good, m in verdicts if not good]
    if bad:
        raise RuntimeError(f"ranks calibrated to different arithmetic tiers: rank {src} has {ref}, but " + "; ".join(f"rank {r} has {m}" for r, m in bad))
    return ref
