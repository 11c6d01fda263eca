"""Read sharding across GPUs and the one collective of the path: the final gather of fixed-size
result records.  Reads are independent units (the reference runs one OpenMP task
per read, src/hmmufotu.cpp:603-610); the database is replicated per GPU, so no data-path
collective exists.  Works with any torch.distributed backend ("nccl" = RCCL over xGMI on the
GPU node, "gloo" in the CPU tests)."""
from __future__ import annotations

import numpy as np


def shard_bounds(n_reads: int, world: int, rank: int):
    """Contiguous block of the read stream for `rank` (pairs stay together: a pair is one unit)."""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(recs: np.ndarray, device: str = "cpu", read_index: np.ndarray | None = None):
    """all_gather of structured result records (one fixed-size record per read).  Ranks may hold
    different counts; records come back ordered by rank, or by `read_index` when given, so the
    output order is deterministic (unlike the reference)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    raw = np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), recs.dtype.itemsize)
    n = torch.tensor([len(recs)], device=device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(counts + [1])
    buf = torch.zeros((mx, recs.dtype.itemsize), dtype=torch.uint8, device=device)
    if len(recs):
        buf[:len(recs)] = torch.from_numpy(raw.copy()).to(device)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    parts = [o[:c].cpu().numpy().reshape(-1).view(recs.dtype) for o, c in zip(outs, counts)]
    allr = np.concatenate(parts) if parts else recs[:0]
    if read_index is not None:
        idx = torch.full((mx,), -1, dtype=torch.int64, device=device)
        idx[:len(read_index)] = torch.as_tensor(np.asarray(read_index, np.int64)).to(device)
        io = [torch.zeros_like(idx) for _ in range(world)]
        dist.all_gather(io, idx)
        order = np.concatenate([o[:c].cpu().numpy() for o, c in zip(io, counts)])
        allr = allr[np.argsort(order, kind="stable")]
    return allr
