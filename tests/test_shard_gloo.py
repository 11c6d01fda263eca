"""N > 1 path on CPU: read sharding + the one collective (result gather) with world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hmmufotu_amd.engine import PLACE_DTYPE
from hmmufotu_amd.shard import gather_records, shard_bounds


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_reads, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(n_reads, world, rank)
    recs = np.zeros(hi - lo, PLACE_DTYPE)
    recs["c_node"] = np.arange(lo, hi) * 3 + 1          # stands in for the per-read results of this rank's shard
    recs["ratio"] = np.arange(lo, hi) / 7.0
    recs["n_cand"] = rank
    out = gather_records(recs, "cpu", read_index=np.arange(lo, hi))
    dist.barrier()
    if rank == 0:
        q.put(out)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_reads", [11, 2, 1])
def test_gather_world2(n_reads):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_reads, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = q.get()
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert len(out) == n_reads
    assert (out["c_node"] == np.arange(n_reads) * 3 + 1).all()          # every read exactly once, in read order
    assert np.array_equal(out["ratio"], np.arange(n_reads) / 7.0)
    lo0, hi0 = shard_bounds(n_reads, 2, 0)
    assert (out["n_cand"][:hi0] == 0).all() and (out["n_cand"][hi0:] == 1).all()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
