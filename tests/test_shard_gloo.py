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


def _bench(*args, env=None, timeout=600):
    import subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(args), env=e, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_bench_launches_its_own_ranks_over_gloo():
    """`python bench.py --gpus 2` without a launcher must run TWO ranks (it starts torch.distributed.run itself), go through
    the real rendezvous / barrier / gather_records / max-over-ranks control flow and print ONE line with n_gpus = 2.
    --rehearse replaces the engine by a sleep (no GPU here); the engine-backed 2-rank run is tests/test_bench_gpu.py."""
    p, lines = _bench("--gpus", "2", "--rehearse", "--steps", "3", "--batch", "5")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    j = lines[0]
    assert j["n_gpus"] == 2 and j["self_launched"] and j["backend"] == "gloo" and j["gathered_records"] == 10 and j["gather_ok"]
    assert j["value"] is None and "rehearsal" in j["data"]           # nothing is measured in a rehearsal


def test_bench_refuses_a_rank_count_it_was_not_asked_for():
    p, lines = _bench("--gpus", "2", "--rehearse", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode == 2 and not lines and "refusing" in p.stderr


def test_bench_failure_of_a_rank_is_a_failure_of_the_run():
    p, lines = _bench("--gpus", "2", "--rehearse", "--batch", "-1")
    assert p.returncode != 0 and not lines
