"""bench.py on the GPU box: the N > 1 control flow with the real engine (two ranks sharing the one GPU of the test box,
gloo for the collective payloads), and the single-rank line's contract fields on a small database."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=900):
    e = dict(os.environ); e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return lines[0]


SMALL = ["--leaves", "600", "--cs-len", "1400", "--batch", "96", "--inflight", "2", "--steps", "4", "--warmup", "1"]


def test_two_ranks_launched_by_bench_itself():
    j = _run(["--gpus", "2", "--cpu-sample", "0"] + SMALL, env={"HU_BENCH_SHARE_GPU": "1", "HU_BENCH_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["gathered_records"] == 2 * 96 and j["value"] > 0 and j["scaling"] == "weak"
    assert "x2" in j["config"]["parallelism"]


def test_single_rank_line_has_the_contract_fields_and_zero_unexplained_differences():
    j = _run(["--cpu-sample", "96", "--e2e-reads", "500"] + SMALL)
    assert j["n_gpus"] == 1 and j["unit"] == "reads/s" and j["dtype"] == "f64" and j["vs_baseline"] is None
    r = j["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = j["cpu_baseline"]
    assert c["value"] is not None, c
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    assert c["unexplained_best_branch_diffs"] == 0 and c["candidate_order"]["swaps_unexplained"] == 0 and c["candidate_order"]["candidate_set_differs"] == 0
    # numbers at the tolerance of the north star, iteration counts equal, and the tie-mode report is there
    mr = c["max_rel"]
    assert mr["candidates_compared"] > 96 and mr["outer_iteration_mismatches"] == 0 and mr["nan_placement_mismatches"] == 0
    for k in ("est_loglik", "ratio", "wnr", "height"):
        assert mr[k]["max_rel"] <= 1e-6, (k, mr[k])
    t = c["tie_mode"]
    assert t["reads"] == 96 and t["seed_set_diffs_all_exact_cutoff_ties"] and t["final_pick_differs"] == t["final_pick_diffs_traced_to_a_cutoff_tie"]
    assert r["bound"] == "valu_fp64_issue" and "hbm_roof" in r and "valu_roof" in r and len(r["kernel_source_hash"]) == 16
    sr = j["seed_order_reference"]
    assert sr["value"] > 0 and sr["reads_compared"] == 96 and sr["unexplained_best_branch_diffs"] == 0 and sr["candidate_set_differs"] == 0 and sr["swaps_unexplained"] == 0, sr
    e = j["end_to_end"]
    assert "failed" not in e, e
    assert e["reads"] == 500 and e["value"] > 0 and e["placed"] > 400 and e["tsv_mb"] > 0.5
