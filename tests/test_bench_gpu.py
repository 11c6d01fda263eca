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
    # the timed region runs the reference's own seed order (libstdc++'s std::sort on dist alone) and is compared with the oracle's literal std::sort;
    # (dist, node id) is the side block, compared with the oracle in that order
    assert j["config"]["seed_order"] == "reference" and c["oracle_seed_order"].startswith("TIE_LIBSTDCXX") and "seed_order_reference" not in j
    sr = j["seed_order_stable"]
    assert sr["value"] > 0 and sr["reads_compared"] == 96 and sr["unexplained_best_branch_diffs"] == 0 and sr["candidate_set_differs"] == 0 and sr["swaps_unexplained"] == 0, sr
    assert sr["oracle_seed_order"].startswith("TIE_STABLE")
    # RCCL has run on this GPU: a process group of one rank over the nccl backend, the final gather of the records through it (device tensors)
    assert j["rccl_ranks"] == 1 and j["rccl_error"] is None and j["backend"] == "nccl" and j["gathered_records"] == 96, (j["rccl_ranks"], j["rccl_error"])
    e = j["end_to_end"]
    assert "failed" not in e, e
    assert e["reads"] == 500 and e["value"] > 0 and e["placed"] > 400 and e["tsv_mb"] > 0.5


def test_paired_line_and_the_stable_order_as_the_timed_region():
    """--paired (BASELINE configs 4 / 5: two mates per unit, merged regions, 32-bit pairs) on a small database, and --seed-order stable as the timed
    region with the reference's order as the side block: both against the oracle in the same order, zero unexplained differences"""
    j = _run(["--paired", "--read-len", "250", "--cpu-sample", "48", "--e2e-reads", "0", "--leaves", "600", "--cs-len", "2800", "--batch", "48", "--inflight", "2",
              "--steps", "2", "--warmup", "1"])
    assert j["unit"] == "pairs/s" and j["value"] > 0 and j["config"]["seed_order"] == "reference"
    c = j["cpu_baseline"]
    assert c["value"] is not None and c["unexplained_best_branch_diffs"] == 0 and c["candidate_order"]["swaps_unexplained"] == 0 and c["candidate_order"]["candidate_set_differs"] == 0, c
    assert c["max_rel"]["outer_iteration_mismatches"] == 0 and c["max_rel"]["est_loglik"]["max_rel"] <= 1e-6
    sr = j["seed_order_stable"]
    assert sr["reads_compared"] == 48 and sr["unexplained_best_branch_diffs"] == 0 and sr["candidate_set_differs"] == 0, sr
    j2 = _run(["--seed-order", "stable", "--cpu-sample", "96", "--e2e-reads", "0"] + SMALL)
    assert j2["config"]["seed_order"] == "stable" and j2["cpu_baseline"]["oracle_seed_order"].startswith("TIE_STABLE") and j2["cpu_baseline"]["unexplained_best_branch_diffs"] == 0
    sr2 = j2["seed_order_reference"]
    assert sr2["reads_compared"] == 96 and sr2["unexplained_best_branch_diffs"] == 0 and sr2["candidate_set_differs"] == 0 and sr2["oracle_seed_order"].startswith("TIE_LIBSTDCXX"), sr2
