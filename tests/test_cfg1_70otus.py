"""BASELINE.json config 1: the reference's own fixture (test/70_otus.fasta + .tree, copied as data under
tests/golden/ref_data) built into a database (JC69 of data/gg_97_otus_JC69.sm) and 1,000 simulated SE 150 bp reads.

CPU part (here): the database builder's numbers against SURVEY.md §8 (249 nodes, csLen 1,486, K 1,291), the Newick
numbering, two independent message evaluations against each other, and the oracle's whole per-read task against the
committed golden file (tests/golden/make_cfg1_golden.py).  GPU part: the engine against the oracle on all 1,000 reads."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_cfg1_golden as G  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_70otus_oracle.npz")
_CACHE = {}


def cfg1():
    if "x" not in _CACHE:
        _CACHE["x"] = G.cfg1_inputs()
    return _CACHE["x"]


def test_database_shape_matches_the_survey():
    db, reads, vps = cfg1()
    assert (db.n_nodes, db.cs_len, db.hmm.K, int(db.is_leaf.sum())) == (249, 1486, 1291, 125)
    assert db.parent[0] == -1 and (db.parent[1:] < np.arange(1, db.n_nodes)).all()      # DFS numbering, root 0
    assert (db.blen[db.is_leaf] > 0).all()                                             # fixBranchLength on leaf branches
    assert len(reads) == 1000 and all(len(r.seq) == 150 for r in reads)


def test_newick_numbering_is_the_references_dfs():
    from hmmufotu_amd import synth
    parent, blen, names = synth.parse_newick("((a:1,b:2)x:3,(c:4,'d e':5)0.9:6,f:7);")
    # stack DFS, children pushed in file order: the last child is numbered first
    assert names == ["", "f", "0.9", "d e", "c", "x", "b", "a"]
    assert list(parent) == [-1, 0, 0, 2, 2, 0, 5, 5] and list(blen) == [0, 7, 6, 5, 4, 3, 2, 1]


def test_messages_two_ways_and_root_invariance():
    """synth.evaluate_tree (numpy, level by level) and the oracle's treeEvaluate (C++, restating loglik/evaluate) agree on the
    real tree, and the tree likelihood does not depend on the edge it is read from (src/PhyloTreeUnrooted.h:23-24)."""
    from oracle import oracle_py as O
    from hmmufotu_amd import synth
    db, _, _ = cfg1()
    m = O.Model(db.model.type_id, db.model.pi, db.model.par)
    leaf_only = np.where(db.is_leaf[:, None], db.seq, 0).astype(np.int8)
    up, down, seq, h = O.tree_evaluate(db.parent, db.blen, leaf_only, m, None)
    fin = np.isfinite(db.up)
    assert (np.isfinite(up) == fin).all() and np.abs(up[fin] - db.up[fin]).max() < 1e-9
    assert np.abs(down[1:] - db.down[1:]).max() < 1e-9 and (seq == db.seq).all() and np.abs(h - db.height).max() < 1e-12
    pi = db.model.pi
    root_ll = np.log((np.exp(db.up[0]) * pi).sum(-1)).sum()
    for u in (1, 17, 100, 248):
        P = synth.model_P(db.model, np.array([db.blen[u]]))[0]
        ll = np.log(((np.exp(db.up[u]) @ P.T) * np.exp(db.down[u]) * pi).sum(-1)).sum()
        assert abs(ll - root_ll) < 1e-8 * abs(root_ll), (u, ll, root_ll)


def test_oracle_pipeline_against_golden():
    db, reads, vps = cfg1()
    res, H, T = G.run_oracle(db, reads, vps, threads=4)
    g = np.load(GOLD)
    assert (g["read_len"] == [len(r.seq) for r in reads]).all() and (g["read_node"] == [r.node for r in reads]).all()   # same simulated inputs
    assert (res["aln_ints"] == g["aln_ints"]).all() and np.array_equal(res["cost"], g["cost"])
    assert (res["n_cand"] == g["n_cand"]).all() and (res["cand_node"] == g["cand_node"]).all() and (res["best_nodes"] == g["best_nodes"]).all()
    assert np.allclose(res["best_vals"], g["best_vals"], rtol=1e-12, atol=0, equal_nan=True)
    # the same under (dist, node id) — the arrays the file held alone until round 4, unchanged
    stab, _, _ = G.run_oracle(db, reads, vps, threads=4, tie=0)
    assert (stab["n_cand"] == g["stable_n_cand"]).all() and (stab["cand_node"] == g["stable_cand_node"]).all() and (stab["best_nodes"] == g["stable_best_nodes"]).all()
    assert np.allclose(stab["best_vals"], g["stable_best_vals"], rtol=1e-12, atol=0, equal_nan=True)
    assert int((stab["best_nodes"][:, 0] != res["best_nodes"][:, 0]).sum()) == 19
    # plumbing sanity: every read is aligned and placed, and mostly next to where it was drawn from
    assert (res["aln_ints"][:, 7] == 1).all() and (res["n_cand"] > 0).all()
    near = 0
    for r, bn in zip(reads, res["best_nodes"]):
        c, p = int(bn[0]), int(bn[1])
        near += r.node in (c, p) or int(db.parent[r.node]) in (c, p)
    assert near > 0.5 * len(reads), near


@pytest.mark.gpu
def test_engine_against_oracle_on_config_1():
    """all 1,000 reads through the HIP path (database handed over as arrays, messages by hu_tree_evaluate on the device) vs the
    oracle: alignment bit-exact, candidate order and picks identical or an explained near-tie, values within tolerance."""
    import torch
    from hmmufotu_amd import engine as E
    from oracle import parity
    if E.device_count() < 1:
        pytest.fail("no gfx950 device")
    db, reads, vps = cfg1()
    md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, None)
    n, L = db.seq.shape
    up = torch.zeros((n, L, 4), dtype=torch.float64, device="cuda:0"); down = torch.zeros_like(up)
    leaf_only = np.where(db.is_leaf[:, None], db.seq, 0).astype(np.int8)
    seq, h = E.tree_evaluate(db.parent, db.blen, leaf_only, md, up.data_ptr(), down.data_ptr())
    torch.cuda.synchronize()
    assert (seq == db.seq).all() and np.abs(h - db.height).max() < 1e-12
    fin = np.isfinite(db.up)
    assert np.abs(up.cpu().numpy()[fin] - db.up[fin]).max() < 1e-9
    D = E.Database.from_arrays(db.hmm, db.parent, db.blen, seq, up.data_ptr(), down.data_ptr(), h, md, db.anno_id, db.anno_dist, msgs_on_device=True)
    B = E.Batch(D, len(reads))
    B.set_reads([r.seq for r in reads], vps)
    B.assign(E.default_opts())
    recs = B.alignments(want_align=False)["recs"]; best = B.placements(); cand = B.candidates()
    res, H, T = G.run_oracle(db, reads, vps)
    assert (recs["status"] == res["aln_ints"][:, 7]).all()
    for k, col in (("seq_start", 0), ("seq_end", 1), ("hmm_start", 2), ("hmm_end", 3), ("cs_start", 4), ("cs_end", 5)):
        assert (recs[k] == res["aln_ints"][:, col]).all(), k
    assert np.array_equal(recs["cost"], res["cost"])                                     # bit-exact
    assert (best["n_cand"] == res["n_cand"]).all()
    per = []
    for i in range(len(reads)):
        k = int(res["n_cand"][i]); a, b = int(cand["offs"][i]), int(cand["offs"][i + 1])
        per.append(parity.classify_read(res["cand_node"][i, :k], res["cand_est"][i, :k], res["cand_ratio0"][i, :k], cand["c_node"][a:b],
                                        db.parent, pos=int(res["best_pos"][i])))
    tot = parity.summarize(per)
    assert tot["set_differs"] == 0 and tot["swaps_unexplained"] == 0 and tot["best_unexplained"] == 0, tot
    same = best["c_node"] == res["best_nodes"][:, 0]
    assert same.sum() == len(reads) - tot["best_differs"]
    bv = res["best_vals"]
    assert (best["a_node"][same] == res["best_nodes"][same, 2]).all()
    for name, col, tol in (("ratio", 0, 1e-6), ("wnr", 1, 1e-6), ("loglik", 2, 1e-12), ("height", 3, 1e-6), ("q_place", 4, 1e-9), ("q_taxon", 5, 1e-9),
                           ("anno_dist", 6, 1e-6), ("est_loglik", 7, 1e-6)):
        g, o = best[name][same], bv[same, col]
        # zero-length inner branches of the real tree give ratio = wur / 0 = NaN in placeSeq (src/PhyloTreeUnrooted.cpp:944):
        # reproduced, so a NaN must sit in the same places on both sides
        assert (np.isnan(g) == np.isnan(o)).all(), name
        fin = ~np.isnan(o)
        d = np.abs(g[fin] - o[fin]) / np.maximum(np.abs(o[fin]), 1e-3)
        assert d.max() <= tol, (name, d.max())
        if name == "ratio":
            print("config 1: placements on zero-length branches (NaN ratio on both sides):", int((~fin).sum()))
    print("config 1:", tot)
    B.close(); D.close()



def test_tie_mode_report_on_config_1():
    """the reference keeps the first 50 of a std::sort on dist alone (src/HmmUFOtu_main.cpp:139,
    src/hmmufotu.cpp:646-647); the product keeps (dist, node id).  Both orders are taken from one scan of the tree for all 1,000
    reads of config 1 and the disagreement is counted — on the seed lists, after filterPlacements and on the final
    cNode / pNode / aNode — and every seed that only one list holds must sit exactly at the cut-off distance (a tie).
    The second phase runs on a tree that holds ONLY the seed nodes' message rows (the form bench.py uses at full scale)."""
    from oracle import oracle_py as O
    db, reads, vps = cfg1()
    m = O.Model(db.model.type_id, db.model.pi, db.model.par)
    H = O.Hmm(db.hmm.K, db.hmm.L, db.hmm.EM, db.hmm.EI, db.hmm.T, db.hmm.p2cs, 0)
    T = O.Tree(db.parent, db.blen, db.seq, db.up, db.down, db.height, m, None, db.anno_id)
    rd = [r.seq for r in reads]
    stable = O.default_opts(tieMode=0)
    whole = O.pipeline_batch(H, T, rd, vps, opts=stable, threads=4, want_cands=True)
    p1 = O.pipeline_batch(H, T, rd, vps, opts=stable, threads=4, mode=1, want_lib=True)
    # phase 2 on compact rows: only the nodes some list names
    nodes = np.unique(np.concatenate([p1["seed_ids"].ravel(), p1["lib_ids"].ravel()])); nodes = nodes[nodes >= 0]
    row_of = np.full(db.n_nodes, -1, np.int32); row_of[nodes] = np.arange(len(nodes))
    T2 = O.Tree(db.parent, db.blen, db.seq, db.up[nodes], db.down[nodes], db.height, m, None, db.anno_id, win_start=0, win_len=db.cs_len)
    T2.set_rows(row_of)
    # the two-phase form of the task is the task: same candidates, same picks, same numbers
    two = O.pipeline_batch(H, T2, rd, vps, threads=4, want_cands=True, mode=2, seeds=(p1["seed_cnt"], p1["seed_ids"]))
    assert (two["cand_node"] == whole["cand_node"]).all() and (two["best_nodes"] == whole["best_nodes"]).all()
    assert np.array_equal(two["best_vals"], whole["best_vals"], equal_nan=True) and (two["cand_iters"] == whole["cand_iters"]).all()
    per, summ = O.tie_report(H, T, rd, vps, threads=4, phase1=p1, tree2=T2)
    print("config 1 tie-mode report:", summ)
    assert summ["reads"] == 1000 and summ["seed_set_diffs_all_exact_cutoff_ties"] and summ["reads_with_nan_dist"] == 0
    assert summ["final_pick_differs"] == summ["final_pick_diffs_traced_to_a_cutoff_tie"]
    # the stable side is the task's own result; the libstdc++ side is what the oracle's literal std::sort mode returns
    d = per["order_differs"]
    assert (per["picks"][d, 0] == whole["best_nodes"][d, :3]).all()
    cd = None
    for i in np.nonzero(per["pick_mask"] != 0)[0][:10]:
        a = H.align(rd[i], vps[i])
        r = T.assign(O.digitize(a["align"]), a["csStart"] - 1, a["csEnd"] - 1, O.default_opts(tieMode=1))
        assert (r["nodes"][0][:3] == per["picks"][i, 1]).all()
    # measured on this fixture (248 candidate nodes, 50 kept: a fifth of the tree, so ties at the cut-off are the rule):
    # 548 reads with differing seed sets, 99 with differing candidates after the filter, 19 with a different final branch
    assert summ["seed_set_differs"] == 548 and summ["filtered_candidate_set_differs"] == 99 and summ["final_pick_differs"] == 19, summ
