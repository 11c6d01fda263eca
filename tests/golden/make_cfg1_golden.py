#!/usr/bin/env python3
"""Generates tests/golden/cfg1_70otus_oracle.npz: BASELINE.json config 1 (70_otus fixture DB under JC69, 1,000 simulated SE
150 bp reads, seed 0) through the CPU oracle's whole per-read task.  The reference binary cannot be built here (SURVEY.md
§8c), so this pins the ORACLE against regressions — it is not a reference-produced vector.  Inputs: the reference's fixture
files copied as data under tests/golden/ref_data.  Run from the repo root: python tests/golden/make_cfg1_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def cfg1_inputs(n_reads=1000, read_len=150, seed=0):
    from hmmufotu_amd import synth
    db = synth.make_db_70otus("JC69")
    rng = np.random.default_rng(seed)
    reads = synth.simulate_reads(db, n_reads, read_len, rng, mean_cols=500, sd_cols=30)      # hmmufotu-sim: window N(500, 30) CS columns, -r 150
    vps = np.stack([synth.read_vpaths(db.hmm, r) for r in reads])
    return db, reads, vps


def run_oracle(db, reads, vps, threads=0, tie=1):
    """tie = 1: seeds in the reference's own order (literal std::sort on dist alone, the default everywhere); 0: (dist, node id)"""
    from oracle import oracle_py as O
    m = O.Model(db.model.type_id, db.model.pi, db.model.par)
    H = O.Hmm(db.hmm.K, db.hmm.L, db.hmm.EM, db.hmm.EI, db.hmm.T, db.hmm.p2cs, 0)
    T = O.Tree(db.parent, db.blen, db.seq, db.up, db.down, db.height, m, None, db.anno_id)
    return O.pipeline_batch(H, T, [r.seq for r in reads], vps, opts=O.default_opts(tieMode=tie), threads=threads, want_cands=True), H, T


if __name__ == "__main__":
    db, reads, vps = cfg1_inputs()
    res, _, _ = run_oracle(db, reads, vps)
    stab, _, _ = run_oracle(db, reads, vps, tie=0)
    out = os.path.join(ROOT, "tests", "golden", "cfg1_70otus_oracle.npz")
    np.savez_compressed(out, aln_ints=res["aln_ints"], cost=res["cost"], best_nodes=res["best_nodes"], best_vals=res["best_vals"],
                        n_cand=res["n_cand"], cand_node=res["cand_node"],
                        stable_best_nodes=stab["best_nodes"], stable_best_vals=stab["best_vals"], stable_n_cand=stab["n_cand"], stable_cand_node=stab["cand_node"],
                        read_len=np.array([len(r.seq) for r in reads]), read_node=np.array([r.node for r in reads]))
    print("wrote", out, "placed", int((res["n_cand"] > 0).sum()), "of", len(reads))
