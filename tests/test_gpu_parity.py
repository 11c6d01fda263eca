"""GPU parity tests: every stage of the HIP path against the CPU oracle, through the C ABI.

Tolerances: integer / index / string results bit-exact; Viterbi cost bit-exact (same additions
and minima); floating-point SEP quantities within 1e-6 relative as BASELINE.json's north_star
states (the kernels work in linear space with their own scaling and device exp/log, so they are
not expected to be bitwise equal to the log-space CPU arithmetic).
"""
import numpy as np
import pytest

from conftest import get_db, oracle_objects, sim_reads

pytestmark = pytest.mark.gpu

REL = 1e-6


def _engine():
    from hmmufotu_amd import engine as E
    if E.device_count() < 1:
        pytest.fail("no gfx950 device: GPU tests must run on the MI355X box (no CPU fallback exists)")
    return E


def _rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return np.abs(a - b) / np.maximum(1e-300, np.maximum(np.abs(a), np.abs(b)))


NEAR_TIE = 1e-9


def check_order_and_best(res, gpu_filt, gpu_best, stats, parent):
    """Candidate order and final pick vs the oracle: bit-exact, or EXPLAINED difference by difference.  The reference ranks
    candidates by the ESTIMATED loglik and then sorts all-tied final keys, so the winner is the candidate at a
    fixed position of the filterPlacements order.  The only accepted difference is the documented near-tie of
    oracle/parity.py: two candidates that attach at the SAME tree node (estimated ratio exactly 0 / 1 on branches incident
    to it: the same tree in exact arithmetic) whose oracle estimates agree to 1e-9 relative.  Anything else fails."""
    from oracle import parity
    ofilt = [int(x) for x in res["filt_order"]]
    seed_est = {int(s_): e for s_, e in zip(res["seed_ids"], res["est"])}
    pos = ofilt.index(int(res["nodes"][0][0]))          # position picked by the all-ties final std::sort
    c = parity.classify_read(ofilt, [seed_est[n_][2] for n_ in ofilt], [seed_est[n_][0] for n_ in ofilt], gpu_filt, parent, pos)
    assert c["set_differs"] == 0 and c["swaps_unexplained"] == 0 and c["best_unexplained"] == 0, c["detail"]
    assert int(gpu_best["c_node"]) == gpu_filt[pos], (pos, gpu_filt, ofilt)
    stats["near_tie_swaps"] += c["swaps_explained"]
    stats["best_differs_by_tie"] += c["best_differs"]
    stats["reads"] += 1
    return c["best_differs"] == 0


def check_all_candidates(res, cplaces, stats):
    """a15: every candidate's PTPlacement after placeSeq + calcQValues against the oracle's, matched by branch: the taxon node,
    height, annotation distance, both q-values (src/HmmUFOtu_main.cpp:182-216, src/PhyloTreeUnrooted.cpp:936-952)."""
    oc = {int(n_[0]): (n_, v) for n_, v in zip(res["nodes"], res["vals"])}
    assert sorted(oc) == sorted(int(x) for x in cplaces["c_node"])
    for g in cplaces:
        n_, v = oc[int(g["c_node"])]
        assert int(g["p_node"]) == int(n_[1])
        ratio, wnr, loglik, height, qp, qt, ad, est = v
        assert abs(g["ratio"] - ratio) <= REL * max(abs(ratio), 1e-3) and abs(g["wnr"] - wnr) <= REL * max(abs(wnr), 1e-3)
        assert _rel(g["loglik"], loglik) < 1e-12 and _rel(g["est_loglik"], est) < REL
        if abs(ratio - 0.5) > 1e-6:                          # aNode switches at ratio 0.5: not decidable closer than the tolerance
            assert int(g["a_node"]) == int(n_[2]), (g, n_)
            assert abs(g["anno_dist"] - ad) <= REL * max(abs(ad), 1e-3)
        else:
            stats["ratio_at_half"] = stats.get("ratio_at_half", 0) + 1
        assert abs(g["height"] - height) <= REL * max(abs(height), 1e-3)
        assert abs(g["q_place"] - qp) <= 1e-9 * max(1.0, abs(qp)), (g["q_place"], qp)
        if int(g["a_node"]) == int(n_[2]):
            assert abs(g["q_taxon"] - qt) <= 1e-9 * max(1.0, abs(qt)), (g["q_taxon"], qt)
        stats["cands"] = stats.get("cands", 0) + 1


@pytest.mark.parametrize("model", ["GTR", "TN93", "HKY85", "F81", "K80", "JC69"])
def test_model_pr(model):
    E = _engine()
    from oracle import oracle_py as O
    db = get_db(20, 200, model, dg_k=0, seed=3)
    D = E.Database.from_synth(db)
    m = O.Model(db.model.type_id, db.model.pi, db.model.par)
    ts = np.array([0.0, 1e-8, 1e-5, 1e-3, 0.05, 0.3, 1.0, 5.0, 50.0])
    P = D.model_pr(ts)
    for i, t in enumerate(ts):
        assert np.abs(P[i] - m.P(float(t))).max() < 1e-12, (model, t)
    D.close()


def classify_batch(ref, cand, best, parent):
    """oracle pipeline_batch(want_cands=True) vs the engine's candidates / picks: totals of oracle/parity.py's classes"""
    from oracle import parity
    per = []
    for i in range(len(best)):
        k = int(ref["n_cand"][i]); a, b = int(cand["offs"][i]), int(cand["offs"][i + 1])
        c = parity.classify_read(ref["cand_node"][i, :k], ref["cand_est"][i, :k], ref["cand_ratio0"][i, :k], cand["c_node"][a:b], parent,
                                 pos=int(ref["best_pos"][i]) if k else None)
        if k:
            assert int(best[i]["c_node"]) == int(cand["c_node"][a + int(ref["best_pos"][i])])   # same position of the same std::sort
        per.append(c)
    return parity.summarize(per)


def _run_stages(E, db, reads, vps, opts, mode=0, mates=None, mvps=None):
    D = E.Database.from_synth(db)
    B = E.Batch(D, max(len(reads), 1))
    B.set_reads([r if isinstance(r, str) else r.seq for r in reads], vps, mates, mvps)
    B.align(opts)
    return D, B


LONG_DB = dict(n_leaves=60, cs_len=1800, db_kw=dict(n_match=1400))      # leaves of ~1,370 bases: full-length 16S reads


@pytest.mark.parametrize("cfg", [dict(model="GTR", dg_k=4, read_len=150), dict(model="JC69", dg_k=0, read_len=100),
                                 dict(model="TN93", dg_k=4, read_len=250),
                                 # reads longer than the 512 rows the wave kernels hold: 600 bp, and (nearly) the whole gene
                                 dict(model="GTR", dg_k=4, read_len=600, n_reads=16, **LONG_DB),
                                 dict(model="GTR", dg_k=4, read_len=1300, n_reads=8, **LONG_DB)])
def test_align_parity(cfg):
    E = _engine()
    if "db_kw" in cfg:
        db = get_db(cfg["n_leaves"], cfg["cs_len"], cfg["model"], dg_k=cfg["dg_k"], **cfg["db_kw"])
    else:
        db = get_db(120, 700 if cfg["read_len"] < 200 else 1400, cfg["model"], dg_k=cfg["dg_k"])
    _, H, _ = oracle_objects(db)
    reads, vps = sim_reads(db, cfg.get("n_reads", 48), cfg["read_len"])
    # exercise: no seed at all (full DP), one seed only, both seeds
    vps = vps.copy(); vps[0] = 0; vps[1, 1] = 0; vps[2, 0] = vps[2, 1]; vps[2, 1] = 0
    _check_alignments(E, db, H, [r.seq for r in reads], vps)


def _check_alignments(E, db, H, seqs, vps):
    from oracle import oracle_py as O
    opts = E.default_opts()
    D, B = _run_stages(E, db, seqs, vps, opts)
    out = B.alignments(want_align=True, want_trace=True, trace_stride=db.hmm.K + max(len(s) for s in seqs) + 400)
    cd, st, en = B.codes()
    for i, seq in enumerate(seqs):
        a = H.align(seq, vps[i])
        rec = out["recs"][i]
        assert a["ok"] and rec["status"] == 1, i
        assert (rec["seq_start"], rec["seq_end"], rec["hmm_start"], rec["hmm_end"], rec["cs_start"], rec["cs_end"]) == \
               (a["seqStart"], a["seqEnd"], a["hmmStart"], a["hmmEnd"], a["csStart"], a["csEnd"]), i
        assert rec["cost"] == a["cost"], (i, rec["cost"], a["cost"])          # bit-exact
        assert out["trace"][i] == a["trace"], i
        assert out["align"][i] == a["align"], i
        assert bool(rec["used_full"]) == a["usedFull"], i
        ds = O.digitize(a["align"])
        assert (cd[i] == ds).all(), i
        assert st[i] == a["csStart"] - 1 and en[i] == a["csEnd"] - 1
    B.close(); D.close()


def test_reads_longer_than_the_profile():
    """A read with more bases than the profile has match states (K = 1,400): 1,300 bases of a leaf followed by 700 random ones (an
    unclipped adapter / chimeric tail), with its 5' seed only, with none (full DP over 2,000 x 1,400 cells), and a short real read in
    front of a long tail; and a batch that mixes such reads with ordinary 150-base ones (ragged lengths in one launch)."""
    E = _engine()
    db = get_db(LONG_DB["n_leaves"], LONG_DB["cs_len"], "GTR", dg_k=4, **LONG_DB["db_kw"])
    _, H, _ = oracle_objects(db)
    rng = np.random.default_rng(17)
    long_reads, lvps = sim_reads(db, 4, 1300)
    short_reads, svps = sim_reads(db, 6, 150)
    tail = lambda n: "".join(rng.choice(list("ACGT"), size=n))
    seqs, vps = [], []
    for k, r in enumerate(long_reads):
        seqs.append(r.seq + tail(700)); v = lvps[k].copy(); v[1] = 0      # the 3' seed would sit in the random tail: not found
        if k == 3:
            v[:] = 0                                                      # no seed at all
        vps.append(v)
    for k, r in enumerate(short_reads):
        seqs.append(r.seq if k % 2 else r.seq + tail(900)); v = svps[k].copy()
        if k % 2 == 0:
            v[1] = 0
        vps.append(v)
    _check_alignments(E, db, H, seqs, np.stack(vps))


def test_hbm_staged_viterbi_kernel(monkeypatch):
    """reads too long for the LDS-staged wavefront take the HBM-staged kernel: same results"""
    monkeypatch.setenv("HU_VITERBI_HBM", "1")
    test_align_parity(dict(model="GTR", dg_k=4, read_len=150))


def test_value_filing_viterbi_kernels(monkeypatch):
    """the LDS wavefront that files (M, I, D) of every cell (the redo pass of the decision-byte kernel): same results"""
    monkeypatch.setenv("HU_VITERBI_VALUES", "1")
    test_align_parity(dict(model="GTR", dg_k=4, read_len=150))


def test_viterbi_redo_pass(monkeypatch):
    """every traceback flagged 'cannot trust the fill-time decisions': all sequences are redone by the value-filing
    kernels and must come out identical"""
    monkeypatch.setenv("HU_VITERBI_FORCE_REDO", "1")
    test_align_parity(dict(model="GTR", dg_k=4, read_len=150))


def test_generic_decision_byte_viterbi(monkeypatch):
    """the decision-byte kernel for reads longer than one row per thread"""
    monkeypatch.setenv("HU_VITERBI_DEC1", "1")
    test_align_parity(dict(model="GTR", dg_k=4, read_len=150))


def test_seed_band_wider_than_a_wave():
    """a seed of 90 bases gives a band phase of 90 rows: k_viterbi_wave hands such sequences (bands beyond 64 rows) to the
    value-filing kernels; a 40-base seed stays on the wave kernel's one-row-per-lane band routine.  Both bit-exact."""
    E = _engine()
    from hmmufotu_amd import synth
    db = get_db(120, 700, "GTR", dg_k=4)
    _, H, _ = oracle_objects(db)
    reads, _ = sim_reads(db, 16, 150)
    cs2p = synth.cs2profile(db.hmm)
    vps = np.zeros((len(reads), 2, 6), np.int32)
    for i, r in enumerate(reads):
        v = synth.seed_vpath(db.hmm, cs2p, r, 5, 90 if i % 2 == 0 else 40)
        if v[0] > 0 and v[0] <= v[1] and v[2] > 0 and v[2] <= v[3]:
            vps[i, 0] = v
    opts = E.default_opts()
    D, B = _run_stages(E, db, reads, vps, opts)
    out = B.alignments(want_align=True, want_trace=True, trace_stride=db.hmm.K + 400)
    n_long = 0
    for i, r in enumerate(reads):
        a = H.align(r.seq, vps[i])
        rec = out["recs"][i]
        assert a["ok"] and rec["status"] == 1, i
        assert rec["cost"] == a["cost"] and out["trace"][i] == a["trace"] and out["align"][i] == a["align"], i
        n_long += int(vps[i, 0, 3] - vps[i, 0, 2] + 1 > 64)
    assert n_long >= 4
    B.close(); D.close()


def test_align_modes_and_bad_reads():
    E = _engine()
    db = get_db(120, 700, "GTR", dg_k=4)
    reads, vps = sim_reads(db, 16, 120)
    seqs = [r.seq for r in reads]
    seqs[3] = seqs[3][:10] + "x" + seqs[3][11:]       # invalid character -> invalid read, not an abort
    seqs[4] = seqs[4][:20] + "N" + seqs[4][21:]       # degenerate base scores as its first expansion
    for mode in (0, 2):
        _, H, _ = oracle_objects(db, mode)
        opts = E.default_opts(align_mode=mode)
        D, B = _run_stages(E, db, seqs, vps, opts)
        out = B.alignments(want_align=True)
        for i, s in enumerate(seqs):
            a = H.align(s, vps[i])
            rec = out["recs"][i]
            if i == 3:
                assert rec["status"] == 0 and not a["ok"]
                continue
            assert rec["status"] == 1 and a["ok"], (mode, i)
            assert rec["cost"] == a["cost"] and out["align"][i] == a["align"], (mode, i)
        B.close(); D.close()


def test_pe_merge_parity():
    E = _engine()
    from hmmufotu_amd import synth
    from oracle import oracle_py as O
    db = get_db(120, 1400, "GTR", dg_k=4)
    _, H, _ = oracle_objects(db)
    rng = np.random.default_rng(5)
    ins = synth.simulate_reads(db, 24, 100000, rng, amplicon_start=60, amplicon_cols=1200, jitter=20)
    fw, rv, vf, vr = [], [], [], []
    for r in ins:
        n = len(r.seq)
        f = synth.SimRead(r.seq[:120], r.cols[:120], r.node, r.rc, r.cs_start, r.cs_end)
        m = synth.SimRead(r.seq[n - 120:], r.cols[n - 120:], r.node, r.rc, r.cs_start, r.cs_end)  # mate after revcom
        fw.append(f.seq); rv.append(m.seq); vf.append(synth.read_vpaths(db.hmm, f)); vr.append(synth.read_vpaths(db.hmm, m))
    fw[0], rv[0] = rv[0], fw[0]; vf[0], vr[0] = vr[0], vf[0]          # wrong orientation -> chimera status
    opts = E.default_opts()
    D, B = _run_stages(E, db, fw, np.stack(vf), opts, mates=rv, mvps=np.stack(vr))
    out = B.alignments(want_align=True)
    for i in range(len(fw)):
        a = H.align(fw[i], vf[i]); b = H.align(rv[i], vr[i])
        ia = [a[k] for k in ("seqStart", "seqEnd", "hmmStart", "hmmEnd", "csStart", "csEnd")]
        ib = [b[k] for k in ("seqStart", "seqEnd", "hmmStart", "hmmEnd", "csStart", "csEnd")]
        ok, im, cm, am = O.merge(db.cs_len, ia, a["cost"], a["align"].encode("latin1"), ib, b["cost"], b["align"].encode("latin1"))
        rec = out["recs"][i]
        if i == 0:
            assert not ok and rec["status"] == 2
            continue
        assert ok and rec["status"] == 1, i
        assert [rec[k] for k in ("seq_start", "seq_end", "hmm_start", "hmm_end", "cs_start", "cs_end")] == list(im[:6]), i
        assert rec["cost"] == cm and out["align"][i] == am.decode("latin1"), i
    # with -C the badly oriented pair is a chimera without any check: no line in the assignment file, one in --chimera-out
    # carrying the forward read's alignment and a default placement (src/hmmufotu.cpp:629-637, 693-706)
    W = E.Batch(D, len(fw))
    B.get_seed(opts)
    chi = B.check_chimera(W, opts)
    assert chi[0]["checked"] == 0 and chi[0]["is_chimera"] == 0 and np.isnan(chi[0]["lod"]) and chi[1:]["checked"].all()
    B.estimate_seq(opts); B.filter_placements(opts); B.place_seq(opts); B.calc_q_values(opts)
    ids = ["p%d" % i for i in range(len(fw))]
    main = B.format_tsv_chimera(ids, None, db.annos, chi, True, 0).strip("\n").split("\n")
    side = B.format_tsv_chimera(ids, None, db.annos, chi, True, 1).strip("\n").split("\n")
    assert [l.split("\t")[0] for l in main] == ids[1:] and len(side) == 1
    a0 = H.align(fw[0], vf[0])
    f = side[0].split("\t")
    assert f[0] == "p0" and [int(x) for x in f[2:8]] == [a0[k] for k in ("seqStart", "seqEnd", "hmmStart", "hmmEnd", "csStart", "csEnd")]
    assert f[9] == a0["align"] and f[10:15] == ["-1", "-1", "UNASSIGNED", "UNASSIGNED", "nan"]
    assert f[15:] == ["NULL", "nan", "-1", "UNASSIGNED", "nan", "nan", "nan", "nan"]
    plain = B.format_tsv(ids, None, db.annos).strip("\n").split("\n")
    assert [l.split("\t")[:10] + l.split("\t")[15:] for l in main] == [l.split("\t") for l in plain]
    W.close(); B.close(); D.close()


@pytest.mark.parametrize("cfg", [dict(model="GTR", dg_k=4, n_leaves=300, cs_len=1400, read_len=250),
                                 dict(model="GTR", dg_k=0, n_leaves=150, cs_len=700, read_len=150),
                                 dict(model="JC69", dg_k=0, n_leaves=100, cs_len=700, read_len=150),
                                 dict(model="HKY85", dg_k=4, n_leaves=100, cs_len=700, read_len=100),
                                 dict(model="K80", dg_k=2, n_leaves=80, cs_len=500, read_len=100),
                                 # two equal pairs of base frequencies outside K80 / JC69: the A and G components of every all-gap column are equal
                                 # in exact arithmetic, and the inferred states (hence the unweighted wnr, compared bit for bit) hang on the tie rule
                                 dict(model="HKY85", dg_k=4, n_leaves=100, cs_len=700, read_len=100, db_kw=dict(pi=(0.3, 0.2, 0.3, 0.2))),
                                 dict(model="F81", dg_k=0, n_leaves=100, cs_len=700, read_len=150, db_kw=dict(pi=(0.2, 0.3, 0.2, 0.3))),
                                 dict(model="TN93", dg_k=3, n_leaves=100, cs_len=700, read_len=100, db_kw=dict(pi=(0.35, 0.15, 0.35, 0.15))),
                                 # 5,199 nodes = 21 blocks of 256 >= 2 x max_nseed: the distance-only scan and its top-k (the gg_97-scale path)
                                 dict(model="GTR", dg_k=4, n_leaves=2600, cs_len=300, read_len=100, max_nseed=10, n_reads=24, db_kw=dict(n_match=200), seed_order=0),
                                 dict(model="HKY85", dg_k=0, n_leaves=2600, cs_len=700, read_len=300, max_nseed=8, n_reads=12, db_kw=dict(n_match=450), seed_order=0),
                                 # the same two trees in the reference's own seed order (the default): pair scan + k_seed_refsort, 16- and 32-bit pairs
                                 dict(model="GTR", dg_k=4, n_leaves=2600, cs_len=300, read_len=100, max_nseed=10, n_reads=24, db_kw=dict(n_match=200)),
                                 dict(model="HKY85", dg_k=0, n_leaves=2600, cs_len=700, read_len=300, max_nseed=8, n_reads=12, db_kw=dict(n_match=450)),
                                 # (dist, node id) on the small-tree kernels (pair matrix + k_seed_topk)
                                 dict(model="GTR", dg_k=4, n_leaves=300, cs_len=1400, read_len=250, seed_order=0),
                                 dict(model="JC69", dg_k=0, n_leaves=100, cs_len=700, read_len=150, seed_order=0),
                                 # full-length reads: > 256 base sites and > 1,536 region columns per read (the kernels without split slots)
                                 dict(model="GTR", dg_k=4, read_len=600, n_reads=12, **LONG_DB),
                                 dict(model="GTR", dg_k=4, read_len=1300, n_reads=8, **LONG_DB)])
def test_sep_parity(cfg):
    """seed scan, top-k, estimate, filter, place, q-values vs the oracle, stage by stage.  seed_order: 1 (default) = the reference's own
    (literal std::sort on dist alone: oracle TIE_LIBSTDCXX, engine HU_SEED_ORDER_LIBSTDCXX), 0 = (dist, node id) on both sides."""
    E = _engine()
    from oracle import oracle_py as O
    db = get_db(cfg["n_leaves"], cfg["cs_len"], cfg["model"], dg_k=cfg["dg_k"], **cfg.get("db_kw", {}))
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, cfg.get("n_reads", 40), cfg["read_len"])
    order = cfg.get("seed_order", 1)
    opts = E.default_opts(max_nseed=cfg.get("max_nseed", 50), seed_order=order)
    D, B = _run_stages(E, db, reads, vps, opts)
    B.get_seed(opts); B.estimate_seq(opts); B.filter_placements(opts); B.place_seq(opts); B.calc_q_values(opts)
    cd, st, en = B.codes()
    cnt, ids, sd, sN = B.seeds()
    er, ew, el = B.estimates()
    cand = B.candidates()
    best = B.placements()
    coffs, cpl = B.candidate_places()
    oo = O.default_opts(maxNSeed=cfg.get("max_nseed", 50), tieMode=order)
    oo_tie = O.default_opts(maxNSeed=cfg.get("max_nseed", 50), tieTol=1e-9, tieMode=order)
    exact_ties = "pi" in cfg.get("db_kw", {})                 # equal base frequencies outside K80 / JC69
    worst = dict(est=0.0, ratio=0.0, wnr=0.0)
    stats = dict(reads=0, near_tie_swaps=0, best_differs_by_tie=0, reads_decided_by_exact_ties=0)
    for i in range(len(reads)):
        d, N = B.pdist(i)
        od, oN = T.pdist_all(cd[i], int(st[i]), int(en[i]))
        assert (d == od).all() and (N == oN).all(), i                      # bit-exact counts for every node
        res = T.assign(cd[i], int(st[i]), int(en[i]), oo)
        k = len(res["seed_ids"])
        if exact_ties and not np.array_equal(ew[i, :k], res["est"][:, 1]):
            # Inferred states of components that are EQUAL in exact arithmetic: the reference's winner is the last bit of Eigen's
            # summation order (row A adds its four products in another order than row G), the product takes the first of the tied
            # components.  Such a read is compared with the oracle run under the product's rule — everything
            # below must then hold bit for bit / to tolerance as for any other read — and counted.
            res = T.assign(cd[i], int(st[i]), int(en[i]), oo_tie)
            stats["reads_decided_by_exact_ties"] += 1
        assert cnt[i] == k and (ids[i, :k] == res["seed_ids"]).all(), i     # bit-exact seed ids and order
        assert (sd[i, :k] == res["seed_d"]).all() and (sN[i, :k] == res["seed_N"]).all()
        assert np.array_equal(er[i, :k], res["est"][:, 0]), i               # ratio: same integer quotients
        assert np.array_equal(ew[i, :k], res["est"][:, 1]), i               # unweighted wnr = count / n
        worst["est"] = max(worst["est"], _rel(el[i, :k], res["est"][:, 2]).max())
        lo, hi = cand["offs"][i], cand["offs"][i + 1]
        assert hi - lo == res["n"], i
        # candidates are reported in filterPlacements order; the oracle's list is in final output order
        oc = {int(n_[0]): (v[0], v[1], int(n_[3])) for n_, v in zip(res["nodes"], res["vals"])}
        for c in range(lo, hi):
            r0, w0_, it = oc[int(cand["c_node"][c])]
            worst["ratio"] = max(worst["ratio"], abs(cand["ratio"][c] - r0) / max(abs(r0), 1e-3))
            worst["wnr"] = max(worst["wnr"], abs(cand["wnr"][c] - w0_) / max(abs(w0_), 1e-3))
        b = best[i]
        same = check_order_and_best(res, [int(x) for x in cand["c_node"][lo:hi]], b, stats, db.parent)   # bit-exact ids, or explained one by one
        check_all_candidates(res, cpl[coffs[i]:coffs[i + 1]], stats)
        assert b["n_cand"] == res["n"]
        assert _rel(b["loglik"], res["vals"][0][2]) < 1e-12
        assert _rel(b["q_place"], res["vals"][0][4]) < 1e-9
        if same:                                             # the record of the winner as written to the TSV (a15)
            n0, v0 = res["nodes"][0], res["vals"][0]
            assert (int(b["p_node"]), int(b["a_node"])) == (int(n0[1]), int(n0[2]))
            assert _rel(b["q_taxon"], v0[5]) < 1e-9 and abs(b["anno_dist"] - v0[6]) <= REL * max(abs(v0[6]), 1e-3)
            assert abs(b["height"] - v0[3]) <= REL * max(abs(v0[3]), 1e-3)
    assert worst["est"] < REL and worst["ratio"] < REL and worst["wnr"] < REL, worst
    print("parity stats", cfg, stats, worst)
    B.close(); D.close()


def test_seed_getters_respect_the_callers_stride():
    """hu_batch_get_seeds / _estimates write rows of HU_MAX_SEEDS (64) entries; the _strided forms write min(stride, 64) per read
    and nothing beyond: a caller that sizes its buffers [n][max_nseed] must not be overrun (guard entries stay untouched)."""
    E = _engine()
    db = get_db(150, 700, "GTR", dg_k=0)
    reads, vps = sim_reads(db, 6, 150)
    opts = E.default_opts(max_nseed=50)
    D, B = _run_stages(E, db, reads, vps, opts)
    B.get_seed(opts); B.estimate_seq(opts)
    cnt, ids, sd, sN = B.seeds(); er, ew, el = B.estimates()
    assert ids.shape[1] == E.HU_MAX_SEEDS == 64
    for stride in (50, 7, 64, 100):
        c2, i2, d2, n2 = B.seeds_strided(stride, guard=40)
        r2, w2, l2 = B.estimates_strided(stride, guard=40)
        m = min(stride, 64)
        assert (c2 == cnt).all()
        for full, got in ((ids, i2), (sd, d2), (sN, n2), (er, r2), (ew, w2), (el, l2)):
            rows = got[:len(reads) * stride].reshape(len(reads), stride)
            assert np.array_equal(rows[:, :m], full[:, :m], equal_nan=True)
            assert (rows[:, m:] == -777).all() and (got[len(reads) * stride:] == -777).all()      # nothing written past min(stride, 64)
    B.close(); D.close()


def test_scan_tiling_is_invisible():
    """the seed scan tiles sixteen reads that start at neighbouring columns (reads sorted by region start on the device); with reads
    whose regions are scattered over the whole consensus the tiles differ completely from read order — the results must not"""
    E = _engine()
    db = get_db(300, 2000, "GTR", dg_k=4, seed=7)
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 40, 120, amplicon=False)             # uniform window starts (src/hmmufotu-sim.cpp:371)
    opts = E.default_opts()
    res = []
    for unsorted in (0, 1):
        D = E.Database.from_synth(db); B = E.Batch(D, 64)
        B.set_knob("tile_unsorted", unsorted)
        B.set_reads([r.seq for r in reads], vps); B.assign(opts)
        cd, st, en = B.codes()
        res.append((B.seeds(), B.placements().copy(), [B.pdist(i) for i in (0, 17, 39)]))
        if not unsorted:
            assert st.max() - st.min() > 800                        # really scattered
            for k, i in enumerate((0, 17, 39)):
                od, oN = T.pdist_all(cd[i], int(st[i]), int(en[i]))
                assert (res[0][2][k][0] == od).all() and (res[0][2][k][1] == oN).all()
        B.close(); D.close()
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    for k in ("c_node", "a_node", "ratio", "wnr", "q_place"):
        assert np.array_equal(res[0][1][k], res[1][1][k], equal_nan=True), k


@pytest.mark.parametrize("read_len", [150, 300])
def test_pair_matrix_widths(read_len):
    """(d, N) pairs are kept in 16 bits when no read of the batch has more than 255 bases in its region, else in 32 (reads of 300 bp
    here); the knob forces 32.  Same (d, N) for every node (vs the oracle), same seeds, same estimates either way."""
    E = _engine()
    db = get_db(300, 2000, "GTR", dg_k=4, seed=7)
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 6, read_len)
    opts = E.default_opts()
    res = []
    for force32 in (0, 1):
        D, B = _run_stages(E, db, reads, vps, opts)
        B.set_knob("pairs32", force32)
        B.get_seed(opts); B.estimate_seq(opts)
        cd, st, en = B.codes()
        pd = [B.pdist(i) for i in range(len(reads))]
        res.append((B.seeds(), B.estimates(), pd))
        if not force32:
            nb = max(int((cd[i, st[i]:en[i] + 1] >= 0).sum()) for i in range(len(reads)))
            assert (nb <= 255) == (read_len <= 255)
            for i in range(len(reads)):
                od, oN = T.pdist_all(cd[i], int(st[i]), int(en[i]))
                assert (pd[i][0] == od).all() and (pd[i][1] == oN).all()
        B.close(); D.close()
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b, equal_nan=True)
    for (d0, n0), (d1, n1) in zip(res[0][2], res[1][2]):
        assert np.array_equal(d0, d1) and np.array_equal(n0, n1)


def test_sep_weighted_and_maxheight():
    E = _engine()
    from oracle import oracle_py as O
    db = get_db(150, 700, "GTR", dg_k=0)
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 16, 150)
    mh = float(np.median(db.height))
    opts = E.default_opts(weighted=1, max_height=mh, max_nseed=20, prior=1)
    D, B = _run_stages(E, db, reads, vps, opts)
    B.assign(opts) if False else (B.get_seed(opts), B.estimate_seq(opts), B.filter_placements(opts), B.place_seq(opts), B.calc_q_values(opts))
    cd, st, en = B.codes()
    cnt, ids, _, _ = B.seeds()
    er, ew, el = B.estimates()
    best = B.placements()
    coffs, cpl = B.candidate_places()
    oo = O.default_opts(weighted=1, maxHeight=mh, maxNSeed=20, prior=1)
    diff = 0
    for i in range(len(reads)):
        res = T.assign(cd[i], int(st[i]), int(en[i]), oo)
        k = len(res["seed_ids"])
        assert cnt[i] == k and (ids[i, :k] == res["seed_ids"]).all()
        assert _rel(ew[i, :k], res["est"][:, 1]).max() < REL and _rel(el[i, :k], res["est"][:, 2]).max() < REL
        # with --prior height the final keys (q_place) are distinct: every candidate's record is compared by branch
        # (a_node under the height bound, height, anno_dist, q_place, q_taxon), and the pick is the oracle's unless the
        # oracle's own two best keys are a tie to 1e-9
        oc = {int(n_[0]): (n_, v) for n_, v in zip(res["nodes"], res["vals"])}
        g = cpl[coffs[i]:coffs[i + 1]]
        assert sorted(oc) == sorted(int(x) for x in g["c_node"])
        for rec in g:
            n_, v = oc[int(rec["c_node"])]
            assert int(rec["a_node"]) == int(n_[2]) or abs(v[0] - 0.5) < 1e-6
            assert abs(rec["height"] - v[3]) <= REL * max(abs(v[3]), 1e-3) and abs(rec["anno_dist"] - v[6]) <= REL * max(abs(v[6]), 1e-3)
            assert abs(rec["q_place"] - v[4]) <= 1e-5 * max(1.0, abs(v[4])), (rec["q_place"], v[4])    # q = -10 log10(1 - p): p to 1e-6
            assert abs(rec["q_taxon"] - v[5]) <= 1e-5 * max(1.0, abs(v[5]))
        b = best[i]
        if b["c_node"] != int(res["nodes"][0][0]):
            diff += 1
            assert len(res["vals"]) > 1 and abs(res["vals"][0][4] - res["vals"][1][4]) <= 1e-9 * max(1.0, abs(res["vals"][0][4])), (i, res["vals"][:2])
        else:
            assert b["a_node"] == int(res["nodes"][0][2]) and abs(b["q_taxon"] - res["vals"][0][5]) <= 1e-5 * max(1.0, abs(res["vals"][0][5]))
    print("prior=height: picks differing by an exact key tie:", diff)
    B.close(); D.close()


def test_set_aligned_entry_and_empty_batch():
    E = _engine()
    db = get_db(120, 700, "GTR", dg_k=4)
    _, H, T = oracle_objects(db)
    from oracle import oracle_py as O
    reads, vps = sim_reads(db, 8, 120)
    codes, s, e = [], [], []
    for r, vp in zip(reads, vps):
        a = H.align(r.seq, vp)
        codes.append(O.digitize(a["align"])); s.append(a["csStart"] - 1); e.append(a["csEnd"] - 1)
    D = E.Database.from_synth(db)
    B = E.Batch(D, 16)
    opts = E.default_opts()
    B.set_aligned(np.stack(codes), s, e)
    B.assign(opts)
    best = B.placements(); cand = B.candidates()
    stats = dict(reads=0, near_tie_swaps=0, best_differs_by_tie=0)
    for i in range(len(reads)):
        res = T.assign(codes[i], s[i], e[i], O.default_opts())
        lo, hi = cand["offs"][i], cand["offs"][i + 1]
        check_order_and_best(res, [int(x) for x in cand["c_node"][lo:hi]], best[i], stats, db.parent)
        assert best[i]["n_cand"] == res["n"]
    print("set_aligned parity", stats)
    B.set_reads([], np.zeros((0, 2, 6), np.int32))
    B.assign(opts)
    assert len(B.placements()) == 0
    B.close(); D.close()


def test_db_load_from_reference_formats(tmp_path):
    E = _engine()
    from hmmufotu_amd import synth
    db = get_db(60, 400, "GTR", dg_k=4, seed=3)
    hp, pp = str(tmp_path / "t.hmm"), str(tmp_path / "t.ptu")
    synth.write_hmm(db.hmm, hp); synth.write_ptu(db, pp)
    D1 = E.Database.load(hp, pp)
    D2 = E.Database.from_synth(db)
    reads, vps = sim_reads(db, 8, 100)
    outs = []
    for D in (D1, D2):
        B = E.Batch(D, 8)
        B.set_reads([r.seq for r in reads], vps)
        B.assign(E.default_opts())
        outs.append((B.placements().copy(), B.alignments()["align"]))
        B.close()
    assert outs[0][1] == outs[1][1]
    for k in ("c_node", "a_node", "ratio", "wnr", "q_place"):
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k
    D1.close(); D2.close()


def test_build_align_path_and_tsv():
    """host helpers of the ABI: buildAlignPath against the oracle, and the TSV line format"""
    E = _engine()
    from hmmufotu_amd import synth
    db = get_db(120, 700, "GTR", dg_k=4)
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 12, 150)
    D = E.Database.from_synth(db)
    for r in reads[:6]:
        for sf in (0, 5, len(r.seq) - 20):
            c0, c1 = int(r.cols[sf]), int(r.cols[sf + 19])
            s = ["-"] * (c1 - c0 + 1)
            for k in range(20):
                s[int(r.cols[sf + k]) - c0] = r.seq[sf + k]
            cs = "".join(s)
            assert list(D.build_align_path(c0 + 1, c1 + 1, cs, sf + 1, sf + 20)) == list(H.build_align_path(c0 + 1, c1 + 1, cs, sf + 1, sf + 20))
    seqs = [r.seq for r in reads]
    seqs[2] = "?" + seqs[2][1:]                                   # invalid read: no TSV line
    B = E.Batch(D, 16)
    B.set_reads(seqs, vps)
    B.assign(E.default_opts())
    ids = ["r%d" % i for i in range(len(seqs))]
    tsv = B.format_tsv(ids, ["desc %d" % i for i in range(len(seqs))], db.annos)
    assert B.format_tsv_copy(ids, ["desc %d" % i for i in range(len(seqs))], db.annos) == tsv     # the (buf, cap) form of the ABI: same bytes
    lines = tsv.strip("\n").split("\n")
    assert len(lines) == len(seqs) - 1 and all(not l.startswith("r2\t") for l in lines)
    best = B.placements(); alns = B.alignments()
    for l in lines:
        f = l.split("\t")
        assert len(f) == 18
        i = int(f[0][1:])
        rec, b = alns["recs"][i], best[i]
        assert f[1] == "desc %d" % i
        assert [int(x) for x in f[2:8]] == [rec[k] for k in ("seq_start", "seq_end", "hmm_start", "hmm_end", "cs_start", "cs_end")]
        assert f[8] == "%g" % rec["cost"] and f[9] == alns["align"][i] and len(f[9]) == db.cs_len
        assert f[10] == "%d->%d" % (b["c_node"], b["p_node"]) and f[11] == "%g" % b["ratio"]
        assert int(f[12]) == b["a_node"] and f[13] == db.annos[b["a_node"]]
        assert f[14] == "%g" % b["anno_dist"] and f[15] == "%g" % b["loglik"] and f[16] == "%g" % b["q_place"] and f[17] == "%g" % b["q_taxon"]
    B.close(); D.close()


@pytest.mark.parametrize("model,dg_k,pi", [("GTR", 4, None), ("GTR", 0, None), ("K80", 0, None), ("TN93", 3, None),
                                           # A and G (C and T) equally frequent outside K80 / JC69: exact-arithmetic ties of the ancestral argmax
                                           ("HKY85", 0, (0.3, 0.2, 0.3, 0.2)), ("F81", 4, (0.2, 0.3, 0.2, 0.3))])
def test_tree_pre_evaluation(model, dg_k, pi):
    """hu_tree_evaluate (device two-pass pruning) vs the oracle's restatement of
    loglik/evaluate: every directed-edge message, the ancestral argmax sequences and the node heights"""
    E = _engine()
    import torch
    from oracle import oracle_py as O
    db = get_db(60, 300, model, dg_k=dg_k, seed=13, **(dict(pi=pi) if pi else {}))
    n, L = db.seq.shape
    md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r if dg_k else None)
    leaf_only = np.where(db.is_leaf[:, None], db.seq, 0).astype(np.int8)
    for (w0, wl) in ((0, 0), (40, 200)):
        W = wl or L
        up = torch.full((n, W, 4), 7.0, dtype=torch.float64, device="cuda:0"); down = torch.zeros_like(up)
        seq, h = E.tree_evaluate(db.parent, db.blen, leaf_only, md, up.data_ptr(), down.data_ptr(), w0, wl)
        torch.cuda.synchronize()
        m = O.Model(db.model.type_id, db.model.pi, db.model.par)
        oup, odown, oseq, oh = O.tree_evaluate(db.parent, db.blen, leaf_only, m, db.dg_r if dg_k else None)
        gu, gd = up.cpu().numpy(), down.cpu().numpy()
        ou, od = oup[:, w0:w0 + W], odown[:, w0:w0 + W]
        fin = np.isfinite(ou)
        assert (np.isfinite(gu) == fin).all()
        assert np.abs(gu[fin] - ou[fin]).max() < 1e-9 * max(1.0, np.abs(ou[fin]).max())
        assert np.abs(gd[1:] - od[1:]).max() < 1e-9 * max(1.0, np.abs(od[1:]).max())
        inner = ~db.is_leaf
        gs, os_ = seq[inner][:, w0:w0 + W].astype(int), oseq[inner][:, w0:w0 + W].astype(int)
        if pi is None:
            assert (gs == os_).all()
        else:
            # components equal in exact arithmetic: the reference's argmax is decided by the last bit of its summation order, the
            # engine returns the first of the tied components.  A differing cell must be exactly that: the
            # oracle's own message holds the two components within 1e-9, and the engine's is the earlier one.
            uu, jj = np.nonzero(gs != os_)
            msg = ou[inner]
            for u, j in zip(uu, jj):
                assert gs[u, j] < os_[u, j] and abs(msg[u, j, gs[u, j]] - msg[u, j, os_[u, j]]) < 1e-9, (u, j, msg[u, j])
            print("ancestral states decided by exact-arithmetic ties:", len(uu), "of", gs.size)
            assert len(uu) < gs.size // 100
        assert (seq[db.is_leaf] == leaf_only[db.is_leaf]).all()
        assert np.abs(h - oh).max() < 1e-12 and np.abs(h - db.height).max() < 1e-12


def test_cli_end_to_end(tmp_path):
    """hmmufotu-amd <DB> <reads.fasta>: DB files in the reference's formats, host seed index, batched
    engine, TSV — identical to driving the ABI from Python with the same seeds; FASTQ + PE too"""
    E = _engine()
    import os, subprocess
    from hmmufotu_amd import synth
    db = get_db(120, 700, "GTR", dg_k=4)
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hmmufotu_amd", "bin", "hmmufotu-amd")
    assert os.path.exists(cli), "CLI binary missing: run __graft_entry__.build()"
    pre = str(tmp_path / "db")
    synth.write_hmm(db.hmm, pre + ".hmm"); synth.write_ptu(db, pre + ".ptu")
    rng = np.random.default_rng(8)
    leaves = np.nonzero(db.is_leaf)[0]
    reads = []
    for i in range(24):                                       # leaf substrings with a few substitutions
        u = int(rng.choice(leaves)); s = db.seq[u]; c = np.nonzero(s >= 0)[0]
        a = int(rng.integers(0, max(1, len(c) - 108))); c = c[a:a + 104]
        b = s[c].copy(); k = rng.integers(25, 80, size=2); b[k] = (b[k] + 1) % 4
        reads.append("".join("ACGT"[x] for x in b))
    fa = str(tmp_path / "r.fasta"); fq = str(tmp_path / "r.fq")
    with open(fa, "w") as f:
        for i, r in enumerate(reads):
            f.write(">read%d sample=%d\n%s\n%s\n" % (i, i % 3, r[:60], r[60:]))
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            f.write("@read%d sample=%d\n%s\n+\n%s\n" % (i, i % 3, r, "I" * len(r)))
    out = subprocess.run([cli, pre, fa, "-s", "1", "--batch", "16"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.split("\n") if l and not l.startswith("#")]
    assert lines[0].split("\t")[0] == "id" and len(lines) == 1 + len(reads)
    # the same through the Python binding
    D = E.Database.load(pre + ".hmm", pre + ".ptu")
    ix = E.SeedIndex(db.parent, db.seq, db.hmm, 20)
    B = E.Batch(D, 32)
    B.set_reads(reads, ix.lookup(reads, 50, 0)); B.assign(E.default_opts())
    want = B.format_tsv(["read%d" % i for i in range(len(reads))], ["sample=%d" % (i % 3) for i in range(len(reads))], db.annos).strip("\n").split("\n")
    assert lines[1:] == want
    # FASTQ input and strand auto-detection give the same assignment
    out2 = subprocess.run([cli, pre, fq, "-t", "10"], capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stderr
    assert [l for l in out2.stdout.split("\n") if l and not l.startswith("#")][1:] == want
    # five batches through the reader / worker pipeline: same lines, same order
    outb = subprocess.run([cli, pre, fa, "-s", "1", "--batch", "5"], capture_output=True, text=True, timeout=300)
    assert outb.returncode == 0, outb.stderr
    assert [l for l in outb.stdout.split("\n") if l and not l.startswith("#")][1:] == want
    # two database replicas x two batches in flight (here both replicas on the one device of the test box): same lines, same order
    outm = subprocess.run([cli, pre, fa, "-s", "1", "--batch", "5", "--gpus", "2", "--inflight", "2", "-v"], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, HU_CLI_SHARE_GPU="1"))
    assert outm.returncode == 0 and "2 device(s) x 2 batches in flight" in outm.stderr, outm.stderr
    assert [l for l in outm.stdout.split("\n") if l and not l.startswith("#")][1:] == want
    assert subprocess.run([cli, pre, fa, "--gpus", "9"], capture_output=True).returncode != 0          # more devices than the box has
    # with the database's own <DB>.csfm beside it (written here by the reference's libcds + libdivsufsort, oracle/_ref/csfm_ref, from the
    # leaf rows): the CLI reads it instead of rebuilding the index; these reads come from single leaves, so the assignments are the same
    ref_writer = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "csfm_ref")
    if os.path.exists(ref_writer):
        msa = str(tmp_path / "msa.fasta")
        with open(msa, "w") as f:
            for u in leaves:
                f.write(">n%d\n%s\n" % (u, "".join("ACGT"[x] if x >= 0 else "-" for x in db.seq[u])))
        assert subprocess.run([ref_writer, msa, pre + ".csfm"], capture_output=True).returncode == 0
        outc = subprocess.run([cli, pre, fa, "-s", "1", "-v"], capture_output=True, text=True, timeout=300)
        assert outc.returncode == 0 and "seed index read from the .csfm" in outc.stderr, outc.stderr
        assert [l for l in outc.stdout.split("\n") if l and not l.startswith("#")][1:] == want
        outn = subprocess.run([cli, pre, fa, "-s", "1", "-v", "--no-csfm"], capture_output=True, text=True, timeout=300)
        assert outn.returncode == 0 and "seed index built" in outn.stderr
        os.remove(pre + ".csfm")
    # -S <seed>: seed hits drawn from all occurrences (CSFMIndex::locateOne) — reproducible, whatever the batching; same lines as the ABI
    # driven from Python with the same draws
    outs = [subprocess.run([cli, pre, fa, "-s", "1", "-S", "7", "--batch", bs], capture_output=True, text=True, timeout=300) for bs in ("16", "5")]
    assert all(o_.returncode == 0 for o_ in outs), outs[0].stderr
    ls = [[l for l in o_.stdout.split("\n") if l and not l.startswith("#")][1:] for o_ in outs]
    B.set_reads(reads, ix.lookup_random(reads, 7, 0, 50, 0)); B.assign(E.default_opts())
    assert ls[0] == ls[1] == B.format_tsv(["read%d" % i for i in range(len(reads))], ["sample=%d" % (i % 3) for i in range(len(reads))], db.annos).strip("\n").split("\n")
    # --seed-order: `reference` (the default: libstdc++'s std::sort on dist alone, hu_opts.seed_order = HU_SEED_ORDER_LIBSTDCXX) spelled out gives the
    # lines of the plain run; `stable` = (dist, node id) gives the ABI's lines under HU_SEED_ORDER_STABLE
    outr = subprocess.run([cli, pre, fa, "-s", "1", "--seed-order", "reference"], capture_output=True, text=True, timeout=300)
    assert outr.returncode == 0, outr.stderr
    assert [l for l in outr.stdout.split("\n") if l and not l.startswith("#")][1:] == want
    outst = subprocess.run([cli, pre, fa, "-s", "1", "--seed-order", "stable"], capture_output=True, text=True, timeout=300)
    assert outst.returncode == 0, outst.stderr
    B.set_reads(reads, ix.lookup(reads, 50, 0)); B.assign(E.default_opts(seed_order=0))
    assert [l for l in outst.stdout.split("\n") if l and not l.startswith("#")][1:] == \
        B.format_tsv(["read%d" % i for i in range(len(reads))], ["sample=%d" % (i % 3) for i in range(len(reads))], db.annos).strip("\n").split("\n")
    assert subprocess.run([cli, pre, fa, "--seed-order", "x"], capture_output=True).returncode != 0
    # bzip2-compressed input and output (.bz2 by name; libbz2 bound at run time)
    import bz2
    fqb = str(tmp_path / "r.fastq.bz2"); outb2 = str(tmp_path / "out.tsv.bz2")
    with open(fq, "rb") as fi, bz2.open(fqb, "wb") as fo:
        fo.write(fi.read())
    outbz = subprocess.run([cli, pre, fqb, "-s", "1", "-o", outb2], capture_output=True, text=True, timeout=300)
    assert outbz.returncode == 0, outbz.stderr
    assert [l for l in bz2.open(outb2, "rt").read().split("\n") if l and not l.startswith("#")][1:] == want
    # gzip-compressed input and output (the reference reads / writes .gz through boost::iostreams)
    import gzip
    fqz = str(tmp_path / "r.fastq.gz"); outz = str(tmp_path / "out.tsv.gz")
    with open(fq, "rb") as fi, gzip.open(fqz, "wb") as fo:
        fo.write(fi.read())
    outg = subprocess.run([cli, pre, fqz, "-s", "1", "-o", outz], capture_output=True, text=True, timeout=300)
    assert outg.returncode == 0 and outg.stdout == "", outg.stderr
    with gzip.open(outz, "rt") as f:
        assert [l for l in f.read().split("\n") if l and not l.startswith("#")][1:] == want
    # reverse-complemented input is recognised (strand 2) and assigned identically
    rc = str(tmp_path / "rc.fasta")
    with open(rc, "w") as f:
        for i, r in enumerate(reads):
            f.write(">read%d sample=%d\n%s\n" % (i, i % 3, synth.revcom(r)))
    out3 = subprocess.run([cli, pre, rc, "-v"], capture_output=True, text=True, timeout=300)
    assert out3.returncode == 0 and "strand determined as 2" in out3.stderr, out3.stderr
    assert [l for l in out3.stdout.split("\n") if l and not l.startswith("#")][1:] == want
    # -C: nothing is flagged (F4 makes the log-odds 0), so the assignment columns are unchanged; --chimera-info adds five
    # columns between the alignment and the placement, --chimera-out gets a header and no line
    chi = str(tmp_path / "chimera.tsv")
    out4 = subprocess.run([cli, pre, fa, "-s", "1", "-C", "--chimera-info", "--chimera-out", chi, "--num-segment", "4", "-v"],
                          capture_output=True, text=True, timeout=300)
    assert out4.returncode == 0 and "0 flagged as chimera" in out4.stderr, out4.stderr
    l4 = [l.split("\t") for l in out4.stdout.split("\n") if l and not l.startswith("#")]
    assert l4[0][10:15] == ["seg5_taxon_id", "seg3_taxon_id", "seg5_taxon_anno", "seg3_taxon_anno", "chimera_lod"]
    W = E.Batch(D, 32)
    B.align(E.default_opts()); B.get_seed(E.default_opts())
    cres = B.check_chimera(W, E.default_opts(), num_seg=4)
    for row, w, c in zip(l4[1:], want, cres):
        assert row[:10] + row[15:] == w.split("\t")
        assert (int(row[10]), int(row[11])) == (int(c["seg5"]["a_node"]), int(c["seg3"]["a_node"]))
        assert row[12] == db.annos[int(row[10])] and row[13] == db.annos[int(row[11])] and row[14] == "0"
    with open(chi) as f:
        assert [l for l in f.read().split("\n") if l and not l.startswith("#")] == ["\t".join(l4[0])]
    # -a: the aligned reads as FASTA, 60 columns a line; --align-only: the alignment columns with a default placement
    alnf = str(tmp_path / "aln.fasta")
    out5 = subprocess.run([cli, pre, fa, "-s", "1", "-a", alnf, "--align-only"], capture_output=True, text=True, timeout=300)
    assert out5.returncode == 0, out5.stderr
    l5 = [l.split("\t") for l in out5.stdout.split("\n") if l and not l.startswith("#")]
    assert [r[:10] for r in l5[1:]] == [w.split("\t")[:10] for w in want]
    assert all(r[10:] == ["NULL", "nan", "-1", "UNASSIGNED", "nan", "nan", "nan", "nan"] for r in l5[1:])
    recs = open(alnf).read().split(">")[1:]
    assert len(recs) == len(reads)
    for i, (rec, w) in enumerate(zip(recs, want)):
        head, *body = rec.strip("\n").split("\n")
        f = w.split("\t")
        assert head == "read%d sample=%d;csStart=%s;csEnd=%s;" % (i, i % 3, f[6], f[7])
        assert "".join(body) == f[9] and all(len(x) == 60 for x in body[:-1]) and len(body[-1]) <= 60
    assert subprocess.run([cli, pre, fa, "-C", "--num-segment", "3"], capture_output=True).returncode != 0
    assert subprocess.run([cli, pre, fa, "-C", "--chimera-err", "0"], capture_output=True).returncode != 0
    W.close()
    # bad options fail like the reference's validation
    assert subprocess.run([cli, pre, fa, "-L", "30"], capture_output=True).returncode != 0
    assert subprocess.run([cli, str(tmp_path / "nodb"), fa], capture_output=True).returncode != 0
    B.close(); D.close()


def test_edge_cases_small_tree_short_reads_and_window():
    """fewer eligible nodes than max_nseed (NaN distances sort last), reads shorter than the seed
    length (full DP), a database resident only for a column window"""
    E = _engine()
    from oracle import oracle_py as O
    from hmmufotu_amd import synth
    db = get_db(16, 400, "GTR", dg_k=4, seed=21, n_match=150)    # 31 nodes < 50 seeds
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 10, 60, cols=250)
    seqs = [r.seq for r in reads]
    seqs[0] = seqs[0][:12]; vps[0] = 0                            # shorter than a 20-mer: no seed, full Viterbi
    seqs[1] = seqs[1][:1]; vps[1] = 0                             # a single base
    opts = E.default_opts()
    D, B = _run_stages(E, db, seqs, vps, opts)
    B.get_seed(opts); B.estimate_seq(opts); B.filter_placements(opts); B.place_seq(opts); B.calc_q_values(opts)
    out = B.alignments(want_align=True); cd, st, en = B.codes(); cnt, ids, sd, sN = B.seeds(); best = B.placements()
    for i, s in enumerate(seqs):
        a = H.align(s, vps[i])
        assert out["recs"][i]["status"] == 1 and a["ok"] and out["align"][i] == a["align"] and out["recs"][i]["cost"] == a["cost"], i
        res = T.assign(cd[i], int(st[i]), int(en[i]), O.default_opts())
        k = len(res["seed_ids"])
        assert k == db.n_nodes - 1 == cnt[i] and (ids[i, :k] == res["seed_ids"]).all(), i    # every non-root node, NaN last
        assert best[i]["n_cand"] == res["n"]
    B.close(); D.close()
    # column-window residency: same placements as the fully resident database; reads outside are refused
    db2 = get_db(120, 700, "GTR", dg_k=4)
    reads, vps = sim_reads(db2, 8, 120)
    md = E.model_desc(db2.model.type_id, db2.model.pi, db2.model.par, db2.dg_r)
    Dfull = E.Database.from_synth(db2)
    lo, hi = 20, 690
    Dwin = E.Database.from_arrays(db2.hmm, db2.parent, db2.blen, db2.seq, db2.up[:, lo:hi].copy(), db2.down[:, lo:hi].copy(), db2.height, md,
                                  db2.anno_id, db2.anno_dist, win_start=lo, win_len=hi - lo)
    res = []
    for D in (Dfull, Dwin):
        B = E.Batch(D, 8); B.set_reads([r.seq for r in reads], vps); B.assign(opts); res.append(B.placements().copy())
        full_recs = B.alignments(want_align=False)["recs"].copy(); B.close()
    for k in ("c_node", "a_node", "ratio", "wnr", "est_loglik"):
        assert np.array_equal(res[0][k], res[1][k]), k
    Dtiny = E.Database.from_arrays(db2.hmm, db2.parent, db2.blen, db2.seq, db2.up[:, 300:400].copy(), db2.down[:, 300:400].copy(), db2.height, md,
                                   win_start=300, win_len=100)
    # reads that align outside the resident window are marked per read (status 16), the batch goes on
    B = E.Batch(Dtiny, 8); B.set_reads([r.seq for r in reads], vps)
    B.assign(opts)
    recs = B.alignments(want_align=False)["recs"]
    assert (recs["status"] == 16).all() and (B.placements()["c_node"] == -1).all()
    B.close(); Dtiny.close()
    # stray reads among good ones: a window that starts just after the left-most alignment(s)
    starts = np.sort(np.unique(full_recs["cs_start"] - 1))
    assert len(starts) > 1
    lo2 = int(starts[1])
    out_ = (full_recs["cs_start"] - 1) < lo2
    assert 0 < out_.sum() < len(reads)
    Dw2 = E.Database.from_arrays(db2.hmm, db2.parent, db2.blen, db2.seq, db2.up[:, lo2:hi].copy(), db2.down[:, lo2:hi].copy(), db2.height, md,
                                 db2.anno_id, db2.anno_dist, win_start=lo2, win_len=hi - lo2)
    B = E.Batch(Dw2, 8); B.set_reads([r.seq for r in reads], vps); B.assign(opts)
    recs = B.alignments(want_align=False)["recs"]; got = B.placements()
    assert (recs["status"][out_] == 16).all() and (recs["status"][~out_] == 1).all() and (got["c_node"][out_] == -1).all()
    assert np.array_equal(recs["cs_start"], full_recs["cs_start"])          # still aligned, just not placed
    for k in ("c_node", "a_node", "ratio", "wnr", "est_loglik"):
        assert np.array_equal(got[k][~out_], res[0][k][~out_]), k
    assert len(B.format_tsv(["r%d" % i for i in range(8)], None, db2.annos).strip("\n").split("\n")) == int((~out_).sum())
    B.close(); Dw2.close(); Dfull.close(); Dwin.close()


def test_abi_returns_status_codes_and_never_aborts():
    """The failure of gpurun_out/t_r2e.log (round 2) and its neighbours, through the C ABI: (1) reads outside the resident column window —
    aligned ones and hu_batch_set_aligned ones, whose empty region once sat at column 0 and sent k_estimate_prod 40 columns before the
    first resident message of node 0 (a GPU memory fault: the process died inside hu_assign_batch) — come back with a per-read status and
    NO seeds, and the batch goes on; (2) an over-large max_reads and an over-large n are a status; (3) C++ exceptions raised inside the
    stages — on the calling thread and inside a worker of the host pool (fault injection knob) — are a status + message, after which
    the same batch still works."""
    E = _engine()
    import ctypes as C
    lib = E.load_library()
    lib.hu_last_error.restype = C.c_char_p
    db = get_db(120, 700, "GTR", dg_k=4)
    reads, vps = sim_reads(db, 8, 120)
    md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r)
    opts = E.default_opts()
    Dfull = E.Database.from_synth(db)
    B = E.Batch(Dfull, 8); B.set_reads([r.seq for r in reads], vps); B.assign(opts)
    full = B.alignments(want_align=False)["recs"].copy(); want = B.placements().copy(); cd, st, en = B.codes(); B.close()
    lo = int(np.sort(np.unique(full["cs_start"] - 1))[1]); hi = 690
    out_ = (full["cs_start"] - 1) < lo
    assert 0 < out_.sum() < len(reads) and lo > 0
    Dw = E.Database.from_arrays(db.hmm, db.parent, db.blen, db.seq, db.up[:, lo:hi].copy(), db.down[:, lo:hi].copy(), db.height, md,
                                db.anno_id, db.anno_dist, win_start=lo, win_len=hi - lo)
    # (1a) aligned here: status 16, no seeds, nothing placed; the others as on the fully resident database
    B = E.Batch(Dw, 8); B.set_reads([r.seq for r in reads], vps); B.assign(opts)
    recs = B.alignments(want_align=False)["recs"]; cnt, ids, _, _ = B.seeds(); got = B.placements()
    assert (recs["status"][out_] == 16).all() and (cnt[out_] == 0).all() and (cnt[~out_] > 0).all() and (got["c_node"][out_] == -1).all()
    assert np.array_equal(got["c_node"][~out_], want["c_node"][~out_])
    # (1b) handed over aligned (hu_batch_set_aligned), regions outside the window / inverted / beyond the consensus: refused per read
    st2, en2 = st.copy(), en.copy()
    k = int(np.nonzero(~out_)[0][0])
    st2[k], en2[k] = 0, lo - 1                                              # wholly left of the window
    B.set_aligned(cd, st2, en2); B.assign(opts)
    recs = B.alignments(want_align=False)["recs"]; cnt, _, _, _ = B.seeds(); got2 = B.placements()
    bad = out_.copy(); bad[k] = True
    assert (recs["status"][bad] == 0).all() and (cnt[bad] == 0).all() and (got2["c_node"][bad] == -1).all()
    assert np.array_equal(got2["c_node"][~bad], want["c_node"][~bad])
    st3, en3 = st2.copy(), en2.copy(); st3[k], en3[k] = 650, 10 ** 6
    B.set_aligned(cd, st3, en3); B.assign(opts)
    assert (B.placements()["c_node"][bad] == -1).all()
    # (2) sizes
    hb = C.c_void_p()
    assert lib.hu_batch_create(Dw.h, C.c_int(2 ** 31 - 1), C.byref(hb)) == 0          # nothing is sized by max_reads before reads arrive
    lib.hu_batch_destroy(hb)
    assert lib.hu_batch_create(Dw.h, C.c_int(0), C.byref(hb)) == -1
    offs = np.zeros(10, np.int64)
    assert lib.hu_batch_set_reads(B.h, C.c_int(9), b"ACGT", offs.ctypes.data_as(C.c_void_p), None, None, None, None) == -1    # n > max_reads
    assert b"bad argument" in lib.hu_last_error()
    # (3) exceptions: a worker of the host pool (needs >= 512 reads to spread over the pool), the calling thread, the formatter's pool
    n = 600
    rs = [reads[i % 8].seq for i in range(n)]; vv = np.stack([vps[i % 8] for i in range(n)])
    Bb = E.Batch(Dfull, n); Bb.set_reads(rs, vv)
    for code, what, status in ((1, b"memory", -4), (2, b"injected", -4)):
        Bb.set_knob("inject_fault", code)
        rc = lib.hu_assign_batch(Bb.h, C.byref(opts))
        assert rc == status and what in lib.hu_last_error(), (code, rc, lib.hu_last_error())
    Bb.set_knob("inject_fault", 3)
    Bb.assign(opts)
    txt = C.c_char_p(); lib.hu_batch_format_tsv_ptr.restype = C.c_int64
    ids_ = (C.c_char_p * n)(*[b"r%d" % i for i in range(n)])
    assert lib.hu_batch_format_tsv_ptr(Bb.h, ids_, None, None, None, C.c_int(0), C.c_int(0), C.byref(txt)) == -5 and b"injected" in lib.hu_last_error()
    Bb.set_knob("inject_fault", 0)
    Bb.assign(opts)                                                          # the same batch object still works
    assert np.array_equal(Bb.placements()["c_node"][:8], want["c_node"])
    assert len(Bb.format_tsv(["r%d" % i for i in range(n)]).strip("\n").split("\n")) == n
    Bb.close(); B.close(); Dw.close(); Dfull.close()


def test_topk_sampled_threshold_path(monkeypatch):
    """k_seed_topk's fast path (threshold bin estimated from one eighth of the pairs) on trees far below its
    default size limit: seed ids, order and (d, N) stay bit-exact, whether the estimate suffices or the exact
    two-pass path has to take over"""
    monkeypatch.setenv("HU_TOPK_FAST_MIN", "1")
    test_sep_parity(dict(model="GTR", dg_k=4, n_leaves=1500, cs_len=700, read_len=150, seed_order=0))
    test_sep_parity(dict(model="K80", dg_k=2, n_leaves=80, cs_len=500, read_len=100, seed_order=0))
    test_topk_degenerate_tie_mass(0)


@pytest.mark.parametrize("mean_blen,max_nseed,read_len,wide", [(0.05, 10, 60, 0), (0.05, 3, 60, 1), (1e-9, 10, 60, 0), (0.002, 8, 60, 0), (0.05, 10, 300, 0), (0.0005, 10, 120, 0)])
def test_distance_only_scan_and_its_topk(mean_blen, max_nseed, read_len, wide):
    """Large trees: the scan keeps the distance d alone (8 bits, 16 when a read has more than 255 bases or on request) and per-block
    minima; the top-k recomputes the (d, N) of its candidates from the bit-planes.  Seed ids, their order and (d, N) equal the oracle's
    and the pair-matrix path's — with ordinary distances, with distances that tie across many blocks, and with every node at distance 0
    (the exact recomputation takes over); hu_batch_get_pdist checks the scan's matrix against the planes, node by node."""
    E = _engine()
    long_ = read_len > 255
    db = get_db(2600, 700 if long_ else 200, "JC69", dg_k=0, seed=11, mean_blen=mean_blen, n_match=450 if long_ else 120)      # 5,199 nodes = 21 blocks of 256
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 12 if long_ else 24, read_len, amplicon=True, cols=650 if long_ else 140)
    opts = E.default_opts(max_nseed=max_nseed, seed_order=0)           # (dist, node id): the order of the distance-only path
    D, B = _run_stages(E, db, reads, vps, opts)
    assert D.n_nodes // 256 >= 2 * max_nseed
    B.set_knob("pairs32", wide)
    B.get_seed(opts)
    cd, st, en = B.codes(); cnt, ids, sd, sN = B.seeds()
    for i in range(len(reads)):
        oid, od, oN, _ = T.get_seed(cd[i], int(st[i]), int(en[i]), tie=0, max_n=max_nseed)
        assert cnt[i] == len(oid) == max_nseed and (ids[i, :cnt[i]] == oid).all() and (sd[i, :cnt[i]] == od).all() and (sN[i, :cnt[i]] == oN).all(), i
    for i in (0, len(reads) - 1):
        d, N = B.pdist(i)                      # from the planes, and the scan's row checked against them inside
        od, oN = T.pdist_all(cd[i], int(st[i]), int(en[i]))
        assert (d == od).all() and (N == oN).all()
    B.estimate_seq(opts); e1 = B.estimates()
    B.set_knob("scan_pairs", 1); B.get_seed(opts)
    c2, i2, d2, n2 = B.seeds()
    assert np.array_equal(c2, cnt) and np.array_equal(i2[:, :max_nseed], ids[:, :max_nseed]) and np.array_equal(d2[:, :max_nseed], sd[:, :max_nseed])
    B.estimate_seq(opts); e2 = B.estimates()
    for a, b in zip(e1, e2):
        assert np.array_equal(a[:, :max_nseed], b[:, :max_nseed], equal_nan=True)      # the parents' (d, N): same ratios either way
    B.close(); D.close()


@pytest.mark.parametrize("kind", ["ends", "holes"])
def test_seed_stage_with_partial_sequences(kind, capfd):
    """Reference sequences that cover only part of the alignment.  "ends": half of the nodes lack a prefix or a suffix, as partial 16S
    sequences do — for many of them a read shares no position at all (N = 0, d = 0): the scan keeps such nodes out of its block minima
    and the top-k out of its candidates by their position intervals (HuDbDev::nodeCover), and every read stays on the block path.
    "holes": the stretch is cut out of the middle, which the intervals cannot see: the top-k meets candidates with N = 0, raises its
    limit past them, and takes the exact recomputation only when that is not enough.  Either way N differs widely between nodes, so the
    bound D1 = floor(d1 L / N1) opens far beyond the first candidate set.  Seeds, order and (d, N) against the oracle and against the
    pair-matrix path."""
    import copy, re
    E = _engine()
    from oracle import oracle_py as O
    db0 = get_db(2600, 300, "JC69", dg_k=0, seed=11, n_match=200)
    db = copy.copy(db0)
    rng = np.random.default_rng(17)
    seq = db0.seq.copy()
    for i in rng.choice(db.n_nodes, size=int(db.n_nodes * 0.5), replace=False):
        if kind == "ends":
            cut = int(rng.integers(40, db.cs_len - 40))
            if rng.random() < 0.5: seq[i, :cut] = -2
            else: seq[i, cut:] = -2
        else:
            a = int(rng.integers(0, db.cs_len - 40)); b = int(rng.integers(a + 20, db.cs_len))
            seq[i, a:b] = -2
    db.seq = seq
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db0, 24, 100)
    opts = E.default_opts(max_nseed=10, seed_order=0)
    D, B = _run_stages(E, db, reads, vps, opts)
    B.set_knob("trace", 1)
    B.set_knob("scan_pairs", -1)        # with so many partial sequences the engine would choose the pair matrix by itself: force the distance-only path
    capfd.readouterr()
    B.get_seed(opts)
    err = capfd.readouterr().err
    m = re.search(r"(\d+) of (\d+) reads on the block path .* (\d+) by the exact recomputation", err)
    assert m and int(m.group(2)) == len(reads), err
    if kind == "ends":
        assert int(m.group(3)) == 0, err
    cd, st, en = B.codes(); cnt, ids, sd, sN = B.seeds()
    for i in range(len(reads)):
        oid, od, oN, _ = T.get_seed(cd[i], int(st[i]), int(en[i]), tie=0, max_n=10)
        assert cnt[i] == len(oid) and (ids[i, :cnt[i]] == oid).all() and (sd[i, :cnt[i]] == od).all() and (sN[i, :cnt[i]] == oN).all(), i
    B.estimate_seq(opts); e1 = B.estimates()
    B.set_knob("scan_pairs", 0)                                   # the engine's own choice: the pair matrix when many sequences lack an end
    capfd.readouterr(); B.get_seed(opts); B.estimate_seq(opts)        # (a hole in the middle does not show in a node's intervals)
    assert ("block path" in capfd.readouterr().err) == (kind == "holes")
    c2, i2, d2, n2 = B.seeds(); e2 = B.estimates()
    assert np.array_equal(i2[:, :10], ids[:, :10]) and np.array_equal(d2[:, :10], sd[:, :10]) and np.array_equal(n2[:, :10], sN[:, :10])
    for a, b in zip(e1, e2):
        assert np.array_equal(a[:, :10], b[:, :10], equal_nan=True)
    print("partial sequences (%s):" % kind, m.group(0))
    B.close(); D.close()


def test_seed_paths_agree_on_a_large_tree():
    """The two seed paths on a tree of 39,999 nodes with the default 50 seeds, 2,048 reads (amplicon reads, reads scattered over the
    consensus, a few reads cut down to a handful of bases): distance-only scan + top-k from the planes against pair matrix + k_seed_topk —
    same seed ids, order, (d, N) and the same estimates (the parents' pairs) for every read."""
    E = _engine()
    import torch
    from hmmufotu_amd import synth, synth_gpu
    db, up, down = synth_gpu.make_db_gpu(20000, 1500, "GTR", dg_k=0, seed=21, device="cuda:0", log=lambda *a: None)
    reads = synth_gpu.simulate_reads_gpu(db, up, down, 1536, 150, seed=5, amplicon_start=200, amplicon_cols=170, device="cuda:0")
    reads += synth_gpu.simulate_reads_gpu(db, up, down, 512, 150, seed=6, amplicon_start=0, amplicon_cols=170, uniform=True, device="cuda:0")
    for i in range(0, 40, 5):                                  # short reads: few bases, many ties
        r = reads[i]; reads[i] = synth.SimRead(r.seq[:12 + i], r.cols[:12 + i], r.node, r.rc, r.cs_start, r.cs_end)
    vps = np.stack([synth.read_vpaths(db.hmm, r) for r in reads])
    md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r)
    D = E.Database.from_arrays(db.hmm, db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, db.anno_id, msgs_on_device=True)
    assert D.n_nodes // 256 >= 100
    B = E.Batch(D, len(reads))
    opts = E.default_opts(seed_order=0)
    B.set_reads([r.seq for r in reads], vps); B.align(opts)
    B.set_knob("trace", 1)
    B.get_seed(opts); a = B.seeds(); B.estimate_seq(opts); ea = B.estimates()
    B.set_knob("scan_pairs", 1)
    B.get_seed(opts); b = B.seeds(); B.estimate_seq(opts); eb = B.estimates()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for x, y in zip(ea, eb):
        assert np.array_equal(x, y, equal_nan=True)
    assert (a[0] == 50).all()
    d, N = B.pdist(3)
    B.set_knob("scan_pairs", 0); B.set_knob("topk_general", 1)    # all 2,048 reads through the general launch: its 1,024 workgroups take two each
    B.get_seed(opts); g = B.seeds(); B.estimate_seq(opts); eg = B.estimates()
    for x, y in zip(a, g):
        assert np.array_equal(x, y)
    for x, y in zip(ea, eg):
        assert np.array_equal(x, y, equal_nan=True)
    B.set_knob("topk_general", 0); B.get_seed(opts)
    d2, N2 = B.pdist(3)                                            # the scan's own row checked inside
    assert np.array_equal(d, d2) and np.array_equal(N, N2)
    B.close(); D.close()


def test_streaming_sep_kernels(monkeypatch):
    """the one-wave-per-unit streaming estimate / place kernels (regions beyond 3,072 columns) and the per-site
    log() estimate kernel give the same placements as the table-driven workgroup kernels"""
    monkeypatch.setenv("HU_STREAMING_SEP", "1")
    test_sep_parity(dict(model="GTR", dg_k=4, n_leaves=150, cs_len=700, read_len=150))
    monkeypatch.delenv("HU_STREAMING_SEP")
    monkeypatch.setenv("HU_EST_VAR", "2")
    test_sep_parity(dict(model="HKY85", dg_k=4, n_leaves=100, cs_len=700, read_len=100))


@pytest.mark.parametrize("order", [0, 1])
def test_topk_degenerate_tie_mass(order):
    """thousands of nodes at exactly the same distance: the bounded slow path of k_seed_topk must still
    return the (dist, id) order (order 0); in the reference's order (1) every element of the row is a stopper of every partition and
    the first 50 places are purely libstdc++'s tie permutation — k_seed_refsort against the oracle's literal std::sort"""
    E = _engine()
    from oracle import oracle_py as O
    db = get_db(2600, 200, "JC69", dg_k=0, seed=5, mean_blen=1e-9, n_match=120)   # all sequences identical up to gaps
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 4, 60, amplicon=True, cols=140)
    opts = E.default_opts(seed_order=order)
    D, B = _run_stages(E, db, reads, vps, opts)
    B.get_seed(opts)
    cd, st, en = B.codes(); cnt, ids, sd, sN = B.seeds()
    for i in range(len(reads)):
        oid, od, oN, _ = T.get_seed(cd[i], int(st[i]), int(en[i]), tie=order)
        assert cnt[i] == len(oid) and (ids[i, :cnt[i]] == oid).all() and (sd[i, :cnt[i]] == od).all()
    B.close(); D.close()


def test_pe_full_pipeline_against_oracle_batch():
    """paired-end reads through the whole path vs the oracle's per-read task (alignSeq x2, merge, SEP)"""
    E = _engine()
    from hmmufotu_amd import synth
    from oracle import oracle_py as O
    db = get_db(120, 1400, "GTR", dg_k=4)
    _, H, T = oracle_objects(db)
    rng = np.random.default_rng(15)
    ins = synth.simulate_reads(db, 20, 100000, rng, amplicon_start=60, amplicon_cols=1200, jitter=20)
    fw, rv, vf, vr = [], [], [], []
    for r in ins:
        n = len(r.seq)
        f = synth.SimRead(r.seq[:110], r.cols[:110], r.node, r.rc, r.cs_start, r.cs_end)
        m = synth.SimRead(r.seq[n - 110:], r.cols[n - 110:], r.node, r.rc, r.cs_start, r.cs_end)
        fw.append(f.seq); rv.append(m.seq); vf.append(synth.read_vpaths(db.hmm, f)); vr.append(synth.read_vpaths(db.hmm, m))
    opts = E.default_opts()
    D = E.Database.from_synth(db); B = E.Batch(D, 32)
    B.set_reads(fw, np.stack(vf), rv, np.stack(vr)); B.assign(opts)
    best = B.placements(); recs = B.alignments(want_align=False)["recs"]
    ref = O.pipeline_batch(H, T, fw, np.stack(vf), mates=rv, mvpaths=np.stack(vr), threads=2, want_cands=True)
    assert (recs["status"] == ref["aln_ints"][:, 7]).all()
    assert (recs["cs_start"] == ref["aln_ints"][:, 4]).all() and (recs["cs_end"] == ref["aln_ints"][:, 5]).all()
    assert np.array_equal(recs["cost"], ref["cost"])
    assert (best["n_cand"] == ref["n_cand"]).all()
    tot = classify_batch(ref, B.candidates(), best, db.parent)
    assert tot["set_differs"] == 0 and tot["swaps_unexplained"] == 0 and tot["best_unexplained"] == 0, tot
    print("PE parity", tot)
    B.close(); D.close()


def test_wide_region_kernels():
    """paired-end reads whose merged alignment spans ~2,400 CS columns: the 4-wave / 12-sites-per-thread placement
    kernel and the 512-thread / 6-sites-per-thread estimate kernel (the 250 bp single-end benchmark uses the 2-wave / 256-thread ones)"""
    E = _engine()
    from hmmufotu_amd import synth
    from oracle import oracle_py as O
    db = get_db(80, 2800, "GTR", dg_k=4)
    _, H, T = oracle_objects(db)
    rng = np.random.default_rng(21)
    ins = synth.simulate_reads(db, 12, 100000, rng, amplicon_start=100, amplicon_cols=2400, jitter=20)
    fw, rv, vf, vr = [], [], [], []
    for r in ins:
        n = len(r.seq)
        f = synth.SimRead(r.seq[:120], r.cols[:120], r.node, r.rc, r.cs_start, r.cs_end)
        m = synth.SimRead(r.seq[n - 120:], r.cols[n - 120:], r.node, r.rc, r.cs_start, r.cs_end)
        fw.append(f.seq); rv.append(m.seq); vf.append(synth.read_vpaths(db.hmm, f)); vr.append(synth.read_vpaths(db.hmm, m))
    opts = E.default_opts()
    D = E.Database.from_synth(db); B = E.Batch(D, 16)
    B.set_reads(fw, np.stack(vf), rv, np.stack(vr)); B.assign(opts)
    best = B.placements(); recs = B.alignments(want_align=False)["recs"]
    ref = O.pipeline_batch(H, T, fw, np.stack(vf), mates=rv, mvpaths=np.stack(vr), threads=2, want_cands=True)
    ok = recs["status"] == 1
    assert ok.sum() >= 8 and (recs["status"] == ref["aln_ints"][:, 7]).all()
    span = (recs["cs_end"] - recs["cs_start"])[ok]
    assert span.max() > 2048, span                                 # beyond the 2-wave placement kernel's 12 x 128 sites and the 256-thread estimate kernel's 8 x 256
    assert np.array_equal(recs["cost"], ref["cost"])
    assert (best["n_cand"] == ref["n_cand"]).all()
    tot = classify_batch(ref, B.candidates(), best, db.parent)
    assert tot["set_differs"] == 0 and tot["swaps_unexplained"] == 0 and tot["best_unexplained"] == 0, tot
    print("wide-region parity", tot)
    # the shipped estimate kernel for such regions runs 512 threads x 6 sites; est_var = 4 is the 256 x 12 form: same estimates
    er, ew, el = B.estimates()
    B.set_knob("est_var", 4); B.assign(opts)
    er4, ew4, el4 = B.estimates()
    assert np.array_equal(er, er4, equal_nan=True) and np.array_equal(ew, ew4, equal_nan=True) and np.nanmax(np.abs(el - el4) / np.maximum(np.abs(el), 1.0)) < 1e-12
    # the shipped placement kernel for such regions keeps one component of the v message in LDS (two candidates per CU); place_var = 9 is the
    # all-register form (one per CU): same candidates, same iteration counts, same lengths
    c0 = B.candidates()
    B.set_knob("est_var", 0); B.set_knob("place_var", 9); B.assign(opts)
    c9 = B.candidates()
    assert np.array_equal(c0["c_node"], c9["c_node"]) and np.array_equal(c0["iters"], c9["iters"])
    assert np.abs(c0["ratio"] - c9["ratio"]).max() < 1e-12 and np.abs(c0["wnr"] - c9["wnr"]).max() < 1e-12
    B.close(); D.close()


@pytest.mark.parametrize("num_seg,model,dg_k", [(2, "GTR", 4), (4, "TN93", 0), (6, "JC69", 0)])
def test_chimera_check_parity(num_seg, model, dg_k):
    """-C (src/hmmufotu.cpp:653-691): per-segment estimate/filter/place on the common seeds, pooled 5'/3' winners and
    the log-odds against each other's branch, vs the oracle on the same codes and seeds.  Half of the inputs are
    real chimeras spliced from two reads at the middle of the region."""
    E = _engine()
    from oracle import oracle_py as O, parity
    db = get_db(150, 700, model, dg_k=dg_k)
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 24, 150)
    opts = E.default_opts()
    D, B = _run_stages(E, db, reads, vps, opts)
    cd, st, en = B.codes()
    cd = cd.copy()
    for i in range(0, len(reads) - 1, 2):           # splice: 5' half of read i, 3' half of read i+1 over read i's region
        mid = (int(st[i]) + int(en[i])) // 2
        cd[i, mid:] = cd[i + 1, mid:]
    B.set_aligned(cd, st, en)
    B.get_seed(opts)
    cnt, ids, _, _ = B.seeds()
    W = E.Batch(D, len(reads))
    res = B.check_chimera(W, opts, num_seg=num_seg)
    # the seeded batch is untouched and goes on to the ordinary placement
    B.estimate_seq(opts); B.filter_placements(opts); B.place_seq(opts); B.calc_q_values(opts)
    best = B.placements()
    oo = O.default_opts()
    tie_diff = 0
    for i in range(len(reads)):
        o = T.chimera(cd[i], int(st[i]), int(en[i]), oo, num_seg=num_seg, seeds=ids[i, :cnt[i]])
        g = res[i]
        assert bool(g["checked"]) == o["checked"] and o["checked"], i
        assert bool(g["is_chimera"]) == o["is_chimera"], i
        assert (g["n_seg5"], g["n_seg3"]) == (o["n5"], o["n3"]), i
        assert g["lod"] == o["lod"] == 0.0, i                       # F4: all placed logliks are segLen * const
        assert g["alt5_loglik"] == g["seg5"]["loglik"] and g["alt3_loglik"] == g["seg3"]["loglik"]
        seg_len = (int(en[i]) - int(st[i]) + 1) // num_seg
        assert g["seg5_end"] - g["seg5_start"] + 1 == seg_len and g["seg3_end"] - g["seg3_start"] + 1 == seg_len
        for side in ("seg5", "seg3"):
            gp, op = g[side], o[side]
            assert _rel(gp["loglik"], op["loglik"]) < 1e-12
            if int(gp["c_node"]) != op["c"] or int(g[side + "_start"]) != op["start"]:
                # the winner sits at a fixed position of the pooled sequence; another node there means two candidates
                # of that segment swapped in filterPlacements order: accepted only for the documented near-tie (both attach at
                # the same tree node, oracle estimates within 1e-9: oracle/parity.py), checked difference by difference
                assert int(g[side + "_start"]) == op["start"], (i, side)
                s0, e0 = op["start"], op["end"]
                dd, NN = T.pdist_all(cd[i], s0, e0)
                eo = {}
                for nd_ in (op["c"], int(gp["c_node"])):
                    eo[nd_] = T.estimate(cd[i], s0, e0, nd_, float(np.float64(dd[nd_]) / np.float64(NN[nd_])))
                assert parity.explained_swap(op["c"], int(gp["c_node"]), {k_: v_["loglik"] for k_, v_ in eo.items()},
                                             {k_: v_["ratio"] for k_, v_ in eo.items()}, db.parent), (i, side, eo)
                tie_diff += 1
                continue
            assert (int(gp["p_node"]), int(gp["a_node"])) == (op["p"], op["a"]), (i, side)
            assert int(g[side + "_end"]) == op["end"]
            assert _rel(gp["est_loglik"], op["est_loglik"]) < REL
            assert abs(gp["ratio"] - op["ratio"]) <= REL * max(abs(op["ratio"]), 1e-3)
            assert abs(gp["wnr"] - op["wnr"]) <= REL * max(abs(op["wnr"]), 1e-3)
        assert best[i]["n_cand"] >= 1
    print("chimera parity: explained near-tie order differences", tie_diff, "of", 2 * len(reads))
    # argument checks of src/hmmufotu.cpp:325-340
    for bad in (dict(num_seg=3), dict(num_seg=8), dict(max_chimera_error=0.0), dict(min_chimera_lod=-1.0)):
        with pytest.raises(E.EngineError):
            B.check_chimera(W, opts, **bad)
    W.close(); B.close(); D.close()


def test_given_seed_stage_matches_pdist():
    """hu_seed_batch_given: the distance of each given node over the CURRENT region, optionally to another node."""
    E = _engine()
    db = get_db(100, 700, "GTR", dg_k=0)
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 6, 120)
    opts = E.default_opts()
    D, B = _run_stages(E, db, reads, vps, opts)
    cd, st, en = B.codes()
    half = (st + en) // 2
    B.set_aligned(cd, st, half)
    rng = np.random.default_rng(5)
    nonroot = np.flatnonzero(np.asarray(db.parent) >= 0)
    ids = rng.choice(nonroot, (len(reads), 7)).astype(np.int32)
    other = rng.integers(0, len(db.parent), ids.shape).astype(np.int32)
    n_seeds = np.array([7, 0, 3, 7, 1, 5], np.int32)
    for dist_ids in (None, other):
        B.get_seed_given(n_seeds, ids, dist_ids)
        cnt, gid, gd, gN = B.seeds()
        assert (cnt == n_seeds).all()
        for i in range(len(reads)):
            od, oN = T.pdist_all(cd[i], int(st[i]), int(half[i]))
            k = n_seeds[i]
            src = ids[i, :k] if dist_ids is None else other[i, :k]
            assert (gid[i, :k] == ids[i, :k]).all()
            assert (gd[i, :k] == od[src]).all() and (gN[i, :k] == oN[src]).all()
    root = int(np.flatnonzero(np.asarray(db.parent) < 0)[0])
    ids[0, 0] = root
    with pytest.raises(E.EngineError):
        B.get_seed_given(n_seeds, ids)
    B.close(); D.close()


def test_chimera_check_edge_cases():
    """Regions too short to cut, one-column segments, the weighted method with a height bound, and a work batch on the
    wrong database."""
    E = _engine()
    from oracle import oracle_py as O, parity
    db = get_db(100, 700, "HKY85", dg_k=4)
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 8, 120)
    mh = float(np.quantile(db.height, 0.8))
    opts = E.default_opts(weighted=1, max_height=mh, max_nseed=12)
    D, B = _run_stages(E, db, reads, vps, opts)
    cd, st, en = B.codes()
    st = st.copy(); en = en.copy()
    en[0] = st[0] + 2            # 3 columns: no segment with num_seg = 4, one-column segments with num_seg = 2
    en[1] = st[1] + 3            # 4 columns: one-column segments with num_seg = 4
    B.set_aligned(cd, st, en)
    B.get_seed(opts)
    cnt, ids, _, _ = B.seeds()
    W = E.Batch(D, len(reads))
    oo = O.default_opts(weighted=1, maxHeight=mh, maxNSeed=12)
    for num_seg in (2, 4):
        res = B.check_chimera(W, opts, num_seg=num_seg)
        for i in range(len(reads)):
            o = T.chimera(cd[i], int(st[i]), int(en[i]), oo, num_seg=num_seg, seeds=ids[i, :cnt[i]])
            g = res[i]
            assert bool(g["checked"]) == o["checked"], (num_seg, i)
            if i == 0 and num_seg == 4:
                assert not o["checked"] and g["seg5"]["c_node"] == -1 and np.isnan(g["lod"])
                continue
            assert (g["n_seg5"], g["n_seg3"]) == (o["n5"], o["n3"]) and g["lod"] == o["lod"] and not g["is_chimera"]
            for side in ("seg5", "seg3"):
                if int(g[side]["c_node"]) == o[side]["c"] and int(g[side + "_start"]) == o[side]["start"]:
                    assert int(g[side]["a_node"]) == o[side]["a"]
                    assert abs(g[side]["wnr"] - o[side]["wnr"]) <= REL * max(abs(o[side]["wnr"]), 1e-3)
                else:   # near-tie swap of the segment's estimates: same attachment node, oracle estimates within 1e-9 (oracle/parity.py)
                    assert int(g[side + "_start"]) == o[side]["start"]
                    s0, e0 = o[side]["start"], o[side]["end"]
                    dd, NN = T.pdist_all(cd[i], s0, e0)
                    eo = {nd_: T.estimate(cd[i], s0, e0, nd_, float(np.float64(dd[nd_]) / np.float64(NN[nd_])), weighted=True)
                          for nd_ in (o[side]["c"], int(g[side]["c_node"]))}
                    assert parity.explained_swap(o[side]["c"], int(g[side]["c_node"]), {k_: v_["loglik"] for k_, v_ in eo.items()},
                                                 {k_: v_["ratio"] for k_, v_ in eo.items()}, db.parent), (num_seg, i, side, eo)
    db2 = get_db(60, 700, "JC69", dg_k=0)
    D2 = E.Database.from_synth(db2)
    W2 = E.Batch(D2, len(reads))
    with pytest.raises(E.EngineError):
        B.check_chimera(W2, opts)
    W3 = E.Batch(D, 2)
    with pytest.raises(E.EngineError):
        B.check_chimera(W3, opts)
    with pytest.raises(E.EngineError):
        B.check_chimera(B, opts)
    B.set_reads([], np.zeros((0, 2, 6), np.int32))          # an empty batch passes through every stage
    B.align(opts); B.get_seed(opts)
    assert len(B.check_chimera(W, opts)) == 0
    W3.close(); W2.close(); D2.close(); W.close(); B.close(); D.close()


@pytest.mark.parametrize("read_len,lo,hi,gap_cap", [(220, 1024, 1536, 1280), (150, 512, 1024, 768)])
def test_split_gap_base_placement_kernel(monkeypatch, capfd, read_len, lo, hi, gap_cap):
    """Regions of 513 .. 1,536 columns with <= 256 bases (the 150 / 250 bp benchmark shapes) take the placement kernel that
    walks a read's gap sites and base sites in separate slots (k_site_count / k_site_perm + k_place_blk GS = 6 / 10): against
    the oracle, and against the kernel that takes the sites in column order (HU_PLACE_NOSPLIT)."""
    E = _engine()
    from oracle import oracle_py as O
    db = get_db(60, 2000, "GTR", dg_k=4, seed=5)
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 10, read_len)
    opts = E.default_opts(max_nseed=16)
    D, B = _run_stages(E, db, reads, vps, opts)
    B.get_seed(opts); B.estimate_seq(opts); B.filter_placements(opts)
    cd, st, en = B.codes()
    span = en - st + 1
    nb = np.array([(cd[i, st[i]:en[i] + 1] >= 0).sum() for i in range(len(reads))])
    assert (span.max() > lo) and (span <= hi).all() and (nb <= 256).all() and (span - nb <= gap_cap).all(), (span, nb)
    B.set_knob("trace", 1)
    B.place_seq(opts); B.calc_q_values(opts)
    split = B.candidates(); best = B.placements()
    B.set_knob("place_nosplit", 1)
    B.place_seq(opts); B.calc_q_values(opts)
    plain = B.candidates()
    B.set_knob("place_nosplit", 0); B.set_knob("trace", 0)
    # regions of <= 1,024 sites: the shipped kernel keeps the v message in LDS (three waves per SIMD); place_var = 6 is the same kernel
    # with v in registers (two waves per SIMD) — same arithmetic up to the compiler's choice of fused multiply-adds in the two instances
    B.set_knob("place_var", 6)
    B.place_seq(opts); B.calc_q_values(opts)
    regs = B.candidates()
    B.set_knob("place_var", 0)
    assert np.array_equal(split["c_node"], regs["c_node"]) and np.array_equal(split["iters"], regs["iters"])
    assert np.abs(split["ratio"] - regs["ratio"]).max() < 1e-12 and np.abs(split["wnr"] - regs["wnr"]).max() < 1e-12
    err = capfd.readouterr().err
    assert "gap/base split slots" in err and "column order" in err, err      # both kernels ran
    assert np.array_equal(split["c_node"], plain["c_node"]) and np.array_equal(split["iters"], plain["iters"])
    assert np.abs(split["ratio"] - plain["ratio"]).max() < 1e-8 and np.abs(split["wnr"] - plain["wnr"]).max() < 1e-8
    oo = O.default_opts(maxNSeed=16)
    worst = 0.0
    for i in range(len(reads)):
        res = T.assign(cd[i], int(st[i]), int(en[i]), oo)
        lo, hi = split["offs"][i], split["offs"][i + 1]
        assert hi - lo == res["n"]
        oc = {int(n_[0]): (v[0], v[1], int(n_[3])) for n_, v in zip(res["nodes"], res["vals"])}
        for c in range(lo, hi):
            r0, w0_, it = oc[int(split["c_node"][c])]
            worst = max(worst, abs(split["ratio"][c] - r0) / max(abs(r0), 1e-3), abs(split["wnr"][c] - w0_) / max(abs(w0_), 1e-3))
            assert (int(split["iters"][c]) & 0xff) == it, (i, c)
        assert best[i]["n_cand"] == res["n"]
    assert worst < REL, worst
    B.close(); D.close()


def test_streaming_kernels_on_a_large_tree():
    """Regions beyond 3,072 columns (BASELINE config 5's 2 x 300 bp pairs span ~3,020 +- jitter) take the streaming estimate / place
    kernels; here on a tree of 16,999 nodes, large enough for the sampled-threshold top-k path as well (config 5's own code path,
    reduced in leaves only), against the oracle on the messages of the window the pairs touch."""
    E = _engine()
    import torch
    from hmmufotu_amd import synth, synth_gpu
    from oracle import oracle_py as O
    db, up, down = synth_gpu.make_db_gpu(8500, 7682, "GTR", dg_k=4, seed=5, device="cuda:0", log=lambda *a: None)
    ins = synth_gpu.simulate_reads_gpu(db, up, down, 12, 100000, seed=3, amplicon_start=1000, amplicon_cols=3300, jitter=20, device="cuda:0")
    fw, mt = zip(*[synth.split_pair(r, 150) for r in ins])
    vf = np.stack([synth.read_vpaths(db.hmm, r) for r in fw]); vr = np.stack([synth.read_vpaths(db.hmm, r) for r in mt])
    lo, hi = 900, 4500
    up_h = up[:, lo:hi].contiguous().cpu().numpy(); down_h = down[:, lo:hi].contiguous().cpu().numpy()
    md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r)
    D = E.Database.from_arrays(db.hmm, db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, db.anno_id, msgs_on_device=True)
    assert D.n_nodes >= 16384
    B = E.Batch(D, 16)
    B.set_knob("trace", 1)
    B.set_reads([r.seq for r in fw], vf, [r.seq for r in mt], vr); B.assign(E.default_opts())
    recs = B.alignments(want_align=False)["recs"]; best = B.placements()
    span = recs["cs_end"] - recs["cs_start"] + 1
    assert (recs["status"] == 1).all() and span.max() > 3072, span
    m = O.Model(db.model.type_id, db.model.pi, db.model.par)
    H = O.Hmm(db.hmm.K, db.hmm.L, db.hmm.EM, db.hmm.EI, db.hmm.T, db.hmm.p2cs, 0)
    T = O.Tree(db.parent, db.blen, db.seq, up_h, down_h, db.height, m, db.dg_r, db.anno_id, win_start=lo, win_len=hi - lo)
    assert recs["cs_start"].min() - 1 >= lo and recs["cs_end"].max() <= hi
    ref = O.pipeline_batch(H, T, [r.seq for r in fw], vf, mates=[r.seq for r in mt], mvpaths=vr, threads=8, want_cands=True)
    assert np.array_equal(recs["cost"], ref["cost"]) and (best["n_cand"] == ref["n_cand"]).all()
    tot = classify_batch(ref, B.candidates(), best, db.parent)
    assert tot["set_differs"] == 0 and tot["swaps_unexplained"] == 0 and tot["best_unexplained"] == 0, tot
    same = best["c_node"] == ref["best_nodes"][:, 0]
    for name, col in (("ratio", 0), ("wnr", 1), ("est_loglik", 7)):
        d = np.abs(best[name][same] - ref["best_vals"][same, col]) / np.maximum(np.abs(ref["best_vals"][same, col]), 1e-3)
        assert d.max() < REL, (name, d.max())
    print("large-tree streaming kernels:", tot, "widest region", int(span.max()))
    B.close(); D.close()


@pytest.mark.parametrize("model,dg_k", [("GTR", 4), ("TN93", 0)])
def test_fix_root_loglik_flag(model, dg_k):
    """hu_opts.fix_root_loglik / --fix-root-loglik (a documented deviation, off by default): candidates ranked by the intended root
    log-likelihood (k_root_loglik) against the oracle's fixRootLoglik; without the flag root_loglik is NaN and loglik the F4 constant."""
    E = _engine()
    from oracle import oracle_py as O
    db = get_db(150, 700, model, dg_k=dg_k)
    _, H, T = oracle_objects(db)
    reads, vps = sim_reads(db, 16, 150)
    opts = E.default_opts(fix_root_loglik=1)
    D, B = _run_stages(E, db, reads, vps, opts)
    B.get_seed(opts); B.estimate_seq(opts); B.filter_placements(opts); B.place_seq(opts); B.calc_q_values(opts)
    cd, st, en = B.codes(); best = B.placements(); coffs, cpl = B.candidate_places()
    oo = O.default_opts(fixRootLoglik=1)
    differ = 0
    for i in range(len(reads)):
        res = T.assign(cd[i], int(st[i]), int(en[i]), oo)
        oc = {int(n_[0]): v for n_, v in zip(res["nodes"], res["vals"])}
        g = cpl[coffs[i]:coffs[i + 1]]
        assert sorted(oc) == sorted(int(x) for x in g["c_node"])
        for rec in g:
            v = oc[int(rec["c_node"])]
            assert _rel(rec["loglik"], v[2]) < 1e-9 and rec["root_loglik"] == rec["loglik"]
            assert abs(rec["q_place"] - v[4]) <= 1e-5 * max(1.0, abs(v[4]))      # q = -10 log10(1 - p): p good to ~1e-7 from logliks good to 1e-9 of ~1e3
        if int(best[i]["c_node"]) != int(res["nodes"][0][0]):                     # only when the oracle's own two best keys tie
            differ += 1
            assert abs(res["vals"][0][4] - res["vals"][1][4]) <= 1e-6 * max(1.0, abs(res["vals"][0][4])), (i, res["vals"][:2])
        else:
            assert _rel(best[i]["loglik"], res["vals"][0][2]) < 1e-9
    opts0 = E.default_opts()
    B.place_seq(opts0); B.calc_q_values(opts0)
    b0 = B.placements()
    assert np.isnan(b0["root_loglik"]).all()
    const = (en - st + 1) * np.log((db.model.pi * np.e).sum())
    assert _rel(b0["loglik"], const).max() < 1e-12
    # the chimera check becomes informative: log-odds are no longer identically 0, and agree with the oracle's
    W = E.Batch(D, len(reads))
    B.set_aligned(cd, st, en); B.get_seed(opts)
    cnt, ids, _, _ = B.seeds()
    res = B.check_chimera(W, opts, num_seg=2)
    lods = []
    for i in range(len(reads)):
        o = T.chimera(cd[i], int(st[i]), int(en[i]), oo, num_seg=2, seeds=ids[i, :cnt[i]])
        assert bool(res[i]["checked"]) == o["checked"]
        if (int(res[i]["seg5"]["c_node"]), int(res[i]["seg3"]["c_node"])) == (o["seg5"]["c"], o["seg3"]["c"]):
            assert abs(res[i]["lod"] - o["lod"]) <= 1e-6 * max(1.0, abs(o["lod"])), (i, res[i]["lod"], o["lod"])
            lods.append(o["lod"])
    assert len(lods) >= len(reads) // 2 and max(abs(x) for x in lods) > 1e-3
    print("fix-root-loglik: picks differing by an exact key tie:", differ, "; chimera log-odds compared:", len(lods))
    W.close(); B.close(); D.close()


def test_build_database_on_device_and_write_it(tmp_path):
    """the reduced hmmufotu-build path end to end on the device side: leaf rows -> hu_tree_evaluate (messages stay in HBM)
    -> hu_ptu_write streaming them to a .ptu -> hu_db_load streaming them back: same file as the host-array route, same placements"""
    E = _engine()
    import filecmp
    import torch
    from hmmufotu_amd import synth
    db = get_db(60, 400, "GTR", dg_k=4, seed=3)
    n, L = db.seq.shape
    md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r)
    up = torch.zeros((n, L, 4), dtype=torch.float64, device="cuda:0"); down = torch.zeros_like(up)
    leaf_only = np.where(db.is_leaf[:, None], db.seq, 0).astype(np.int8)
    seq, h = E.tree_evaluate(db.parent, db.blen, leaf_only, md, up.data_ptr(), down.data_ptr())
    torch.cuda.synchronize()
    kw = dict(names=db.names, annos=db.annos, anno_dist=db.anno_dist, model_text=db.model.text, dg_alpha=db.dg_alpha, dg_breaks=db.dg_b)
    a, b = str(tmp_path / "dev.ptu"), str(tmp_path / "host.ptu")
    E.write_ptu(a, db.parent, db.blen, seq, up.data_ptr(), down.data_ptr(), h, md, msgs_on_device=True, **kw)
    E.write_ptu(b, db.parent, db.blen, seq, up.cpu().numpy(), down.cpu().numpy(), h, md, **kw)
    assert filecmp.cmp(a, b, shallow=False)
    hp = str(tmp_path / "t.hmm"); synth.write_hmm(db.hmm, hp)
    D1 = E.Database.load(hp, a); D2 = E.Database.from_synth(db)
    reads, vps = sim_reads(db, 8, 100)
    outs = []
    for D in (D1, D2):
        B = E.Batch(D, 8); B.set_reads([r.seq for r in reads], vps); B.assign(E.default_opts()); outs.append(B.placements().copy()); B.close()
    for k in ("c_node", "a_node", "n_cand"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert np.allclose(outs[0]["ratio"], outs[1]["ratio"], rtol=1e-9, atol=1e-12)      # device-evaluated vs numpy-evaluated messages: 1e-9 apart
    D1.close(); D2.close()


def test_align_with_misplaced_and_overlapping_seeds():
    """Seed paths as a real index returns them on repeats and low-complexity reads: shifted off the true diagonal, with wrong insert / delete
    counts, the 3' seed upstream of the 5' seed, the two seeds overlapping or identical, seeds touching the first / last profile column or
    read base.  Any path with 0 < start <= end <= K and 0 < from <= to <= L is accepted by the reference (src/BandedHMMP7.cpp:773-892): the
    band geometry, the phases rewriting each other's cells and the fall back to the full DP must come out as in the oracle, bit for bit."""
    E = _engine()
    db = get_db(120, 700, "GTR", dg_k=4)
    _, H, _ = oracle_objects(db)
    K = db.hmm.K
    reads, vps0 = sim_reads(db, 64, 150)
    rng = np.random.default_rng(23)
    seqs, vps = [], []
    for i, r in enumerate(reads):
        L = len(r.seq)
        v = vps0[i].copy()
        kind = i % 8
        for p in range(2):
            if v[p, 0] == 0:
                continue
            if kind == 0:                      # shifted along the profile: off the true diagonal by up to 25 columns
                sh = int(rng.integers(-25, 26)); v[p, 0] += sh; v[p, 1] += sh
            elif kind == 1:                    # shifted along the read
                sh = int(rng.integers(-10, 11)); v[p, 2] += sh; v[p, 3] += sh
            elif kind == 2:                    # wrong gap counts: a band much wider / narrower than the path
                v[p, 4] = int(rng.integers(0, 30)); v[p, 5] = int(rng.integers(0, 30))
            elif kind == 3:                    # a longer path on the profile than on the read and the other way round
                v[p, 1] += int(rng.integers(0, 15)); v[p, 3] += int(rng.integers(0, 6))
        if kind == 4 and v[1, 0] > 0:          # 3' seed first
            v = v[::-1].copy()
        if kind == 5 and v[1, 0] > 0:          # overlapping seeds: the second starts inside the first
            v[1] = v[0]; v[1, 0] += 5; v[1, 1] += 5; v[1, 2] += 5; v[1, 3] += 5
        if kind == 6:                          # at the edges of the profile and of the read
            v[0] = [1, 20, 1, 20, 0, 0]; v[1] = [K - 19, K, L - 19, L, 0, 0]
        if kind == 7 and v[1, 0] > 0:          # the same seed twice
            v[1] = v[0]
        for p in range(2):                     # keep what the reference would accept; anything else means "no seed"
            s, e, f, t, ni, nd = (int(x) for x in v[p])
            if not (0 < s <= e <= K and 0 < f <= t <= L and ni >= 0 and nd >= 0):
                v[p] = 0
        if v[0, 0] == 0 and v[1, 0] != 0:
            v[0] = v[1]; v[1] = 0
        seqs.append(r.seq); vps.append(v)
    _check_alignments(E, db, H, seqs, np.stack(vps))


@pytest.mark.gpu
def test_width_split_of_a_batch(capfd):
    """A few reads with alignment regions of ~1,500 columns among 400 reads of ~300: the estimate and placement kernels take their shape from the widest
    region of a launch, so those reads get launches of their own (plan_width_split: HuDbDev::rLo / rHi / wideList) and the rest keep the narrow kernels —
    same candidates, same iteration counts and the same numbers as the one-launch form (knob width_split = 0), and parity with the oracle as ever."""
    E = _engine()
    from hmmufotu_amd import synth
    from oracle import oracle_py as O
    db = get_db(80, 2800, "GTR", dg_k=4)
    _, H, T = oracle_objects(db)
    rng = np.random.default_rng(33)
    narrow = synth.simulate_reads(db, 400, 150, rng, amplicon_start=300, amplicon_cols=330, jitter=20)
    wide = synth.simulate_reads(db, 3, 100000, rng, amplicon_start=100, amplicon_cols=1500, jitter=20)
    sims = list(narrow)
    for k, w in zip((17, 211, 399), wide):
        sims[k] = w
    reads = [r.seq for r in sims]
    vps = np.stack([synth.read_vpaths(db.hmm, r) for r in sims])
    opts = E.default_opts()
    D = E.Database.from_synth(db); B = E.Batch(D, 512)
    B.set_knob("trace", 1)
    B.set_reads(reads, vps); B.assign(opts)
    err = capfd.readouterr().err
    assert "width split: 3 of 400 reads beyond" in err, err[-600:]
    recs = B.alignments(want_align=False)["recs"]
    span = recs["cs_end"] - recs["cs_start"] + 1
    assert (span > 1024).sum() == 3 and np.sort(span)[-4] <= 512, np.sort(span)[-6:]
    best = B.placements().copy(); c1 = {k: v.copy() for k, v in B.candidates().items()}
    er, ew, el = [x.copy() for x in B.estimates()]
    ref = O.pipeline_batch(H, T, reads, vps, threads=4, want_cands=True)
    assert np.array_equal(recs["cost"], ref["cost"]) and (best["n_cand"] == ref["n_cand"]).all()
    tot = classify_batch(ref, c1, best, db.parent)
    assert tot["set_differs"] == 0 and tot["swaps_unexplained"] == 0 and tot["best_unexplained"] == 0, tot
    for k in (17, 211, 399):                                       # the wide reads are placed like any other
        assert best["c_node"][k] >= 0 and best["n_cand"][k] == ref["n_cand"][k]
    # one launch for everything, shaped by the widest region: the same results
    B.set_knob("width_split", 0); B.assign(opts)
    assert "width split" not in capfd.readouterr().err
    c0 = B.candidates(); b0 = B.placements()
    er0, ew0, el0 = B.estimates()
    assert np.array_equal(er, er0, equal_nan=True) and np.array_equal(ew, ew0, equal_nan=True) and np.nanmax(np.abs(el - el0) / np.maximum(np.abs(el0), 1.0)) < 1e-12
    assert np.array_equal(c1["offs"], c0["offs"]) and np.array_equal(c1["c_node"], c0["c_node"]) and np.array_equal(c1["iters"], c0["iters"])
    assert np.abs(c1["ratio"] - c0["ratio"]).max() < 1e-11 and np.abs(c1["wnr"] - c0["wnr"]).max() < 1e-11
    assert np.array_equal(best["c_node"], b0["c_node"])
    B.close(); D.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model,pi,prior", [("GTR", None, 0), ("JC69", None, 0), ("HKY85", (0.3, 0.2, 0.3, 0.2), 1)])
def test_filter_and_final_sort_fast_paths(model, pi, prior):
    """filterPlacements on a wave per read (place = number of greater keys while a read's keys are pairwise different) and bestPlace without the
    sort (a unique maximum; the all-equal table of SURVEY.md F4) against the same stages with the restated std::sort forced for every read
    (knob sort_seq): identical candidate lists, order and best records — on databases with exact ties among the estimated logliks too
    (JC69 / equal base frequencies: the two writings of an attachment at a node), with a uniform and a non-uniform prior."""
    E = _engine()
    kw = dict(pi=pi) if pi is not None else {}
    db = get_db(90, 700, model, dg_k=0 if model == "JC69" else 4, seed=9, **kw)
    reads, vps = sim_reads(db, 48, 150)
    opts = E.default_opts(max_nseed=50, prior=prior)
    D, B = _run_stages(E, db, reads, vps, opts)
    B.get_seed(opts); B.estimate_seq(opts)
    out = []
    for seq in (0, 1):
        B.set_knob("sort_seq", seq)
        B.filter_placements(opts); B.place_seq(opts); B.calc_q_values(opts)
        c = B.candidates(); offs, recs = B.candidate_places()
        out.append(({k: np.array(v, copy=True) for k, v in c.items()}, recs.copy(), B.placements().copy()))
    (fast, rf, bf), (slow, rs, bs) = out
    assert np.array_equal(fast["offs"], slow["offs"]) and fast["offs"][-1] > len(reads)
    for k in ("c_node", "iters"):
        assert np.array_equal(fast[k], slow[k]), k
    for k in ("ratio", "wnr", "est_loglik"):
        assert np.array_equal(fast[k], slow[k], equal_nan=True), k
    assert rf.tobytes() == rs.tobytes()             # every candidate's record: nodes, q-values, heights
    assert bf.tobytes() == bs.tobytes()             # bestPlace of every read
    B.close(); D.close()
