"""BASELINE.json configs 4 and 5 at their own READ shape on reduced trees: real 2 x 250 and 2 x 300 mates from 460- / 552-base inserts over the
full 7,682-column consensus (merged regions of ~2,500 / ~3,000 columns), through the C ABI against the oracle's per-read task, in both seed orders.
What these shapes switch on and the 110-150-base mates of the other PE tests do not: more than 255 bases per merged read (16-bit distances in the
distance-only scan, 32-bit (d, N) pairs in the pair scan and k_seed_refsort<uint32_t>), the 512-thread estimate kernel and the 4-wave placement
kernel of regions of 2,049 .. 3,072 columns, the streaming estimate / placement kernels beyond 3,072 columns, and the width split of a batch
whose few widest regions fall into the next class."""
import re

import numpy as np
import pytest

from conftest import get_db, oracle_objects

pytestmark = pytest.mark.gpu
REL = 1e-6


def _pairs(db, n, read_len, seed, n_wide=0):
    """n_wide of the n inserts come from an amplicon 130 columns wider (at 2 x 300: merged regions of ~3,160 columns, beyond the 3,072 the
    register-resident kernels hold)"""
    from hmmufotu_amd import synth
    rng = np.random.default_rng(seed)
    ins_len = int(round(read_len * 1.84))                       # bench.py: 460-base inserts for 2 x 250, 552 for 2 x 300
    amp_cols = int(round(ins_len * db.cs_len / 1400.0))
    ins = synth.simulate_reads(db, n - n_wide, 100000, rng, amplicon_start=1000, amplicon_cols=amp_cols, jitter=20)
    if n_wide:
        wide = synth.simulate_reads(db, n_wide, 100000, rng, amplicon_start=960, amplicon_cols=amp_cols + 130, jitter=5)
        at = len(ins) // 3
        ins = ins[:at] + wide + ins[at:]
    fw, mt = zip(*[synth.split_pair(r, read_len) for r in ins])
    vf = np.stack([synth.read_vpaths(db.hmm, r) for r in fw]); vr = np.stack([synth.read_vpaths(db.hmm, r) for r in mt])
    return [r.seq for r in fw], vf, [r.seq for r in mt], vr


@pytest.mark.parametrize("read_len,n_pairs,order,n_wide", [(250, 272, 1, 0), (250, 272, 0, 0), (300, 272, 1, 2), (300, 272, 0, 40)])
def test_paired_reads_of_configs_4_and_5(read_len, n_pairs, order, n_wide, capfd):
    from hmmufotu_amd import engine as E
    from oracle import oracle_py as O, parity
    if E.device_count() < 1:
        pytest.fail("no gfx950 device: GPU tests must run on the MI355X box (no CPU fallback exists)")
    db = get_db(400, 7682, "GTR", dg_k=4, n_match=1400)        # 799 nodes x the whole consensus width of the gg_97 / SILVA-scale databases
    _, H, T = oracle_objects(db)
    fw, vf, mt, vr = _pairs(db, n_pairs, read_len, seed=40 + read_len, n_wide=n_wide)
    opts = E.default_opts(seed_order=order)
    D = E.Database.from_synth(db); B = E.Batch(D, n_pairs)
    B.set_knob("trace", 1)
    capfd.readouterr()
    B.set_reads(fw, vf, mt, vr); B.assign(opts)
    err = capfd.readouterr().err
    recs = B.alignments(want_align=False)["recs"]; best = B.placements(); cand = B.candidates(); _, cpl = B.candidate_places()
    ref = O.pipeline_batch(H, T, fw, vf, mates=mt, mvpaths=vr, opts=O.default_opts(tieMode=order), threads=16, want_cands=True)
    # the shapes this test is about
    ok = recs["status"] == 1
    span = (recs["cs_end"] - recs["cs_start"] + 1)[ok]
    cd, st, en = B.codes()
    bases = np.array([(cd[i, st[i]:en[i] + 1] >= 0).sum() for i in np.nonzero(ok)[0]])
    assert ok.sum() >= n_pairs - 8 and bases.max() > 255 and span.min() > 2048, (int(ok.sum()), int(bases.max()), int(span.min()))
    if order == 1:
        assert re.search(r"k_seed_refsort: %d reads.* 32-bit pairs.* 0 reads left to the host" % n_pairs, err), err[-2000:]
    if n_wide == 2:          # two regions beyond 3,072 columns beside a main class below: they take a launch of their own on the streaming kernels
        assert span.max() > 3072 and re.search(r"width split: 2 of %d reads beyond 3072 columns" % n_pairs, err), (int(span.max()), err[-2000:])
    elif n_wide:             # many of them: the whole batch streams (config 5's own path when the amplicon is that wide)
        assert (span > 3072).sum() >= n_wide - 2 and "width split" not in err
    else:
        assert span.max() <= 3072
    # alignment: bit-exact
    assert (recs["status"] == ref["aln_ints"][:, 7]).all()
    for k, col in (("seq_start", 0), ("seq_end", 1), ("hmm_start", 2), ("hmm_end", 3), ("cs_start", 4), ("cs_end", 5)):
        assert (recs[k][ok] == ref["aln_ints"][ok, col]).all(), k
    assert np.array_equal(recs["cost"][ok], ref["cost"][ok])
    # candidates: sets, order and picks identical or an explained near-tie (oracle/parity.py); numbers at the north star's tolerance; iteration counts equal
    assert (best["n_cand"] == ref["n_cand"]).all()
    per = []; worst = dict(est=0.0, ratio=0.0, wnr=0.0, height=0.0); it_bad = ncmp = 0
    for i in range(n_pairs):
        k = int(ref["n_cand"][i]); a, b = int(cand["offs"][i]), int(cand["offs"][i + 1])
        per.append(parity.classify_read(ref["cand_node"][i, :k], ref["cand_est"][i, :k], ref["cand_ratio0"][i, :k], cand["c_node"][a:b], db.parent,
                                        pos=int(ref["best_pos"][i]) if k else None))
        pos = {int(nn): j for j, nn in enumerate(ref["cand_node"][i, :k])}
        for c in range(a, b):
            j = pos[int(cand["c_node"][c])]
            ncmp += 1
            for nm, g, o_ in (("est", cand["est_loglik"][c], ref["cand_est"][i, j]), ("ratio", cand["ratio"][c], ref["cand_placed"][i, j, 0]),
                              ("wnr", cand["wnr"][c], ref["cand_placed"][i, j, 1]), ("height", cpl["height"][c], ref["cand_placed"][i, j, 2])):
                assert np.isnan(g) == np.isnan(o_)
                if not np.isnan(g):
                    worst[nm] = max(worst[nm], abs(g - o_) / max(abs(o_), 1e-3))
            it_bad += int((int(cand["iters"][c]) & 255) != int(ref["cand_iters"][i, j, 0])) + int((int(cand["iters"][c]) >> 8) != int(ref["cand_iters"][i, j, 1]))
    tot = parity.summarize(per)
    assert tot["set_differs"] == 0 and tot["swaps_unexplained"] == 0 and tot["best_unexplained"] == 0, tot
    same = best["c_node"] == ref["best_nodes"][:, 0]
    assert same.sum() == n_pairs - tot["best_differs"]
    assert max(worst.values()) < REL and it_bad == 0 and ncmp > n_pairs, (worst, it_bad, ncmp)
    print("config %d shape, 2 x %d, seed order %s: %d pairs, %d candidates, %s, worst %s" % (4 if read_len == 250 else 5, read_len, "reference" if order else "(dist, id)", n_pairs, ncmp, tot, worst))
    B.close(); D.close()
