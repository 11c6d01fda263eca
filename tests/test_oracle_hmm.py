"""Pins of the oracle's profile-HMM alignment (CPU): an independent dense Viterbi written from
the recurrences, banded == full when the path lies in the band, profile post-load chain."""
import numpy as np
import pytest

from conftest import get_db, oracle_objects, sim_reads
from hmmufotu_amd import synth
from oracle import oracle_py as O

INF = float("inf")
CODE = {c: i for i, c in enumerate("ACGT")}


def dense_viterbi_cost(h, entry, exit_, tsp, read):
    """min over all cells of S, by a plain O(N K) numpy DP over the whole matrix (row-major order,
    i.e. a different evaluation order from both the reference and the oracle)."""
    K, N = h.K, len(read)
    tNN, tNB, tEC, tCC = tsp
    M = np.full((N + 1, K + 1), INF); I = np.full((N + 1, K + 1), INF); Dm = np.full((N + 1, K + 1), INF)
    for i in range(1, N + 1):
        M[i, 0] = (0 if i == 1 else tNN * (i - 1)) + tNB
        I[i, 0] = M[i, 0]
    T = h.T
    for i in range(1, N + 1):
        b = CODE[read[i - 1]]
        for j in range(1, K + 1):
            M[i, j] = h.EM[j, b] + min(M[i, 0] + entry[j], M[i - 1, j - 1] + T[j - 1, 0], I[i - 1, j - 1] + T[j - 1, 3], Dm[i - 1, j - 1] + T[j - 1, 5])
            I[i, j] = h.EI[j, b] + min(M[i - 1, j] + T[j, 1], I[i - 1, j] + T[j, 4])
            if 1 < j < K:
                Dm[i, j] = min(M[i, j - 1] + T[j - 1, 2], Dm[i, j - 1] + T[j - 1, 6])
    S = M + exit_[None, :] + tEC
    SK = I[:, K] + T[K, 3] + tEC
    for i in range(1, N):
        S[i] += tCC * (N - i); SK[i] += tCC * (N - i)
    return min(S[1:, 1:].min(), SK[1:].min())


@pytest.mark.parametrize("mode", [0, 2])
def test_full_viterbi_against_independent_dp(mode):
    db = get_db(40, 260, "GTR", dg_k=0, seed=5)
    _, H, _ = oracle_objects(db, mode)
    entry, exit_, tsp = H.params()
    reads, _ = sim_reads(db, 6, 60, amplicon=False)
    for r in reads:
        a = H.align(r.seq, None)
        assert a["ok"] and a["usedFull"]
        ref = dense_viterbi_cost(db.hmm, entry, exit_, tsp, r.seq)
        assert a["cost"] == ref                       # same additions/minima => bit-exact


def test_banded_equals_full_when_path_in_band():
    db = get_db(120, 700, "GTR", dg_k=4)
    _, H, _ = oracle_objects(db)
    reads, vps = sim_reads(db, 40, 150)
    same = 0
    for r, vp in zip(reads, vps):
        a, f = H.align(r.seq, vp), H.align(r.seq, None)
        assert a["ok"] and f["ok"] and not a["usedFull"]
        assert a["cost"] >= f["cost"] - 1e-9          # the band restricts the search space
        same += a["align"] == f["align"]
        ds = O.digitize(a["align"])
        assert len(ds) == db.cs_len
        codes = np.array([CODE[c] for c in r.seq], np.int8)
        assert (ds[r.cols] == codes).mean() > 0.9    # recovers the simulated alignment
    assert same >= 36


def test_profile_postload_chain():
    """adjustProfileLocalMode + wingRetract: entry/exit probabilities are B->Mj + the folded
    delete chain, clamped at 1 (src/BandedHMMP7.cpp:721-733,1083-1120)."""
    db = get_db(40, 260, "GTR", dg_k=0, seed=5)
    h = db.hmm
    _, H, _ = oracle_objects(db)
    entry, exit_, tsp = H.params()
    T = h.T
    assert np.isinf(entry[0]) and np.isinf(exit_[0])
    assert entry[1] == -np.log(np.exp(-T[0, 0]))
    j = 5
    chain = T[0, 2] + sum(T[i, 6] for i in range(1, j - 1)) + T[j - 1, 5]
    assert abs(entry[j] + np.log(min(1.0, np.exp(-T[0, 0]) + np.exp(-chain)))) < 1e-12
    i = h.K - 4
    chain = T[i, 2] + sum(T[k, 6] for k in range(i + 1, h.K)) + T[h.K, 5]
    assert abs(exit_[i] + np.log(min(1.0, np.exp(-T[h.K, 0]) + np.exp(-chain)))) < 1e-12
    assert tsp[0] == INF and tsp[1] == 0 and tsp[2] == 0 and tsp[3] == INF       # GLOBAL
    H.set_mode(2)
    tsp = H.params()[2]
    K = max(h.K, 350)
    assert tsp[0] == INF and abs(tsp[3] + np.log(1 - K / (K + 1.0))) < 1e-12    # NGCL: T_CC = 1 - p1


def test_build_align_path_matches_generator():
    db = get_db(120, 700, "GTR", dg_k=4)
    _, H, _ = oracle_objects(db)
    reads, _ = sim_reads(db, 10, 150)
    cs2p = synth.cs2profile(db.hmm)
    for r in reads:
        for sf in (0, 7, len(r.seq) - 20):
            c0, c1 = int(r.cols[sf]), int(r.cols[sf + 19])
            s = ["-"] * (c1 - c0 + 1)
            for k in range(20):
                s[int(r.cols[sf + k]) - c0] = r.seq[sf + k]
            v = H.build_align_path(c0 + 1, c1 + 1, "".join(s), sf + 1, sf + 20)
            assert list(v) == synth.seed_vpath(db.hmm, cs2p, r, sf)


def test_padding_quirks_and_alphabet():
    # degenerate bases -> first expansion; gaps -> -2 (src/IUPACNucl.cpp:33-50)
    d = O.digitize("ACGTUMRWSYKVHDBN-._acgtn")
    assert list(d) == [0, 1, 2, 3, 3, 0, 0, 0, 1, 1, 2, 0, 0, 0, 1, 0, -2, -2, -2, 0, 1, 2, 3, 0]
    assert len(O.digitize("AC?GT")) == 4            # invalid characters are dropped
