"""Synthetic HmmUFOtu databases and reads (data tooling, numpy only).

No gg_97_otus database ships with the reference, so every configuration of
BASELINE.json is run on a synthetic DB built here: random binary tree, sequences evolved under
the reference's trained substitution-model parameters (tests/golden/ref_data/*.sm, copied from
the reference's data/ directory), messages for every directed edge by a two-pass pruning, the
profile HMM written from column statistics, and reads drawn with the recipe of
src/hmmufotu-sim.cpp:351-424.  Writers emit the reference's on-disk formats (.hmm text,
.ptu binary) so that the C++ readers of the engine are exercised.

This module is NOT the oracle and is not on the product's compute path.
"""
from __future__ import annotations

import gzip
import os
import struct
from dataclasses import dataclass, field

import numpy as np

REF_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_data")

MODEL_TYPES = {"GTR": 0, "TN93": 1, "HKY85": 2, "F81": 3, "K80": 4, "JC69": 5}
MIN_LOGLIK_EXP = -510.0
PROG_NAME = b"HmmUFOtu"
PROG_VERSION = (1, 5, 1)


# ----------------------------------------------------------------------------- models
@dataclass
class SubModel:
    name: str
    pi: np.ndarray
    par: np.ndarray  # layout of the C ABI (include/hmmufotu_amd.h: hu_model_desc.par)
    text: str        # the model block exactly as stored inside a .ptu

    @property
    def type_id(self) -> int:
        return MODEL_TYPES[self.name]


def parse_sm(text: str) -> SubModel:
    """Parse a reference substitution-model text block (src/GTR.cpp:43-81 and friends)."""
    toks = text.split()
    name = None
    pi = np.full(4, 0.25)
    vals = {}
    R = None
    i = 0
    while i < len(toks):
        t = toks[i]
        if t.startswith("#"):
            # comment line: skip to the token after 'Model' if present
            while i < len(toks) and toks[i] != "Type:":
                i += 1
            continue
        if t == "Type:":
            name = toks[i + 1]; i += 2
        elif t == "pi:":
            pi = np.array([float(x) for x in toks[i + 1:i + 5]]); i += 5
        elif t == "R:":
            R = np.array([float(x) for x in toks[i + 1:i + 17]]).reshape(4, 4); i += 17
        elif t == "Q:":
            vals["Q"] = np.array([float(x) for x in toks[i + 1:i + 17]]).reshape(4, 4); i += 17
        elif t.endswith(":"):
            vals[t[:-1]] = float(toks[i + 1]); i += 2
        else:
            i += 1
    if name == "GTR":
        par = R.reshape(-1)
    elif name == "TN93":
        par = np.array([vals["kr"], vals["ky"], vals["beta"]])
    elif name == "HKY85":
        par = np.array([vals["kappa"], vals["beta"]])
    elif name == "F81":
        par = np.array([vals["beta"]])
    elif name == "K80":
        par = np.array([vals["kappa"]])
        pi = np.full(4, 0.25)
    elif name == "JC69":
        par = np.zeros(0)
        pi = np.full(4, 0.25)
    else:
        raise ValueError("unknown model type %r" % name)
    m = SubModel(name, pi.astype(np.float64), par.astype(np.float64), text)
    if "Q" in vals:
        m.Q_file = vals["Q"]
    return m


def load_model(name: str) -> SubModel:
    with open(os.path.join(REF_DATA, "gg_97_otus_%s.sm" % name)) as f:
        return parse_sm(f.read())


def model_P(m: SubModel, t) -> np.ndarray:
    """P(t) for an array of times -> [..., 4, 4] (row-stochastic).  Generator-side
    implementation (numpy); the engine and the oracle have their own."""
    t = np.asarray(t, dtype=np.float64)
    shp = t.shape
    t = t.reshape(-1)
    a, c, g, tt = m.pi
    P = np.zeros((t.size, 4, 4))
    if m.name == "GTR":
        R = m.par.reshape(4, 4)
        Q = R * m.pi[None, :]
        np.fill_diagonal(Q, 0.0)
        np.fill_diagonal(Q, -Q.sum(1))
        Q = Q / -np.trace(Q)
        sq = np.sqrt(m.pi)
        S = sq[:, None] * Q / sq[None, :]
        S = 0.5 * (S + S.T)
        lam, V = np.linalg.eigh(S)
        U = V / sq[:, None]
        U1 = V.T * sq[None, :]
        e = np.exp(lam[None, :] * t[:, None])
        P = np.einsum("ik,tk,kj->tij", U, e, U1)
        P[t == 0] = np.eye(4)
    elif m.name in ("TN93", "HKY85"):
        if m.name == "TN93":
            kr, ky, beta = m.par
        else:
            kr = ky = m.par[0]; beta = m.par[1]
        e = np.exp(-beta * t)
        eR = np.exp(-(1 + (a + g) * (kr - 1)) * beta * t)
        eY = np.exp(-(1 + (c + tt) * (ky - 1)) * beta * t)
        P[:, 0, 0] = (a * (a + g + (c + tt) * e) + g * eR) / (a + g)
        P[:, 0, 1] = c * (1 - e)
        P[:, 0, 2] = (g * (a + g + (c + tt) * e) - g * eR) / (a + g)
        P[:, 0, 3] = tt * (1 - e)
        P[:, 1, 0] = a * (1 - e)
        P[:, 1, 1] = (c * (c + tt + (a + g) * e) + tt * eY) / (c + tt)
        P[:, 1, 2] = g * (1 - e)
        P[:, 1, 3] = (tt * (c + tt + (a + g) * e) - tt * eY) / (c + tt)
        P[:, 2, 0] = (a * (a + g + (c + tt) * e) - a * eR) / (a + g)
        P[:, 2, 1] = c * (1 - e)
        P[:, 2, 2] = (g * (a + g + (c + tt) * e) + a * eR) / (a + g)
        P[:, 2, 3] = tt * (1 - e)
        P[:, 3, 0] = a * (1 - e)
        P[:, 3, 1] = (c * (c + tt + (a + g) * e) - c * eY) / (c + tt)
        P[:, 3, 2] = g * (1 - e)
        P[:, 3, 3] = (tt * (c + tt + (a + g) * e) + c * eY) / (c + tt)
        P = np.maximum(P, 0.0)
    elif m.name == "F81":
        e = np.exp(-m.par[0] * t)
        P[:] = (m.pi[None, None, :] * (1 - e)[:, None, None])
        for i in range(4):
            P[:, i, i] += e
    elif m.name == "K80":
        kappa = m.par[0]; beta = 1 / (2 * kappa)
        e = np.exp(-4 * beta * t); eV = np.exp(-2 * (1 + kappa) * beta * t)
        P[:] = ((1.0 - e) / 4)[:, None, None]
        for i in range(4):
            P[:, i, i] = (1.0 + e + 2 * eV) / 4
        for (i, j) in ((0, 2), (2, 0), (1, 3), (3, 1)):
            P[:, i, j] = (1.0 + e - 2 * eV) / 4
    else:
        off = (1 - np.exp(-4 * t / 3)) / 4
        P[:] = off[:, None, None]
        for i in range(4):
            P[:, i, i] = (1 + 3 * np.exp(-4 * t / 3)) / 4
    return P.reshape(shp + (4, 4))


def dgamma(K: int, alpha: float):
    """Discrete-Gamma breaks and rates exactly as src/DiscreteGammaModel.cpp:40-55 computes them
    (rates sum to 1, i.e. are NOT multiplied by K: SURVEY.md F6)."""
    from scipy.special import gammainc
    from scipy.stats import gamma as gdist
    b = np.empty(K + 1)
    for i in range(K):
        b[i] = gdist.ppf(i / float(K), a=alpha, scale=alpha)
    b[K] = np.inf
    r = np.empty(K)
    for i in range(K):
        lo, hi = b[i], b[i + 1]
        r[i] = (gammainc(alpha + 1, hi * alpha) - gammainc(alpha + 1, lo * alpha)) if np.isfinite(hi) \
            else 1 - gammainc(alpha + 1, lo * alpha)
    return b, r


# ----------------------------------------------------------------------------- database
@dataclass
class SynthDB:
    n_nodes: int
    cs_len: int
    parent: np.ndarray      # int32 [n], -1 for root (node 0); preorder numbering (parent < child)
    blen: np.ndarray        # float64 [n], length of the branch to the parent (0 for root)
    seq: np.ndarray         # int8 [n, cs_len]; leaves: 0..3 / -2 gaps, inner nodes: inferred (no gaps)
    up: np.ndarray          # float64 [n, cs_len, 4]  message node -> parent (root row: root loglik)
    down: np.ndarray        # float64 [n, cs_len, 4]  message parent -> node (root row unused)
    height: np.ndarray      # float64 [n]
    model: SubModel
    dg_k: int = 0
    dg_alpha: float = 0.0
    dg_b: np.ndarray = field(default_factory=lambda: np.zeros(0))
    dg_r: np.ndarray = field(default_factory=lambda: np.zeros(0))
    is_leaf: np.ndarray = None
    anno_id: np.ndarray = None
    names: list = None
    annos: list = None
    anno_dist: np.ndarray = None
    gap_wfrac: np.ndarray = None  # per-column (weighted) gap fraction of the leaf MSA
    hmm: "SynthHMM" = None

    @property
    def rates(self):
        return self.dg_r if self.dg_k > 0 else np.ones(1)


@dataclass
class SynthHMM:
    K: int
    L: int
    EM: np.ndarray      # cost [K+1, 4]
    EI: np.ndarray      # cost [K+1, 4]
    T: np.ndarray       # cost [K+1, 7]  MM MI MD IM II DM DD ; inf == '*'
    p2cs: np.ndarray    # int32 [K+1], 1-based CS column of profile column k (0 for k = 0)
    name: str = "synth"


def make_tree(n_leaves: int, rng: np.random.Generator, mean_blen: float = 0.05):
    """Random binary topology by sequential random joins; branch lengths Exp(mean) clipped to
    >= 1e-5 (BRANCH_EPS, src/PhyloTreeUnrooted.cpp:71).  Returned in preorder numbering."""
    n = 2 * n_leaves - 1
    left = np.full(n, -1, np.int64); right = np.full(n, -1, np.int64)
    active = list(range(n_leaves))
    nxt = n_leaves
    while len(active) > 1:
        i = int(rng.integers(len(active))); a = active[i]; active[i] = active[-1]; active.pop()
        j = int(rng.integers(len(active))); b = active[j]; active[j] = active[-1]; active.pop()
        left[nxt] = a; right[nxt] = b
        active.append(nxt); nxt += 1
    root = active[0]
    newid = np.full(n, -1, np.int64)
    parent = np.full(n, -1, np.int32)
    is_leaf = np.zeros(n, bool)
    stack = [(root, -1)]
    k = 0
    while stack:
        u, p = stack.pop()
        newid[u] = k; parent[k] = p
        if left[u] < 0:
            is_leaf[k] = True
        else:
            stack.append((right[u], k)); stack.append((left[u], k))
        k += 1
    blen = np.maximum(rng.exponential(mean_blen, n), 1e-5)
    blen[0] = 0.0
    return parent, blen, is_leaf


def _conv(P, msg):
    """log(P . exp(msg + scale)) - scale with the reference's scaling rule
    (src/PhyloTreeUnrooted.h:1495-1503).  P [...,4,4], msg [...,S,4] -> [...,S,4]"""
    mx = msg.max(-1, keepdims=True)
    scale = np.where(np.isfinite(mx) & (mx < MIN_LOGLIK_EXP), MIN_LOGLIK_EXP - mx, 0.0)
    e = np.exp(msg + scale)
    with np.errstate(divide="ignore"):
        return np.log(np.einsum("...ij,...sj->...si", P, e)) - scale


def _row_mean_exp(X):
    """X [K, S, 4] -> [S, 4] (src/PhyloTreeUnrooted.h:1521-1529)"""
    mx = X.max(0)
    scale = np.where(np.isfinite(mx) & (mx < MIN_LOGLIK_EXP), MIN_LOGLIK_EXP - mx, 0.0)
    with np.errstate(divide="ignore"):
        return np.log(np.exp(X + scale[None]).mean(0)) - scale


def evaluate_tree(parent, blen, leaf_seq, is_leaf, model: SubModel, rates, use_dg: bool):
    """Two-pass pruning: messages of every directed edge (what hmmufotu-build stores in the
    .ptu, src/hmmufotu-build.cpp:415-466), ancestral argmax sequences and node heights."""
    n, L = leaf_seq.shape
    K = len(rates)
    children = [[] for _ in range(n)]
    for i in range(1, n):
        children[parent[i]].append(i)
    P = model_P(model, blen[:, None] * np.asarray(rates)[None, :])  # [n, K, 4, 4]
    logpi = np.log(model.pi)
    up = np.zeros((n, L, 4)); down = np.zeros((n, L, 4))

    def combine(contribs, leaf_codes=None):
        X = np.zeros((K, L, 4))
        for (Pk, msg) in contribs:
            X += np.stack([_conv(Pk[k], msg) for k in range(K)])
        if leaf_codes is None:
            return _row_mean_exp(X) if use_dg else X[0]
        r = np.zeros((L, 4))
        leaf = np.where(leaf_codes[:, None] >= 0,
                        np.where(np.arange(4)[None, :] == leaf_codes[:, None], 0.0, -np.inf), logpi[None, :])
        return r + leaf

    for u in range(n - 1, -1, -1):
        if is_leaf[u]:
            up[u] = combine([], leaf_seq[u])
        else:
            up[u] = combine([(P[c], up[c]) for c in children[u]])
    for u in range(1, n):
        p = parent[u]
        contribs = []
        if p != 0:
            contribs.append((P[p], down[p]))
        contribs += [(P[c], up[c]) for c in children[p] if c != u]
        down[u] = combine(contribs)
    seq = leaf_seq.copy()
    for u in range(n):
        if not is_leaf[u]:
            seq[u] = up[u].argmax(-1).astype(np.int8)
    height = np.full(n, -1.0)
    for l in np.nonzero(is_leaf)[0]:
        h = 0.0; node = l
        while node >= 0:
            if height[node] < 0 or h < height[node]:
                height[node] = h
            if parent[node] >= 0:
                h += blen[node]
            node = parent[node]
    return up, down, seq, height


def evolve_sequences(parent, blen, cs_len, model: SubModel, site_rate, rng):
    """Root iid from pi; each child base ~ P(len * rate_site)[parent base, :]."""
    n = len(parent)
    seq = np.empty((n, cs_len), np.int8)
    seq[0] = rng.choice(4, size=cs_len, p=model.pi / model.pi.sum())
    urates, inv = np.unique(site_rate, return_inverse=True)
    for u in range(1, n):
        P = model_P(model, blen[u] * urates)            # [nr, 4, 4]
        cdf = np.cumsum(P, -1)
        pb = seq[parent[u]]
        x = rng.random(cs_len)
        c = cdf[inv, pb]                                  # [L, 4]
        seq[u] = np.minimum((x[:, None] > c).sum(1), 3).astype(np.int8)
    return seq


def build_hmm(leaf_seq: np.ndarray, symfrac: float = 0.5, name: str = "synth") -> SynthHMM:
    """Profile written directly from column statistics (any valid .hmm with MAP yes is accepted
    by the engine; the reference's Dirichlet-mixture training is out of scope)."""
    nseq, L = leaf_seq.shape
    nongap = (leaf_seq >= 0)
    occ = nongap.mean(0)
    match_cols = np.nonzero(occ >= symfrac)[0]
    K = len(match_cols)
    EM = np.full((K + 1, 4), np.inf); EI = np.full((K + 1, 4), np.inf); T = np.full((K + 1, 7), np.inf)
    p2cs = np.zeros(K + 1, np.int32)
    p2cs[1:] = match_cols + 1
    bg = np.array([(leaf_seq == b).sum() for b in range(4)], float) + 1.0
    bg /= bg.sum()
    EM[0] = -np.log(bg)
    EI[:] = -np.log(bg)[None, :]
    for k in range(1, K + 1):
        col = leaf_seq[:, match_cols[k - 1]]
        cnt = np.array([(col == b).sum() for b in range(4)], float) + 0.5 * bg * 4
        EM[k] = -np.log(cnt / cnt.sum())
    # transitions from leaf paths (pseudocount 1 per row split by a fixed prior)
    state = np.where(nongap[:, match_cols], 0, 2)        # M or D at each profile column
    ins_between = np.zeros((nseq, K + 1), bool)          # any base in non-profile columns after column k
    bounds = np.concatenate(([-1], match_cols, [L]))
    csum = np.concatenate((np.zeros((nseq, 1), np.int64), np.cumsum(nongap, 1)), 1)
    for k in range(K + 1):
        lo, hi = bounds[k] + 1, bounds[k + 1]
        ins_between[:, k] = (csum[:, hi] - csum[:, lo]) > 0
    prior = {0: np.array([0.96, 0.02, 0.02]), 1: np.array([0.6, 0.4]), 2: np.array([0.7, 0.3])}
    for k in range(K + 1):
        if k == 0:
            cur = np.zeros(nseq, np.int64)               # B behaves as M0
        else:
            cur = state[:, k - 1]
        nxt = state[:, k] if k < K else np.zeros(nseq, np.int64)
        ins = ins_between[:, k]
        m = cur == 0; d = cur == 2
        cMM = (m & ~ins & (nxt == 0)).sum(); cMI = (m & ins).sum(); cMD = (m & ~ins & (nxt == 2)).sum()
        cIM = (ins & (nxt == 0)).sum() + 1.0; cII = ins.sum() * 0.5 + 1.0
        cDM = (d & (nxt == 0)).sum(); cDD = (d & (nxt == 2)).sum()
        if k == K:
            pm = np.array([cMM + cMD, cMI], float) + 10 * np.array([0.98, 0.02])
            pm /= pm.sum()
            T[k, 0] = -np.log(pm[0]); T[k, 1] = -np.log(pm[1]); T[k, 2] = np.inf
            pi_ = np.array([cIM, cII]); pi_ /= pi_.sum()
            T[k, 3] = -np.log(pi_[0]); T[k, 4] = -np.log(pi_[1])
            T[k, 5] = 0.0; T[k, 6] = np.inf
            continue
        pm = np.array([cMM, cMI, cMD], float) + 10 * prior[0]; pm /= pm.sum()
        pi_ = np.array([cIM, cII]); pi_ /= pi_.sum()
        pd = np.array([cDM, cDD], float) + 10 * prior[2]; pd /= pd.sum()
        T[k, 0:3] = -np.log(pm); T[k, 3:5] = -np.log(pi_)
        if k == 0:
            T[k, 5] = 0.0; T[k, 6] = np.inf
        else:
            T[k, 5:7] = -np.log(pd)
    return SynthHMM(K, L, EM, EI, T, p2cs, name)


def make_db(n_leaves: int, cs_len: int, model_name: str = "GTR", dg_k: int = 0, dg_alpha: float = 0.5,
            n_match: int | None = None, seed: int = 97, mean_blen: float = 0.05,
            match_gap: float = 0.02, sparse_gap: float = 0.999, n_taxa: int = 8, pi=None) -> SynthDB:
    """pi: base frequencies in place of the model file's (TN93 / HKY85 / F81 only) — e.g. two equal pairs, which makes the A and G
    components of every all-gap column tie in exact arithmetic; such a database is for in-memory tests (its model text is not rewritten)."""
    rng = np.random.default_rng(seed)
    model = load_model(model_name)
    if pi is not None:
        assert model_name in ("TN93", "HKY85", "F81") and abs(sum(pi) - 1.0) < 1e-12
        model = SubModel(model.name, np.asarray(pi, np.float64), model.par, "")
    parent, blen, is_leaf = make_tree(n_leaves, rng, mean_blen)
    n = len(parent)
    if dg_k > 0:
        b, r = dgamma(dg_k, dg_alpha)
        site_rate = r[rng.integers(dg_k, size=cs_len)]
        rates = r
    else:
        b = np.zeros(0); r = np.zeros(0)
        site_rate = np.ones(cs_len)
        rates = np.ones(1)
    full = evolve_sequences(parent, blen, cs_len, model, site_rate, rng)
    if n_match is None:
        n_match = max(1, int(round(cs_len * 1400 / 7682)))
    match = np.zeros(cs_len, bool)
    match[np.sort(rng.choice(cs_len, size=n_match, replace=False))] = True
    gap_p = np.where(match, match_gap, sparse_gap)
    leaf_seq = full.copy()
    gaps = rng.random((n, cs_len)) < gap_p[None, :]
    leaf_seq[gaps] = -2
    leaf_seq[~is_leaf] = 0
    up, down, seq, height = evaluate_tree(parent, blen, leaf_seq, is_leaf, model, rates, dg_k > 0)
    # taxonomy annotation classes: clades cut at a fixed depth
    anno_id = np.zeros(n, np.int32)
    for u in range(1, n):
        anno_id[u] = anno_id[parent[u]] if u > n_taxa else u
    names = ["n%d" % i for i in range(n)]
    annos = ["k__Synth;p__clade%d" % anno_id[i] for i in range(n)]
    leaves = leaf_seq[is_leaf]
    db = SynthDB(n, cs_len, parent, blen, seq, up, down, height, model, dg_k, dg_alpha, b, r,
                 is_leaf, anno_id, names, annos, np.zeros(n), (leaves < 0).mean(0))
    db.hmm = build_hmm(leaves, name="synth%d" % n_leaves)
    return db


# ----------------------------------------------------------------------------- reads
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


@dataclass
class SimRead:
    seq: str
    cols: np.ndarray     # 0-based CS column of every read base
    node: int
    rc: float
    cs_start: int
    cs_end: int


def simulate_reads(db: SynthDB, n_reads: int, read_len: int, rng: np.random.Generator,
                   amplicon_start: int | None = None, amplicon_cols: int | None = None,
                   jitter: int = 30, mean_cols: float = 500, sd_cols: float = 30) -> list:
    """src/hmmufotu-sim.cpp:351-424: uniform non-root node, uniform branch point, CS window,
    per column gap with prob gapWFrac(j) else base ~ posterior at the branch point; the read is
    the ungapped string truncated to read_len."""
    out = []
    L = db.cs_len
    tries = 0
    while len(out) < n_reads:
        tries += 1
        if tries > 5000 * n_reads + 5000:
            raise RuntimeError("simulate_reads: cannot draw reads with >= 40 bases from this database/window")
        node = int(rng.integers(1, db.n_nodes))
        rc = float(rng.random())
        if amplicon_start is None:
            start = int(rng.integers(0, L))
            ln = int(max(10, rng.normal(mean_cols, sd_cols)))
        else:
            start = int(amplicon_start + rng.integers(-jitter, jitter + 1))
            ln = int(amplicon_cols)
        start = max(0, start)
        end = start + ln
        if end >= L:
            continue
        v = db.blen[node]
        Pu = model_P(db.model, v * rc); Pv = model_P(db.model, v * (1 - rc))
        U = db.up[node, start:end + 1]; V = db.down[node, start:end + 1]
        ll = _conv(Pu, U) + _conv(Pv, V)
        ll -= ll.max(-1, keepdims=True)
        p = np.exp(ll); p /= p.sum(-1, keepdims=True)
        isgap = rng.random(end - start + 1) <= db.gap_wfrac[start:end + 1]
        x = rng.random(end - start + 1)
        base = np.minimum((x[:, None] > np.cumsum(p, -1)).sum(1), 3)
        cols = np.nonzero(~isgap)[0]
        if len(cols) < min(40, max(8, read_len // 2)):
            continue
        cols = cols[:read_len]
        s = BASES[base[cols]].tobytes().decode()
        out.append(SimRead(s, cols + start, node, rc, start, end))
    return out


def split_pair(ins: SimRead, read_len: int):
    """A simulated insert -> (forward read, mate after reverse-complementing it back: src/hmmufotu.cpp:609): the first and
    the last read_len bases of the insert."""
    n = len(ins.seq)
    m = min(read_len, n)
    f = SimRead(ins.seq[:m], ins.cols[:m], ins.node, ins.rc, ins.cs_start, ins.cs_end)
    r = SimRead(ins.seq[n - m:], ins.cols[n - m:], ins.node, ins.rc, ins.cs_start, ins.cs_end)
    return f, r


def seed_vpath(hmm: SynthHMM, cs2p: np.ndarray, rd: SimRead, seed_from: int, seed_len: int = 20):
    """The CSLoc a CSFM hit at read[seed_from:+seed_len] would return for the read's true
    alignment, pushed through buildAlignPath (src/BandedHMMP7.cpp:894-941).  Returns the 6 ints
    (start,end,from,to,nIns,nDel); start == 0 marks an invalid path."""
    c0, c1 = int(rd.cols[seed_from]), int(rd.cols[seed_from + seed_len - 1])
    has = np.zeros(c1 - c0 + 1, bool)
    has[rd.cols[seed_from:seed_from + seed_len] - c0] = True
    start = end = frm = to = nins = ndel = 0
    i = seed_from + 1
    for off in range(c1 - c0 + 1):
        j = c0 + off + 1                       # 1-based CS column
        k = int(cs2p[j]) if j < len(cs2p) else 0
        ng = bool(has[off])
        if frm == 0 and ng:
            frm = i
        if ng:
            to = i
        if k != 0:
            if start == 0:
                start = k
            end = k
            if not ng:
                ndel += 1
        elif ng:
            nins += 1
        if ng:
            i += 1
    return [start, end, frm, to, nins, ndel]


def cs2profile(hmm: SynthHMM) -> np.ndarray:
    """cs2ProfileIdx as the .hmm loader leaves it (src/BandedHMMP7.cpp:192, :701-705)."""
    m = np.zeros(max(hmm.L, int(hmm.p2cs[hmm.K])) + 2, np.int32)
    m[hmm.p2cs[1:]] = np.arange(1, hmm.K + 1)
    m[hmm.p2cs[hmm.K] + 1: hmm.L + 1] = hmm.K
    return m


def read_vpaths(hmm: SynthHMM, rd: SimRead, seed_len: int = 20, seed_region: int = 50, global_mode: bool = True):
    """Emulates the seed scans of alignSeq (src/HmmUFOtu_main.cpp:50-84) with every k-mer 'found'
    at the read's true location.  Returns int32 [2, 6] (unused rows zero)."""
    cs2p = cs2profile(hmm)
    n = len(rd.seq)
    vp = np.zeros((2, 6), np.int32)
    k = 0
    region = min(seed_region, n)
    for sf in range(0, region - seed_len + 1):
        v = seed_vpath(hmm, cs2p, rd, sf, seed_len)
        if v[0] > 0 and v[0] <= v[1] and v[2] > 0 and v[2] <= v[3]:
            vp[k] = v; k += 1
            break
    if global_mode and (k == 0 or n >= 2 * region):
        st = n - 1
        while st - seed_len + 1 >= n - region:
            sf = st - seed_len + 1
            if sf < 0:
                break
            v = seed_vpath(hmm, cs2p, rd, sf, seed_len)
            if v[0] > 0 and v[0] <= v[1] and v[2] > 0 and v[2] <= v[3]:
                vp[k] = v; k += 1
                break
            st -= 1
    return vp


# ----------------------------------------------------------------------------- writers
def _fmt_cost(v: float) -> str:
    return "*" if not np.isfinite(v) else repr(float(v))


def write_hmm(h: SynthHMM, path: str):
    """HMMER3/f-style text as src/BandedHMMP7.cpp:324-378 writes it (full precision)."""
    with open(path, "w") as f:
        f.write("HMMER3/f\tsynth\nNAME\t%s\nLENG\t%d\nALPH\tDNA\n" % (h.name, h.K))
        f.write("MAXL  %d\nRF  no\nMM  no\nCONS  no\nCS  no\nMAP  yes\nNSEQ  1\nEFFN  1\n" % h.L)
        f.write("HMM\t\tA\tC\tG\tT\n\t\tm->m\tm->i\tm->d\ti->m\ti->i\td->m\td->d\n")
        for k in range(h.K + 1):
            if k == 0:
                f.write("\tCOMPO\t" + "\t".join(repr(float(x)) for x in h.EM[0]) + "\n")
            else:
                f.write("\t%d\t" % k + "\t".join(repr(float(x)) for x in h.EM[k]) + "\t%d\n" % h.p2cs[k])
            f.write("\t" + "".join("\t" + _fmt_cost(x) for x in h.EI[k]) + "\n")
            f.write("\t\t" + "\t".join(_fmt_cost(x) for x in h.T[k]) + "\n")
        f.write("//\n")


def _wstr(f, s: bytes):
    f.write(struct.pack("<Q", len(s))); f.write(s)


def write_ptu(db: SynthDB, path: str):
    """Binary .ptu exactly as PTUnrooted::save lays it out."""
    n, L = db.n_nodes, db.cs_len
    children = [[] for _ in range(n)]
    for i in range(1, n):
        children[db.parent[i]].append(i)
    with open(path, "wb") as f:
        f.write(PROG_NAME); f.write(struct.pack("<3i", *PROG_VERSION))
        f.write(struct.pack("<Qi", n, L))
        for i in range(n):
            f.write(struct.pack("<q", i)); _wstr(f, db.names[i].encode())
            f.write(struct.pack("<?", False)); _wstr(f, db.names[i].encode()); _wstr(f, db.seq[i].tobytes())
            _wstr(f, db.annos[i].encode()); f.write(struct.pack("<d", float(db.anno_dist[i])))
        f.write(struct.pack("<Q", 2 * (n - 1)))
        for u in range(n):
            nbrs = ([int(db.parent[u])] if u > 0 else []) + children[u]
            for v in nbrs:
                u_is_parent = (v != db.parent[u]) if u > 0 else True
                f.write(struct.pack("<qq?", u, v, u_is_parent))
                child = v if u_is_parent else u
                f.write(struct.pack("<dQ", float(db.blen[child]), 4 * L))
                msg = db.down[child] if u_is_parent else db.up[child]
                f.write(np.ascontiguousarray(msg, np.float64).tobytes())
        f.write(struct.pack("<q", 0)); f.write(np.ascontiguousarray(db.up[0], np.float64).tobytes())
        for i in range(n):
            f.write(struct.pack("<qd", i, float(db.height[i])))
        leaves = np.nonzero(db.is_leaf)[0]
        f.write(struct.pack("<I", len(leaves)))
        for k, i in enumerate(leaves):
            f.write(struct.pack("<Iq", k, int(i)))
        assert db.model.text, "a database made with pi= has no model text to write"
        f.write((db.model.name + "\n").encode()); f.write(db.model.text.encode())
        if not db.model.text.endswith("\n"):
            f.write(b"\n")
        f.write(struct.pack("<?", db.dg_k > 0))
        if db.dg_k > 0:
            f.write(struct.pack("<id", db.dg_k, db.dg_alpha))
            f.write(np.asarray(db.dg_b, np.float64).tobytes()); f.write(np.asarray(db.dg_r, np.float64).tobytes())


def revcom(s: str) -> str:
    return s[::-1].translate(str.maketrans("ACGTUYRSWKMBVDHN", "TGCAARYSWMKVBHDN"))


def load_70otus():
    """The reference's only fixture (test/70_otus.fasta + .tree), parsed here as DATA."""
    seqs = {}
    with gzip.open(os.path.join(REF_DATA, "70_otus.fasta.gz"), "rt") as f:
        name = None
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                name = line[1:].split()[0]; seqs[name] = []
            elif name is not None:
                seqs[name].append(line)
    seqs = {k: "".join(v) for k, v in seqs.items()}
    with open(os.path.join(REF_DATA, "70_otus.tree")) as f:
        nwk = f.read().strip()
    return seqs, nwk


# ----------------------------------------------------------------------------- config 1: the reference's own fixture
def parse_newick(nwk: str):
    """Newick text -> (parent, blen, names) with node ids as PTUnrooted(const NewickTree&) assigns them
    (src/PhyloTreeUnrooted.cpp:131-182): depth-first from the root with an explicit stack, children pushed in file order,
    so the LAST child is numbered first; root = 0, parents before children.  Quoted labels and inner labels / support
    values are kept as names; a missing length is 0."""
    s = nwk.strip()
    if s.endswith(";"):
        s = s[:-1]
    pos = 0

    def label():
        nonlocal pos
        if pos < len(s) and s[pos] == "'":
            e = s.index("'", pos + 1); out = s[pos + 1:e]; pos = e + 1
            return out
        st = pos
        while pos < len(s) and s[pos] not in ",():;":
            pos += 1
        return s[st:pos].strip()

    def node():
        nonlocal pos
        kids = []
        if s[pos] == "(":
            pos += 1
            while True:
                kids.append(node())
                if s[pos] == ",":
                    pos += 1; continue
                if s[pos] == ")":
                    pos += 1; break
                raise ValueError("newick: unexpected '%s' at %d" % (s[pos], pos))
        name = label()
        length = 0.0
        if pos < len(s) and s[pos] == ":":
            pos += 1
            st = pos
            while pos < len(s) and s[pos] not in ",();":
                pos += 1
            length = float(s[st:pos])
        return (name, length, kids)

    root = node()
    parent, blen, names = [], [], []
    stack = [(root, -1)]
    while stack:
        (name, length, kids), par = stack.pop()
        i = len(parent)
        parent.append(par); blen.append(length if par >= 0 else 0.0); names.append(name)
        for k in kids:
            stack.append((k, i))
    return np.asarray(parent, np.int32), np.asarray(blen, np.float64), names


_IUPAC_FIRST = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3, "M": 0, "R": 0, "W": 0, "S": 1, "Y": 1, "K": 2, "V": 0, "H": 0, "D": 0, "B": 1, "N": 0}


def make_db_70otus(model_name: str = "JC69", dg_k: int = 0, dg_alpha: float = 0.5, min_blen: float = 1e-5) -> SynthDB:
    """BASELINE.json config 1: a database from the reference's test fixture (test/70_otus.fasta + .tree + taxonomy, copied
    as data under tests/golden/ref_data) with the reduced hmmufotu-build recipe of SURVEY.md §8d: MSA pruned of its all-gap
    columns (MSA::prune, src/MSA.cpp:87), tree from the Newick file with the reference's node numbering, leaf branches of
    length <= 0 set to BRANCH_EPS (fixBranchLength, src/PhyloTreeUnrooted.cpp:289-296), messages for every directed edge by
    two-pass pruning under the trained model of data/gg_97_otus_<model>.sm, ancestors by per-site argmax, profile from the
    leaf columns (the HMM training of the reference is out of scope: any valid .hmm is accepted)."""
    seqs, nwk = load_70otus()
    parent, blen, names = parse_newick(nwk)
    n = len(parent)
    is_leaf = np.ones(n, bool); is_leaf[parent[parent >= 0]] = False
    L0 = len(next(iter(seqs.values())))
    raw = np.full((n, L0), -2, np.int8)
    for i in range(n):
        if is_leaf[i]:
            if names[i] not in seqs:
                raise ValueError("leaf %s has no sequence in the fixture" % names[i])
            a = np.frombuffer(seqs[names[i]].upper().encode(), np.uint8)
            row = np.full(L0, -2, np.int8)
            for ch, code in _IUPAC_FIRST.items():
                row[a == ord(ch)] = code
            raw[i] = row
    keep = (raw[is_leaf] >= 0).any(0)                       # MSA::prune: drop the columns that are all gaps
    leaf_seq = np.ascontiguousarray(raw[:, keep])
    cs_len = leaf_seq.shape[1]
    blen = blen.copy()
    fix = is_leaf & (parent >= 0) & (blen <= 0)
    blen[fix] = min_blen
    model = load_model(model_name)
    if dg_k > 0:
        b, r = dgamma(dg_k, dg_alpha); rates = r
    else:
        b = np.zeros(0); r = np.zeros(0); rates = np.ones(1)
    leaf_only = leaf_seq.copy(); leaf_only[~is_leaf] = 0
    up, down, seq, height = evaluate_tree(parent, blen, leaf_only, is_leaf, model, rates, dg_k > 0)
    tax = {}
    with open(os.path.join(REF_DATA, "70_otus_taxonomy.txt")) as f:
        for line in f:
            k, _, v = line.rstrip("\n").partition("\t")
            tax[k] = v
    annos = [tax.get(names[i], "") if is_leaf[i] else None for i in range(n)]
    for i in range(n - 1, -1, -1):                            # inner nodes: the common taxonomy prefix of their children
        if annos[i] is None:
            ks = [annos[c] for c in np.nonzero(parent == i)[0]]
            pre = [x.split("; ") for x in ks]
            com = []
            for parts in zip(*pre):
                if all(q == parts[0] for q in parts):
                    com.append(parts[0])
                else:
                    break
            annos[i] = "; ".join(com)
    cls = {}
    anno_id = np.array([cls.setdefault(a, len(cls)) for a in annos], np.int32)
    leaves = leaf_seq[is_leaf]
    db = SynthDB(n, cs_len, parent, blen, seq, up, down, height, model, dg_k, dg_alpha, b, r, is_leaf, anno_id, names, annos,
                 np.zeros(n), (leaves < 0).mean(0))
    db.hmm = build_hmm(leaves, name="70_otus")
    return db
