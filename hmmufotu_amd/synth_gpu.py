"""gg_97 / SILVA-scale synthetic databases built on the GPU with torch (data tooling for bench.py).

Same recipe as synth.make_db, but the sequence evolution and the two-pass message
evaluation run level-by-level on the device so that a 198,643-node x 7,682-column database (98 GB
of FP64 messages) is produced in minutes and handed to the engine without leaving HBM
(hu_tree_desc.msgs_on_device).  torch is used here only to make synthetic data; it is not on the
engine's compute path.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import synth

NEG = -510.0


def _levels(parent):
    n = len(parent)
    depth = np.zeros(n, np.int32)
    for i in range(1, n):
        depth[i] = depth[parent[i]] + 1
    order = np.argsort(depth, kind="stable")
    bounds = np.searchsorted(depth[order], np.arange(depth.max() + 2))
    return depth, [order[bounds[d]:bounds[d + 1]] for d in range(depth.max() + 1)]


def _conv(P, msg):
    """P [n,K,4,4], msg [n,S,4] -> [n,K,S,4] = log(P . exp(msg + scale)) - scale (reference scaling rule)"""
    mx = msg.max(-1, keepdim=True).values
    scale = torch.where(torch.isfinite(mx) & (mx < NEG), NEG - mx, torch.zeros_like(mx))
    e = torch.exp(msg + scale)
    out = torch.einsum("nkij,nsj->nksi", P, e)
    return torch.log(out) - scale.unsqueeze(1)


def _row_mean_exp(X):
    """X [n,K,S,4] -> [n,S,4]"""
    mx = X.max(1).values
    scale = torch.where(torch.isfinite(mx) & (mx < NEG), NEG - mx, torch.zeros_like(mx))
    return torch.log(torch.exp(X + scale.unsqueeze(1)).mean(1)) - scale


def make_db_gpu(n_leaves: int, cs_len: int, model_name="GTR", dg_k=0, dg_alpha=0.5, seed=97, mean_blen=0.05,
                n_match=None, match_gap=0.02, sparse_gap=0.999, win=None, device="cuda:0", chunk=2048, log=print, partial_frac=0.0):
    """Returns (SynthDB-like object with host arrays, up_dev, down_dev).  `win` = (start, len) keeps
    messages for a column window only (None = all columns).  partial_frac: that fraction of the leaves lose a prefix or a suffix
    of random length (partial reference sequences: all gaps there)."""
    t0 = time.time()
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=device); gen.manual_seed(seed)
    model = synth.load_model(model_name)
    parent, blen, is_leaf = synth.make_tree(n_leaves, rng, mean_blen)
    n = len(parent)
    depth, levels = _levels(parent)
    if dg_k > 0:
        b, r = synth.dgamma(dg_k, dg_alpha); rates = r
        site_cat = rng.integers(dg_k, size=cs_len)
    else:
        b = np.zeros(0); r = np.zeros(0); rates = np.ones(1)
        site_cat = np.zeros(cs_len, np.int64)
    K = len(rates)
    Pall = synth.model_P(model, blen[:, None] * rates[None, :])          # [n, K, 4, 4] host
    log("tree + P: %.1fs (n=%d, depth=%d)" % (time.time() - t0, n, depth.max()))
    # ---- evolve sequences level by level
    seq = torch.empty((n, cs_len), dtype=torch.int8, device=device)
    pi = torch.tensor(model.pi / model.pi.sum(), device=device)
    seq[0] = torch.multinomial(pi, cs_len, replacement=True, generator=gen).to(torch.int8)
    cat_d = torch.tensor(site_cat, device=device)
    for lv in levels[1:]:
        for a in range(0, len(lv), chunk * 4):
            idx = lv[a:a + chunk * 4]
            idx_d = torch.tensor(idx, device=device)
            cdf = torch.tensor(np.cumsum(Pall[idx], -1), dtype=torch.float32, device=device)   # [m,K,4,4]
            pb = seq[torch.tensor(parent[idx], device=device)].long()                            # [m,L]
            sel = cdf[:, cat_d]                                                                  # [m,L,4,4]
            row = torch.gather(sel, 2, pb[:, :, None, None].expand(-1, -1, 1, 4)).squeeze(2)      # [m,L,4]
            x = torch.rand(row.shape[:2], device=device, generator=gen)
            seq[idx_d] = (x[:, :, None] > row).sum(-1).clamp_(max=3).to(torch.int8)
    log("evolve: %.1fs" % (time.time() - t0))
    if n_match is None:
        n_match = max(1, int(round(cs_len * 1400 / 7682)))
    match = np.zeros(cs_len, bool)
    match[np.sort(rng.choice(cs_len, size=n_match, replace=False))] = True
    gap_p = torch.tensor(np.where(match, match_gap, sparse_gap), dtype=torch.float32, device=device)
    leaf_idx = np.nonzero(is_leaf)[0]
    leaf_d = torch.tensor(leaf_idx, device=device)
    for a in range(0, len(leaf_idx), chunk * 4):
        li = leaf_d[a:a + chunk * 4]
        g = torch.rand((len(li), cs_len), device=device, generator=gen) < gap_p[None, :]
        s = seq[li]; s[g] = -2; seq[li] = s
    if partial_frac > 0:
        part = leaf_idx[rng.random(len(leaf_idx)) < partial_frac]
        cut = rng.integers(cs_len // 8, cs_len - cs_len // 8, size=len(part)); head = rng.random(len(part)) < 0.5
        col = torch.arange(cs_len, device=device)[None, :]
        for a in range(0, len(part), chunk * 4):
            pi_ = torch.tensor(part[a:a + chunk * 4], device=device)
            c = torch.tensor(cut[a:a + chunk * 4], device=device)[:, None]; h = torch.tensor(head[a:a + chunk * 4], device=device)[:, None]
            s = seq[pi_]; s[torch.where(h, col < c, col >= c)] = -2; seq[pi_] = s
    # ---- messages
    w0, wl = (0, cs_len) if win is None else (int(win[0]), int(win[1]))
    up = torch.empty((n, wl, 4), dtype=torch.float64, device=device)
    down = torch.zeros((n, wl, 4), dtype=torch.float64, device=device)
    # messages, ancestral sequences and heights by the engine's own tree pre-evaluation kernels
    # (hu_tree_evaluate); tests/test_gpu_parity.py checks them against the oracle
    from . import engine as E
    md = E.model_desc(model.type_id, model.pi, model.par, r if dg_k > 0 else None)
    torch.cuda.synchronize()
    seq_h = seq.cpu().numpy()
    seq_h, height = E.tree_evaluate(parent, blen, seq_h, md, up.data_ptr(), down.data_ptr(), w0, wl if win is not None else 0,
                                    device=torch.device(device).index or 0)
    log("tree pre-evaluation (up + down): %.1fs" % (time.time() - t0))
    sub = leaf_idx[rng.choice(len(leaf_idx), size=min(len(leaf_idx), 4000), replace=False)]
    leaves = seq_h[sub]
    n_taxa = 64
    anno_id = np.zeros(n, np.int32)
    for u in range(1, n):
        anno_id[u] = anno_id[parent[u]] if u > n_taxa else u
    db = synth.SynthDB(n, cs_len, parent, blen, seq_h, None, None, height, model, dg_k, dg_alpha, b, r, is_leaf, anno_id,
                       None, None, np.zeros(n), (leaves < 0).mean(0))
    db.hmm = synth.build_hmm(leaves, name="synth%d" % n_leaves)
    db.win = (w0, wl)
    log("db ready: %.1fs" % (time.time() - t0))
    return db, up, down


def simulate_reads_gpu(db, up, down, n_reads, read_len, seed, amplicon_start, amplicon_cols, jitter=30, device="cuda:0", uniform=False):
    """Batched version of synth.simulate_reads (src/hmmufotu-sim.cpp:351-424).  uniform: window starts uniform over the
    resident columns (the reference simulator's own choice, :371) instead of one amplicon window +- jitter."""
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=device); gen.manual_seed(seed)
    w0, wl = db.win
    nodes = rng.integers(1, db.n_nodes, size=n_reads)
    rc = rng.random(n_reads)
    if uniform:
        start = rng.integers(w0, w0 + wl - amplicon_cols - 1, size=n_reads)
    else:
        start = np.clip(amplicon_start + rng.integers(-jitter, jitter + 1, size=n_reads), w0, w0 + wl - amplicon_cols - 1)
    S = amplicon_cols + 1
    v = db.blen[nodes]
    Pu = torch.tensor(synth.model_P(db.model, v * rc), device=device)[:, None]
    Pv = torch.tensor(synth.model_P(db.model, v * (1 - rc)), device=device)[:, None]
    nd = torch.tensor(nodes, device=device)
    col = torch.tensor(start - w0, device=device)[:, None] + torch.arange(S, device=device)[None, :]
    U = up[nd[:, None], col]; V = down[nd[:, None], col]
    ll = _conv(Pu, U)[:, 0] + _conv(Pv, V)[:, 0]
    ll = ll - ll.max(-1, keepdim=True).values
    p = torch.exp(ll); p = p / p.sum(-1, keepdim=True)
    x = torch.rand((n_reads, S), device=device, generator=gen, dtype=torch.float64)
    base = (x[:, :, None] > torch.cumsum(p, -1)).sum(-1).clamp_(max=3).cpu().numpy()
    gw = torch.tensor(db.gap_wfrac, device=device)
    isgap = (torch.rand((n_reads, S), device=device, generator=gen, dtype=torch.float64) <= gw[col + w0]).cpu().numpy()
    out = []
    for i in range(n_reads):
        cols = np.nonzero(~isgap[i])[0][:read_len]
        s = synth.BASES[base[i, cols]].tobytes().decode()
        out.append(synth.SimRead(s, cols + start[i], int(nodes[i]), float(rc[i]), int(start[i]), int(start[i] + S - 1)))
    return out


def simulate_pool_gpu(db, up, down, n_reads, read_len, seed, amplicon_start, amplicon_cols, jitter=30, device="cuda:0", chunk=32768):
    """n_reads DISTINCT single-end reads by the recipe of simulate_reads_gpu, returned packed — one uint8 buffer of bases and
    offsets [n + 1] — without a Python object per read (bench.py's end-to-end pool of a million reads)."""
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=device); gen.manual_seed(seed)
    w0, wl = db.win
    S = amplicon_cols + 1
    gw = torch.tensor(db.gap_wfrac, device=device)
    lut = torch.tensor(np.frombuffer(b"ACGT", np.uint8).copy(), device=device)
    parts, lens = [], []
    for a in range(0, n_reads, chunk):
        m = min(chunk, n_reads - a)
        nodes = rng.integers(1, db.n_nodes, size=m); rc = rng.random(m)
        start = np.clip(amplicon_start + rng.integers(-jitter, jitter + 1, size=m), w0, w0 + wl - amplicon_cols - 1)
        v = db.blen[nodes]
        Pu = torch.tensor(synth.model_P(db.model, v * rc), device=device)[:, None]
        Pv = torch.tensor(synth.model_P(db.model, v * (1 - rc)), device=device)[:, None]
        nd = torch.tensor(nodes, device=device)
        col = torch.tensor(start - w0, device=device)[:, None] + torch.arange(S, device=device)[None, :]
        ll = _conv(Pu, up[nd[:, None], col])[:, 0] + _conv(Pv, down[nd[:, None], col])[:, 0]
        ll = ll - ll.max(-1, keepdim=True).values
        p = torch.exp(ll); p = p / p.sum(-1, keepdim=True)
        x = torch.rand((m, S), device=device, generator=gen, dtype=torch.float64)
        base = (x[:, :, None] > torch.cumsum(p, -1)).sum(-1).clamp_(max=3)
        keep = ~(torch.rand((m, S), device=device, generator=gen, dtype=torch.float64) <= gw[col + w0])
        keep &= torch.cumsum(keep.to(torch.int32), 1) <= read_len
        parts.append(lut[base[keep]].cpu().numpy()); lens.append(keep.sum(1).cpu().numpy())
        del ll, p, x, base, keep
    lens = np.concatenate(lens).astype(np.int64)
    offs = np.zeros(n_reads + 1, np.int64); offs[1:] = np.cumsum(lens)
    return np.concatenate(parts), offs
