"""ctypes binding of the CPU oracle (liboracle.so).  TEST INFRASTRUCTURE ONLY: import this from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from hmmufotu_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.cpp", "oracle_models.h", "oracle_hmm.h", "oracle_phylo.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_model_new.restype = C.c_void_p
        L.orc_hmm_new.restype = C.c_void_p
        L.orc_tree_new.restype = C.c_void_p
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class OrcOpts(C.Structure):
    _fields_ = [("maxDiff", C.c_double), ("maxHeight", C.c_double), ("maxError", C.c_double),
                ("maxNSeed", C.c_int), ("weighted", C.c_int), ("onlyML", C.c_int), ("prior", C.c_int), ("tieMode", C.c_int),
                ("fixRootLoglik", C.c_int), ("tieTol", C.c_double)]


def default_opts(**kw):
    """the reference's defaults; tieMode = 1 (TIE_LIBSTDCXX: getSeed's literal std::sort on dist alone, src/HmmUFOtu_main.cpp:139)"""
    o = OrcOpts(float("inf"), float("inf"), 20.0, 50, 0, 0, 0, 1, 0, 0.0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Model:
    def __init__(self, type_id: int, pi, par):
        self.pi = np.ascontiguousarray(pi, np.float64)
        self.par = np.ascontiguousarray(np.concatenate([np.asarray(par, np.float64).ravel(), np.zeros(16)]))
        self.h = C.c_void_p(lib().orc_model_new(C.c_int(type_id), _p(self.pi, C.c_double), _p(self.par, C.c_double)))

    def P(self, t: float):
        out = np.empty(16)
        lib().orc_model_pr(self.h, C.c_double(t), _p(out, C.c_double))
        return out.reshape(4, 4)

    def Q(self):
        pi = np.empty(4); q = np.empty(16)
        lib().orc_model_get(self.h, _p(pi, C.c_double), _p(q, C.c_double))
        return q.reshape(4, 4)

    def __del__(self):
        try:
            lib().orc_model_free(self.h)
        except Exception:
            pass


class Hmm:
    def __init__(self, K, L, EM, EI, T, p2cs, mode=0):
        self.K, self.L = int(K), int(L)
        self.EM = np.ascontiguousarray(EM, np.float64); self.EI = np.ascontiguousarray(EI, np.float64)
        self.T = np.ascontiguousarray(T, np.float64); self.p2cs = np.ascontiguousarray(p2cs, np.int32)
        self.h = C.c_void_p(lib().orc_hmm_new(C.c_int(K), C.c_int(L), _p(self.EM, C.c_double), _p(self.EI, C.c_double),
                                              _p(self.T, C.c_double), _p(self.p2cs, C.c_int), C.c_int(mode)))

    def set_mode(self, mode):
        lib().orc_hmm_set_mode(self.h, C.c_int(mode))

    def params(self):
        e = np.empty(self.K + 1); x = np.empty(self.K + 1); t = np.empty(4)
        lib().orc_hmm_get(self.h, _p(e, C.c_double), _p(x, C.c_double), _p(t, C.c_double))
        return e, x, t

    def build_align_path(self, loc_start, loc_end, cs: str, cs_from, cs_to):
        out = np.zeros(6, np.int32)
        lib().orc_build_align_path(self.h, C.c_int(loc_start), C.c_int(loc_end), cs.encode(), C.c_int(cs_from), C.c_int(cs_to), _p(out, C.c_int))
        return out

    def align(self, read: str, vpaths=None):
        vp = np.ascontiguousarray(vpaths if vpaths is not None else np.zeros((0, 6)), np.int32).reshape(-1, 6)
        ints = np.zeros(8, np.int32); cost = C.c_double(0)
        aln = C.create_string_buffer(self.L + 1); tr = C.create_string_buffer(4 * (len(read) + self.K) + 16)
        ok = lib().orc_align(self.h, read.encode(), C.c_int(len(read)), _p(vp, C.c_int), C.c_int(len(vp)), _p(ints, C.c_int),
                             C.byref(cost), aln, tr, C.c_int(len(tr)))
        return dict(ok=bool(ok), seqStart=int(ints[0]), seqEnd=int(ints[1]), hmmStart=int(ints[2]), hmmEnd=int(ints[3]),
                    csStart=int(ints[4]), csEnd=int(ints[5]), usedFull=bool(ints[6]), cost=cost.value,
                    align=aln.raw[:self.L].decode("latin1") if ok else "", trace=tr.value.decode())

    def __del__(self):
        try:
            lib().orc_hmm_free(self.h)
        except Exception:
            pass


def digitize(align: str) -> np.ndarray:
    out = np.empty(len(align), np.int8)
    n = lib().orc_digitize(align.encode("latin1"), C.c_int(len(align)), _p(out, C.c_int8))
    return out[:n]


def merge(L, intsF, costF, alignF: bytes, intsR, costR, alignR: bytes):
    i = np.ascontiguousarray(intsF, np.int32).copy(); c = C.c_double(costF); a = C.create_string_buffer(alignF, L)
    ir = np.ascontiguousarray(intsR, np.int32); cr = C.c_double(costR)
    ok = lib().orc_merge(C.c_int(L), _p(i, C.c_int), C.byref(c), a, _p(ir, C.c_int), C.byref(cr), alignR)
    return bool(ok), i, c.value, a.raw[:L]


class Tree:
    def __init__(self, parent, blen, seq, up, down, height, model: Model, dg_r=None, anno_id=None, win_start=0, win_len=0):
        self.parent = np.ascontiguousarray(parent, np.int32); self.blen = np.ascontiguousarray(blen, np.float64)
        self.seq = np.ascontiguousarray(seq, np.int8); self.up = np.ascontiguousarray(up, np.float64)
        self.down = np.ascontiguousarray(down, np.float64); self.height = np.ascontiguousarray(height, np.float64)
        self.anno = None if anno_id is None else np.ascontiguousarray(anno_id, np.int32)
        self.model = model
        self.dgr = np.ascontiguousarray(dg_r if dg_r is not None else np.zeros(0), np.float64)
        n, L = self.seq.shape
        self.n, self.L = n, L
        self.h = C.c_void_p(lib().orc_tree_new(C.c_int(n), C.c_int(L), _p(self.parent, C.c_int), _p(self.blen, C.c_double),
                                               _p(self.seq, C.c_int8), _p(self.up, C.c_double), _p(self.down, C.c_double),
                                               _p(self.height, C.c_double), _p(self.anno, C.c_int) if self.anno is not None else None,
                                               model.h, C.c_int(len(self.dgr)), _p(self.dgr, C.c_double), C.c_long(win_start), C.c_long(win_len)))

    def set_rows(self, row_of):
        """up / down hold the message rows of some nodes only: row_of [n_nodes] int32, -1 = absent (None: row = node)"""
        self.row_of = None if row_of is None else np.ascontiguousarray(row_of, np.int32)
        lib().orc_tree_set_rows(self.h, _p(self.row_of, C.c_int) if self.row_of is not None else None)

    def get_seed(self, seq, start, end, max_diff=float("inf"), max_height=float("inf"), tie=1, max_n=50):
        seq = np.ascontiguousarray(seq, np.int8)
        ids = np.zeros(max(max_n, 1), np.int64); d = np.zeros_like(ids); N = np.zeros_like(ids); dist = np.zeros(len(ids))
        n = lib().orc_get_seed(self.h, _p(seq, C.c_int8), C.c_int(start), C.c_int(end), C.c_double(max_diff), C.c_double(max_height),
                               C.c_int(tie), C.c_int(max_n), _p(ids, C.c_long), _p(d, C.c_long), _p(N, C.c_long), _p(dist, C.c_double))
        return ids[:n], d[:n], N[:n], dist[:n]

    def pdist_all(self, seq, start, end):
        seq = np.ascontiguousarray(seq, np.int8)
        d = np.zeros(self.n, np.int64); N = np.zeros(self.n, np.int64)
        lib().orc_pdist_all(self.h, _p(seq, C.c_int8), C.c_int(start), C.c_int(end), _p(d, C.c_long), _p(N, C.c_long))
        return d, N

    def estimate(self, seq, start, end, node, dist, weighted=False):
        seq = np.ascontiguousarray(seq, np.int8)
        out = np.zeros(4); nodes = np.zeros(3, np.int32)
        lib().orc_estimate(self.h, _p(seq, C.c_int8), C.c_int(start), C.c_int(end), C.c_long(int(node)), C.c_double(dist),
                           C.c_int(int(weighted)), _p(out, C.c_double), _p(nodes, C.c_int))
        return dict(ratio=out[0], wnr=out[1], loglik=out[2], wuv=out[3], cNode=int(nodes[0]), pNode=int(nodes[1]), aNode=int(nodes[2]))

    def place(self, seq, start, end, c_node, ratio0, wnr0, max_height=float("inf")):
        seq = np.ascontiguousarray(seq, np.int8)
        out = np.zeros(4); a = C.c_int(0)
        it = lib().orc_place(self.h, _p(seq, C.c_int8), C.c_int(start), C.c_int(end), C.c_int(int(c_node)), C.c_double(ratio0),
                             C.c_double(wnr0), C.c_double(max_height), _p(out, C.c_double), C.byref(a))
        return dict(loglik=out[0], wnr=out[1], ratio=out[2], height=out[3], aNode=a.value, iters=it)

    def assign(self, seq, start, end, opts=None):
        opts = opts or default_opts()
        seq = np.ascontiguousarray(seq, np.int8)
        m = max(opts.maxNSeed, 1)
        ni = np.zeros((m, 4), np.int32); nd = np.zeros((m, 8)); ns = C.c_int(0)
        sid = np.zeros(m, np.int64); sd = np.zeros(m, np.int64); sn = np.zeros(m, np.int64); est = np.zeros((m, 3))
        fo = np.zeros(m, np.int32)
        n = lib().orc_assign(self.h, _p(seq, C.c_int8), C.c_int(start), C.c_int(end), C.byref(opts), _p(ni, C.c_int), _p(nd, C.c_double),
                             C.byref(ns), _p(sid, C.c_long), _p(sd, C.c_long), _p(sn, C.c_long), _p(est, C.c_double), _p(fo, C.c_int))
        k = ns.value
        return dict(n=n, nodes=ni[:n], vals=nd[:n], seed_ids=sid[:k], seed_d=sd[:k], seed_N=sn[:k], est=est[:k], filt_order=fo[:n])

    def chimera(self, seq, start, end, opts=None, num_seg=2, max_chimera_error=None, min_chimera_lod=0.0, seeds=None):
        """Chimera check of one aligned read (src/hmmufotu.cpp:653-691); seeds=None takes getSeed's."""
        opts = opts or default_opts()
        if max_chimera_error is None:
            max_chimera_error = opts.maxError / num_seg      # src/hmmufotu.cpp:147
        seq = np.ascontiguousarray(seq, np.int8)
        oi = np.zeros(16, np.int32); od = np.zeros(12)
        sd = np.ascontiguousarray(seeds if seeds is not None else np.zeros(0), np.int64)
        lib().orc_chimera(self.h, _p(seq, C.c_int8), C.c_int(start), C.c_int(end), C.byref(opts), C.c_int(num_seg), C.c_double(max_chimera_error),
                          C.c_double(min_chimera_lod), C.c_int(-1 if seeds is None else len(sd)), _p(sd, C.c_long), _p(oi, C.c_int), _p(od, C.c_double))
        return dict(checked=bool(oi[0] == 1), is_chimera=bool(oi[1] == 1), lod=od[0],
                    seg5=dict(c=int(oi[2]), p=int(oi[3]), a=int(oi[4]), start=int(oi[5]), end=int(oi[6]), ratio=od[1], wnr=od[2], loglik=od[3], est_loglik=od[4]),
                    seg3=dict(c=int(oi[7]), p=int(oi[8]), a=int(oi[9]), start=int(oi[10]), end=int(oi[11]), ratio=od[5], wnr=od[6], loglik=od[7], est_loglik=od[8]),
                    n5=int(oi[12]), n3=int(oi[13]), alt5_loglik=od[9], alt3_loglik=od[10])

    def __del__(self):
        try:
            lib().orc_tree_free(self.h)
        except Exception:
            pass


def tree_evaluate(parent, blen, leaf_seq, model: Model, dg_r=None):
    parent = np.ascontiguousarray(parent, np.int32); blen = np.ascontiguousarray(blen, np.float64)
    seq = np.ascontiguousarray(leaf_seq, np.int8).copy()
    n, L = seq.shape
    dgr = np.ascontiguousarray(dg_r if dg_r is not None else np.zeros(0), np.float64)
    up = np.zeros((n, L, 4)); down = np.zeros((n, L, 4)); root = np.zeros((L, 4)); h = np.zeros(n)
    lib().orc_tree_evaluate(C.c_int(n), C.c_int(L), _p(parent, C.c_int), _p(blen, C.c_double), _p(seq, C.c_int8), model.h,
                            C.c_int(len(dgr)), _p(dgr, C.c_double), _p(up, C.c_double), _p(down, C.c_double), _p(root, C.c_double), _p(h, C.c_double))
    up[0] = root
    return up, down, seq, h


def pipeline_batch(hmm: Hmm, tree: Tree, reads, vpaths, mates=None, mvpaths=None, opts=None, threads=0, want_align=False, want_cands=False,
                   mode=0, seeds=None, want_lib=False):
    """Whole per-read task on the CPU (OpenMP over reads).  reads/mates: list of str.
    want_cands: also the candidates of every read in filterPlacements order (node, estimated loglik, estimated ratio;
    rows padded with -1 / NaN to 64), their placed (ratio, wnr, height) and (outer, EM) iteration counts, and the position of the
    final pick in that order.
    mode 0: the whole task.  mode 1: alignment + getSeed only; returns seed_cnt / seed_ids [n][64] (and with want_lib the list under the
    reference's literal std::sort, lib_ids, + tie_info, from the same scan).  mode 2: the rest of the task on GIVEN seed lists
    (seeds = (cnt [n], ids [n][64])) — the tree then only needs the message rows of those nodes (Tree.set_rows)."""
    opts = opts or default_opts()
    n = len(reads)
    cat = "".join(reads).encode(); offs = np.zeros(n + 1, np.int64); offs[1:] = np.cumsum([len(r) for r in reads])
    vp = np.ascontiguousarray(vpaths, np.int32).reshape(n, 2, 6)
    if mates is not None:
        mcat = "".join(mates).encode(); moffs = np.zeros(n + 1, np.int64); moffs[1:] = np.cumsum([len(r) for r in mates])
        mvp = np.ascontiguousarray(mvpaths, np.int32).reshape(n, 2, 6)
    ai = np.zeros((n, 8), np.int32); cost = np.zeros(n); bi = np.zeros((n, 4), np.int32); bd = np.zeros((n, 8))
    nc = np.zeros(n, np.int32); st = np.zeros(4)
    aln = np.zeros((n, hmm.L), np.uint8) if want_align else None
    cn = np.full((n, 64), -1, np.int32) if want_cands else None
    ce = np.full((n, 64), np.nan) if want_cands else None
    cr = np.full((n, 64), np.nan) if want_cands else None
    bp = np.full(n, -1, np.int32) if want_cands else None
    cp = np.full((n, 64, 3), np.nan) if want_cands else None
    ci = np.zeros((n, 64, 2), np.int32) if want_cands else None
    if mode == 2:
        scnt = np.ascontiguousarray(seeds[0], np.int32); sids = np.ascontiguousarray(seeds[1], np.int32).reshape(n, 64)
    else:
        scnt = np.zeros(n, np.int32); sids = np.full((n, 64), -1, np.int32)
    lids = np.full((n, 64), -1, np.int32) if (mode == 1 and want_lib) else None
    tinfo = np.zeros((n, 4), np.int32) if lids is not None else None
    extra = C.c_double(0)
    lib().orc_pipeline_batch(hmm.h, tree.h, C.c_int(n), cat, _p(offs, C.c_long),
                             mcat if mates is not None else None, _p(moffs, C.c_long) if mates is not None else None,
                             _p(vp, C.c_int), _p(mvp, C.c_int) if mates is not None else None, C.byref(opts), C.c_int(threads),
                             _p(ai, C.c_int), _p(cost, C.c_double), _p(aln, C.c_char) if aln is not None else None,
                             _p(bi, C.c_int), _p(bd, C.c_double), _p(nc, C.c_int), _p(st, C.c_double),
                             _p(cn, C.c_int) if want_cands else None, _p(ce, C.c_double) if want_cands else None,
                             _p(cr, C.c_double) if want_cands else None, _p(bp, C.c_int) if want_cands else None,
                             _p(cp, C.c_double) if want_cands else None, _p(ci, C.c_int) if want_cands else None,
                             C.c_int(mode), _p(scnt, C.c_int), _p(sids, C.c_int), _p(lids, C.c_int) if lids is not None else None,
                             _p(tinfo, C.c_int) if tinfo is not None else None, C.byref(extra))
    return dict(aln_ints=ai, cost=cost, align=aln, best_nodes=bi, best_vals=bd, n_cand=nc, stage_sec=st,
                cand_node=cn, cand_est=ce, cand_ratio0=cr, best_pos=bp, cand_placed=cp, cand_iters=ci,
                seed_cnt=scnt, seed_ids=sids, lib_ids=lids, tie_info=tinfo, extra_thread_sec=extra.value)


def tie_report(hmm: Hmm, tree: Tree, reads, vpaths, mates=None, mvpaths=None, opts=None, threads=0, phase1=None, tree2=None):
    """the per-read task under the reference's seed order (first max_nseed of a literal std::sort on dist alone,
    src/HmmUFOtu_main.cpp:139, src/hmmufotu.cpp:646-647) against the product's (dist, node id) order.  Both lists come from ONE scan
    per read (pipeline_batch mode 1, want_lib); the reads whose ordered lists differ are run through the rest of the task on each
    list (mode 2; reads with equal lists give estimateSeq the same inputs in the same order and cannot differ).
    phase1: a mode-1 result to reuse; tree2: the tree for the mode-2 runs (e.g. one holding only the seed nodes' message rows).
    Returns (per_read dict of arrays, summary dict)."""
    opts = opts or default_opts(tieMode=0)       # phase 1's own list (seed_ids) is the (dist, node id) one; lib_ids is the literal std::sort's
    n = len(reads)
    p1 = phase1 if phase1 is not None else pipeline_batch(hmm, tree, reads, vpaths, mates, mvpaths, opts, threads, mode=1, want_lib=True)
    ok = p1["aln_ints"][:, 7] == 1
    cnt, sid, lid, ti = p1["seed_cnt"], p1["seed_ids"], p1["lib_ids"], p1["tie_info"]
    order_diff = ok & (sid != lid).any(axis=1)
    only_lib = np.array([len(set(lid[i, :cnt[i]]) - set(sid[i, :cnt[i]])) if order_diff[i] else 0 for i in range(n)])
    set_diff = only_lib > 0
    idx = np.nonzero(order_diff)[0]
    pick_mask = np.zeros(n, np.int32); filt_diff = np.zeros(n, bool); via_tie = np.zeros(n, bool)
    picks = np.full((n, 2, 3), -1, np.int32)
    rl = None
    if len(idx):
        t2 = tree2 if tree2 is not None else tree
        sub = lambda x: None if x is None else [x[i] for i in idx]
        vp = np.ascontiguousarray(vpaths, np.int32).reshape(n, 2, 6)[idx]
        mvp = None if mvpaths is None else np.ascontiguousarray(mvpaths, np.int32).reshape(n, 2, 6)[idx]
        rs = pipeline_batch(hmm, t2, sub(reads), vp, sub(mates), mvp, opts, threads, want_cands=True, mode=2, seeds=(cnt[idx], sid[idx]))
        rl = pipeline_batch(hmm, t2, sub(reads), vp, sub(mates), mvp, opts, threads, want_cands=True, mode=2, seeds=(cnt[idx], lid[idx]))
        for k, i in enumerate(idx):
            cs = set(rs["cand_node"][k, :rs["n_cand"][k]]); cl = set(rl["cand_node"][k, :rl["n_cand"][k]])
            filt_diff[i] = cs != cl
            bs, bl = rs["best_nodes"][k, :3], rl["best_nodes"][k, :3]
            picks[i, 0], picks[i, 1] = bs, bl
            pick_mask[i] = int(bs[0] != bl[0]) | (int(bs[1] != bl[1]) << 1) | (int(bs[2] != bl[2]) << 2)
            if filt_diff[i] or pick_mask[i]:
                only_l = set(lid[i, :cnt[i]]) - set(sid[i, :cnt[i]]); only_s = set(sid[i, :cnt[i]]) - set(lid[i, :cnt[i]])
                via_tie[i] = bool((cl & only_l) or (cs & only_s))
    pick = pick_mask != 0
    # a seed that only one list holds can only come from the tie at the cut-off: both lists are valid ascending sorts of the same
    # distances, so they hold every node below the cut-off distance and end AT it (tie_info[:, 2])
    summ = dict(reads=int(ok.sum()), max_nseed=int(opts.maxNSeed),
                seed_order_differs=int(order_diff.sum()), seed_set_differs=int(set_diff.sum()),
                seed_set_diffs_all_exact_cutoff_ties=bool((ti[set_diff, 2] == 1).all() and (ti[set_diff, 0] > ti[set_diff, 1]).all()),
                mean_nodes_tied_at_cutoff=float(ti[ok, 0].mean()) if ok.any() else 0.0,
                reads_whose_cutoff_tie_reaches_beyond_the_list=int(((ti[:, 0] > ti[:, 1]) & ok).sum()),
                reads_with_nan_dist=int((ti[:, 3] != 0).sum()),
                filtered_candidate_set_differs=int(filt_diff.sum()),
                final_c_node_differs=int(((pick_mask & 1) != 0).sum()), final_p_node_differs=int(((pick_mask & 2) != 0).sum()),
                final_a_node_differs=int(((pick_mask & 4) != 0).sum()), final_pick_differs=int(pick.sum()),
                final_pick_diffs_traced_to_a_cutoff_tie=int(via_tie[pick].sum()))
    return dict(order_differs=order_diff, seeds_only_in_libstdcxx=only_lib, filtered_set_differs=filt_diff, pick_mask=pick_mask,
                picks=picks, via_cutoff_seed=via_tie, phase1=p1, lib_idx=idx, lib_run=rl, stable_run=rs if len(idx) else None), summ


def std_sort_prefix(dist, k):
    """first k indices of a literal std::sort of PTLocs on dist alone (orc_std_sort_prefix)"""
    dist = np.ascontiguousarray(dist, np.float64); n = len(dist)
    out = np.zeros(min(n, k), np.int32)
    lib().orc_std_sort_prefix(_p(dist, C.c_double), C.c_long(n), C.c_long(k), _p(out, C.c_int))
    return out


def antiqsort(n):
    """an input that drives std::sort into its heap-sort branch (orc_antiqsort)"""
    out = np.zeros(n)
    lib().orc_antiqsort(C.c_long(n), _p(out, C.c_double))
    return out


def std_sort_desc(keys):
    """the order a literal std::sort(rbegin, rend, less-by-key) leaves (orc_std_sort_desc)"""
    keys = np.ascontiguousarray(keys, np.float64)
    out = np.zeros(len(keys), np.int32)
    lib().orc_std_sort_desc(_p(keys, C.c_double), C.c_int(len(keys)), _p(out, C.c_int))
    return out


def max_threads():
    return lib().orc_max_threads()
