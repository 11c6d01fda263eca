"""ctypes host binding of libhmmufotu_amd.so (include/hmmufotu_amd.h).

The Python side mirrors the per-read surface of the reference (src/HmmUFOtu_main.h:70-113):
`Batch.align / get_seed / estimate_seq / filter_placements / place_seq / calc_q_values`
operate on a whole batch of reads.  There is no CPU fallback: `load_library()` raises if the
HIP extension is missing and every compute call raises `EngineError` without a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HU_LIB") or os.path.join(_HERE, "libhmmufotu_amd.so")      # HU_LIB: another build of the same library (kernel-geometry experiments)
_LIB = None

HU_MAX_SEEDS = 64
T_NAMES = ["viterbi", "align_build", "seed_pdist", "seed_topk", "estimate", "place", "_6", "_7"]
MODE = {"global": 0, "local": 1, "ngcl": 2, "cgnl": 3}


class EngineError(RuntimeError):
    pass


class ProfileDesc(C.Structure):
    _fields_ = [("K", C.c_int32), ("L", C.c_int32), ("EM", C.POINTER(C.c_double)), ("EI", C.POINTER(C.c_double)),
                ("T", C.POINTER(C.c_double)), ("p2cs", C.POINTER(C.c_int32))]


class ModelDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("pi", C.c_double * 4), ("par", C.c_double * 16), ("dg_k", C.c_int32),
                ("dg_rate", C.c_double * 16)]


class TreeDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("cs_len", C.c_int32), ("parent", C.POINTER(C.c_int32)), ("blen", C.POINTER(C.c_double)),
                ("seq", C.POINTER(C.c_int8)), ("up", C.c_void_p), ("down", C.c_void_p), ("height", C.POINTER(C.c_double)),
                ("anno_id", C.POINTER(C.c_int32)), ("anno_dist", C.POINTER(C.c_double)), ("win_start", C.c_int64),
                ("win_len", C.c_int64), ("msgs_on_device", C.c_int32)]


class Opts(C.Structure):
    _fields_ = [("align_mode", C.c_int32), ("max_nseed", C.c_int32), ("max_diff", C.c_double), ("max_height", C.c_double),
                ("max_error", C.c_double), ("weighted", C.c_int32), ("only_ml", C.c_int32), ("prior", C.c_int32),
                ("ignore_orient", C.c_int32), ("fix_root_loglik", C.c_int32), ("seed_order", C.c_int32)]


class AlignRec(C.Structure):
    _fields_ = [("seq_start", C.c_int32), ("seq_end", C.c_int32), ("hmm_start", C.c_int32), ("hmm_end", C.c_int32),
                ("cs_start", C.c_int32), ("cs_end", C.c_int32), ("status", C.c_int32), ("used_full", C.c_int32), ("cost", C.c_double)]


class PlaceRec(C.Structure):
    _fields_ = [("c_node", C.c_int32), ("p_node", C.c_int32), ("a_node", C.c_int32), ("n_cand", C.c_int32),
                ("wuv", C.c_double), ("ratio", C.c_double), ("wnr", C.c_double), ("loglik", C.c_double), ("height", C.c_double),
                ("q_place", C.c_double), ("q_taxon", C.c_double), ("anno_dist", C.c_double), ("est_loglik", C.c_double),
                ("root_loglik", C.c_double)]


READ_OK, READ_INVALID, READ_CHIMERA, READ_OUT_OF_WINDOW = 1, 0, 2, 16     # hu_align_rec.status (HU_READ_*)
ALIGN_DTYPE = np.dtype([("seq_start", "i4"), ("seq_end", "i4"), ("hmm_start", "i4"), ("hmm_end", "i4"), ("cs_start", "i4"),
                        ("cs_end", "i4"), ("status", "i4"), ("used_full", "i4"), ("cost", "f8")])
PLACE_DTYPE = np.dtype([("c_node", "i4"), ("p_node", "i4"), ("a_node", "i4"), ("n_cand", "i4"), ("wuv", "f8"), ("ratio", "f8"),
                        ("wnr", "f8"), ("loglik", "f8"), ("height", "f8"), ("q_place", "f8"), ("q_taxon", "f8"),
                        ("anno_dist", "f8"), ("est_loglik", "f8"), ("root_loglik", "f8")])


class ChimeraOpts(C.Structure):
    _fields_ = [("num_seg", C.c_int32), ("reserved", C.c_int32), ("max_chimera_error", C.c_double), ("min_chimera_lod", C.c_double)]


CHIMERA_DTYPE = np.dtype([("checked", "i4"), ("is_chimera", "i4"), ("seg5_start", "i4"), ("seg5_end", "i4"), ("seg3_start", "i4"),
                          ("seg3_end", "i4"), ("n_seg5", "i4"), ("n_seg3", "i4"), ("seg5", PLACE_DTYPE), ("seg3", PLACE_DTYPE),
                          ("alt5_loglik", "f8"), ("alt3_loglik", "f8"), ("lod", "f8")])


def build_library(force: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    deps = [os.path.join(src, f) for f in os.listdir(src) if f.endswith((".hip", ".cpp", ".h"))]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "hmmufotu_amd.h"))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(d) > os.path.getmtime(LIB_PATH) for d in deps):
        subprocess.check_call(["make", "-C", src, "-B", "all"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def load_library():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)" % LIB_PATH)
        # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64.so (same soname as the system one),
        # and a process that has both mapped sees the GPU from only the first.  Map torch's copy first (without importing
        # torch) so that the engine binds to it whichever of the two is imported first.
        try:
            import importlib.util
            spec = importlib.util.find_spec("torch")
            hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else ""
            if hip and os.path.exists(hip):
                C.CDLL(hip, mode=C.RTLD_GLOBAL)
        except OSError:
            pass
        _LIB = C.CDLL(LIB_PATH)
        _LIB.hu_last_error.restype = C.c_char_p
    return _LIB


def _chk(rc: int):
    if rc != 0:
        raise EngineError("hmmufotu_amd error %d: %s" % (rc, load_library().hu_last_error().decode()))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def device_count() -> int:
    return int(load_library().hu_device_count())


def default_opts(**kw) -> Opts:
    o = Opts()
    load_library().hu_default_opts(C.byref(o))
    for k, v in kw.items():
        if k == "align_mode" and isinstance(v, str):
            v = MODE[v]
        setattr(o, k, v)
    return o


def model_desc(type_id: int, pi, par, dg_rates=None) -> ModelDesc:
    m = ModelDesc()
    m.type = int(type_id)
    for i in range(4):
        m.pi[i] = float(pi[i])
    par = np.asarray(par, np.float64).ravel()
    for i in range(min(16, len(par))):
        m.par[i] = float(par[i])
    dg = np.asarray(dg_rates if dg_rates is not None else [], np.float64)
    m.dg_k = len(dg)
    for i in range(len(dg)):
        m.dg_rate[i] = float(dg[i])
    return m


def model_spectral(md: ModelDesc):
    """Host-only: (U, lam, U1) with P(t) = U diag(exp(lam t)) U1 as the kernels use it."""
    U = np.zeros(16); lam = np.zeros(4); U1 = np.zeros(16)
    _chk(load_library().hu_model_spectral(C.byref(md), _p(U, C.c_double), _p(lam, C.c_double), _p(U1, C.c_double)))
    return U.reshape(4, 4), lam, U1.reshape(4, 4)


def parse_files(hmm_path=None, ptu_path=None):
    """Host-only parse of the reference's .hmm / .ptu formats (no device needed)."""
    L_ = load_library()
    K = C.c_int32(0); L = C.c_int32(0); n = C.c_int32(0); root = C.c_int32(0); md = ModelDesc()
    hp = hmm_path.encode() if hmm_path else None
    pp = ptu_path.encode() if ptu_path else None
    _chk(L_.hu_files_parse(hp, pp, C.byref(K), C.byref(L), C.byref(n), C.byref(root), *([None] * 12), C.byref(md), 0))
    out = dict(K=K.value, L=L.value, n_nodes=n.value, root=root.value, model=md)
    args = [None] * 12
    if hmm_path:
        out.update(EM=np.zeros((K.value + 1, 4)), EI=np.zeros((K.value + 1, 4)), T=np.zeros((K.value + 1, 7)),
                   p2cs=np.zeros(K.value + 1, np.int32), entry_cost=np.zeros(K.value + 1), exit_cost=np.zeros(K.value + 1))
        args[0:6] = [_p(out["EM"], C.c_double), _p(out["EI"], C.c_double), _p(out["T"], C.c_double), _p(out["p2cs"], C.c_int32),
                     _p(out["entry_cost"], C.c_double), _p(out["exit_cost"], C.c_double)]
    if ptu_path:
        nn, ll = n.value, L.value
        out.update(parent=np.zeros(nn, np.int32), blen=np.zeros(nn), seq=np.zeros((nn, ll), np.int8), height=np.zeros(nn),
                   up=np.zeros((nn, ll, 4)), down=np.zeros((nn, ll, 4)))
        args[6:12] = [_p(out["parent"], C.c_int32), _p(out["blen"], C.c_double), _p(out["seq"], C.c_int8), _p(out["height"], C.c_double),
                      _p(out["up"], C.c_double), _p(out["down"], C.c_double)]
    _chk(L_.hu_files_parse(hp, pp, C.byref(K), C.byref(L), C.byref(n), C.byref(root), *args, C.byref(md), 1))
    return out


def tree_evaluate(parent, blen, seq, model: ModelDesc, up_ptr: int, down_ptr: int, win_start=0, win_len=0, device=0):
    """hu_tree_evaluate: fills the DEVICE buffers at up_ptr/down_ptr ([n][win_len][4] float64, log space),
    returns (seq with inferred inner rows, heights)."""
    parent = np.ascontiguousarray(parent, np.int32); blen = np.ascontiguousarray(blen, np.float64)
    seq = np.ascontiguousarray(seq, np.int8).copy()
    n, L = seq.shape
    h = np.zeros(n)
    _chk(load_library().hu_tree_evaluate(C.c_int32(n), C.c_int32(L), _p(parent, C.c_int32), _p(blen, C.c_double), _p(seq, C.c_int8),
                                         C.byref(model), C.c_int(device), C.c_int64(win_start), C.c_int64(win_len),
                                         C.c_void_p(int(up_ptr)), C.c_void_p(int(down_ptr)), _p(h, C.c_double)))
    return seq, h


def write_ptu(path, parent, blen, seq, up, down, height, model: ModelDesc, names=None, annos=None, anno_dist=None, model_text=None,
              dg_alpha=0.0, dg_breaks=None, msgs_on_device=False):
    """hu_ptu_write: the database file in the reference's .ptu format; up / down are [n][cs_len][4] arrays, or device pointers (ints)
    with msgs_on_device=True."""
    keep = dict(parent=np.ascontiguousarray(parent, np.int32), blen=np.ascontiguousarray(blen, np.float64), seq=np.ascontiguousarray(seq, np.int8),
                height=np.ascontiguousarray(height, np.float64))
    td = TreeDesc()
    td.n_nodes, td.cs_len = keep["seq"].shape
    td.parent = _p(keep["parent"], C.c_int32); td.blen = _p(keep["blen"], C.c_double); td.seq = _p(keep["seq"], C.c_int8); td.height = _p(keep["height"], C.c_double)
    if msgs_on_device:
        td.up = C.c_void_p(int(up)); td.down = C.c_void_p(int(down)); td.msgs_on_device = 1
    else:
        keep["up"] = np.ascontiguousarray(up, np.float64); keep["down"] = np.ascontiguousarray(down, np.float64)
        td.up = keep["up"].ctypes.data_as(C.c_void_p); td.down = keep["down"].ctypes.data_as(C.c_void_p)
    if anno_dist is not None:
        keep["ad"] = np.ascontiguousarray(anno_dist, np.float64); td.anno_dist = _p(keep["ad"], C.c_double)
    arr = lambda xs: (C.c_char_p * len(xs))(*[x.encode() for x in xs]) if xs is not None else None
    br = np.ascontiguousarray(dg_breaks, np.float64) if dg_breaks is not None else None
    _chk(load_library().hu_ptu_write(path.encode(), C.byref(td), arr(names), arr(annos), C.byref(model), model_text.encode() if model_text else None,
                                     C.c_double(dg_alpha), _p(br, C.c_double) if br is not None else None))


class SeedIndex:
    """Host k-mer index standing in for the CSFM lookup of alignSeq (hu_seed_index_*)."""

    def __init__(self, parent, seq, hmm, seed_len=20, csfm=None):
        """from the leaf rows of a tree (parent, seq), or — csfm=<path> — from the reference's own <DB>.csfm (hu_seed_index_load_csfm)"""
        self.p2cs = np.ascontiguousarray(hmm.p2cs, np.int32)
        self.h = C.c_void_p()
        if csfm is not None:
            _chk(load_library().hu_seed_index_load_csfm(str(csfm).encode(), C.c_int32(int(hmm.K)), _p(self.p2cs, C.c_int32), C.c_int32(seed_len), C.byref(self.h)))
        else:
            self.parent = np.ascontiguousarray(parent, np.int32); self.seq = np.ascontiguousarray(seq, np.int8)
            n, L = self.seq.shape
            _chk(load_library().hu_seed_index_create(C.c_int32(n), C.c_int32(L), _p(self.parent, C.c_int32), _p(self.seq, C.c_int8),
                                                     C.c_int32(int(hmm.K)), _p(self.p2cs, C.c_int32), C.c_int32(seed_len), C.byref(self.h)))
        lib = load_library(); lib.hu_seed_index_size.restype = C.c_int64; lib.hu_seed_index_bytes.restype = C.c_int64
        lib.hu_seed_index_occurrences.restype = C.c_int64
        self.size = int(lib.hu_seed_index_size(self.h))
        npos = C.c_int64(0)
        self.bytes = int(lib.hu_seed_index_bytes(self.h, C.byref(npos)))
        self.positions = int(npos.value)

    def occurrences(self, kmer: str, cap=4096):
        """all occurrences of one seed in index order: (sequence number, offset in it, first CS column)"""
        a = np.zeros(cap, np.int32); b = np.zeros(cap, np.int32); c = np.zeros(cap, np.int32)
        n = int(load_library().hu_seed_index_occurrences(self.h, kmer.encode(), _p(a, C.c_int32), _p(b, C.c_int32), _p(c, C.c_int32), C.c_int64(cap)))
        m = min(n, cap)
        return n, a[:m], b[:m], c[:m]

    def locate_first(self, kmer: str):
        """CSFMIndex::locateFirst + count: (csStart, csEnd) 1-based of the first hit (0, 0 = none), number of hits"""
        a = C.c_int32(0); b = C.c_int32(0); n = C.c_int64(0)
        load_library().hu_seed_index_locate_first(self.h, kmer.encode(), C.byref(a), C.byref(b), C.byref(n))
        return int(a.value), int(b.value), int(n.value)

    def lookup(self, reads, seed_region=50, align_mode=0):
        n = len(reads)
        cat = "".join(reads).encode("latin1")
        offs = np.zeros(n + 1, np.int64); offs[1:] = np.cumsum([len(r) for r in reads])
        vp = np.zeros((n, 2, 6), np.int32)
        _chk(load_library().hu_seed_index_lookup(self.h, C.c_int(n), cat, _p(offs, C.c_int64), C.c_int(seed_region), C.c_int(align_mode),
                                                 _p(vp, C.c_int32)))
        return vp

    def lookup_random(self, reads, seed, first_read=0, seed_region=50, align_mode=0):
        """CSFMIndex::locateOne's hit choice (hu_seed_index_lookup_random): a member of each seed's hit range drawn from a hash of
        (seed, number of the read = first_read + index, seed position)"""
        n = len(reads)
        cat = "".join(reads).encode("latin1")
        offs = np.zeros(n + 1, np.int64); offs[1:] = np.cumsum([len(r) for r in reads])
        vp = np.zeros((n, 2, 6), np.int32)
        _chk(load_library().hu_seed_index_lookup_random(self.h, C.c_int(n), cat, _p(offs, C.c_int64), C.c_int(seed_region), C.c_int(align_mode),
                                                        C.c_uint64(seed), C.c_int64(first_read), _p(vp, C.c_int32)))
        return vp

    def lookup_packed(self, cat, offs, seed_region=50, align_mode=0):
        """the same on reads that are already one byte buffer + offsets [n + 1] (what a FASTA parser holds)"""
        offs = np.ascontiguousarray(offs, np.int64); n = len(offs) - 1
        vp = np.zeros((n, 2, 6), np.int32)
        _chk(load_library().hu_seed_index_lookup(self.h, C.c_int(n), cat.ctypes.data_as(C.c_char_p) if isinstance(cat, np.ndarray) else cat,
                                                 _p(offs, C.c_int64), C.c_int(seed_region), C.c_int(align_mode), _p(vp, C.c_int32)))
        return vp

    def __del__(self):
        try:
            load_library().hu_seed_index_destroy(self.h)
        except Exception:
            pass


class Database:
    """Profile + pre-evaluated tree packed once into HBM (hu_db)."""

    def __init__(self, handle, keep=None):
        self.h = handle
        self._keep = keep
        K = C.c_int32(); L = C.c_int32(); n = C.c_int32(); r = C.c_int32(); hb = C.c_int64()
        _chk(load_library().hu_db_info(self.h, C.byref(K), C.byref(L), C.byref(n), C.byref(r), C.byref(hb)))
        self.K, self.cs_len, self.n_nodes, self.root, self.hbm_bytes = K.value, L.value, n.value, r.value, hb.value

    @classmethod
    def from_arrays(cls, hmm, parent, blen, seq, up, down, height, model: ModelDesc, anno_id=None, anno_dist=None,
                    win_start=0, win_len=0, device=0, msgs_on_device=False):
        lib = load_library()
        keep = dict(EM=np.ascontiguousarray(hmm.EM, np.float64), EI=np.ascontiguousarray(hmm.EI, np.float64),
                    T=np.ascontiguousarray(hmm.T, np.float64), p2cs=np.ascontiguousarray(hmm.p2cs, np.int32),
                    parent=np.ascontiguousarray(parent, np.int32), blen=np.ascontiguousarray(blen, np.float64),
                    seq=np.ascontiguousarray(seq, np.int8), height=np.ascontiguousarray(height, np.float64))
        pd = ProfileDesc(int(hmm.K), int(hmm.L), _p(keep["EM"], C.c_double), _p(keep["EI"], C.c_double), _p(keep["T"], C.c_double),
                         _p(keep["p2cs"], C.c_int32))
        td = TreeDesc()
        td.n_nodes, td.cs_len = keep["seq"].shape
        td.parent = _p(keep["parent"], C.c_int32); td.blen = _p(keep["blen"], C.c_double); td.seq = _p(keep["seq"], C.c_int8)
        td.height = _p(keep["height"], C.c_double)
        if msgs_on_device:
            td.up = C.c_void_p(int(up)); td.down = C.c_void_p(int(down)); td.msgs_on_device = 1
        else:
            keep["up"] = np.ascontiguousarray(up, np.float64); keep["down"] = np.ascontiguousarray(down, np.float64)
            td.up = keep["up"].ctypes.data_as(C.c_void_p); td.down = keep["down"].ctypes.data_as(C.c_void_p)
        if anno_id is not None:
            keep["anno"] = np.ascontiguousarray(anno_id, np.int32); td.anno_id = _p(keep["anno"], C.c_int32)
        if anno_dist is not None:
            keep["annod"] = np.ascontiguousarray(anno_dist, np.float64); td.anno_dist = _p(keep["annod"], C.c_double)
        td.win_start = int(win_start); td.win_len = int(win_len)
        h = C.c_void_p()
        _chk(lib.hu_db_create(C.byref(pd), C.byref(td), C.byref(model), C.c_int(device), C.byref(h)))
        return cls(h, keep if msgs_on_device else None)

    @classmethod
    def from_synth(cls, db, device=0):
        md = model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r if db.dg_k > 0 else None)
        return cls.from_arrays(db.hmm, db.parent, db.blen, db.seq, db.up, db.down, db.height, md, db.anno_id, db.anno_dist, device=device)

    @classmethod
    def load(cls, hmm_path: str, ptu_path: str, device=0, win_start=0, win_len=0):
        """hu_db_load / hu_db_load_window (win_len > 0: only the messages of the CS columns [win_start, win_start + win_len) stay on the device)"""
        h = C.c_void_p()
        _chk(load_library().hu_db_load_window(hmm_path.encode(), ptu_path.encode(), C.c_int(device), C.c_int64(win_start), C.c_int64(win_len), C.byref(h)))
        return cls(h)

    def model_pr(self, t):
        t = np.ascontiguousarray(t, np.float64).ravel()
        P = np.zeros((len(t), 4, 4))
        _chk(load_library().hu_db_model_pr(self.h, C.c_int(len(t)), _p(t, C.c_double), _p(P, C.c_double)))
        return P

    def build_align_path(self, cs_start, cs_end, cs: str, cs_from, cs_to):
        out = np.zeros(6, np.int32)
        _chk(load_library().hu_build_align_path(self.h, C.c_int(cs_start), C.c_int(cs_end), cs.encode(), C.c_int(cs_from), C.c_int(cs_to),
                                                _p(out, C.c_int32)))
        return out

    def profile(self):
        K = self.K
        out = dict(EM=np.zeros((K + 1, 4)), EI=np.zeros((K + 1, 4)), T=np.zeros((K + 1, 7)), p2cs=np.zeros(K + 1, np.int32),
                   entry_cost=np.zeros(K + 1), exit_cost=np.zeros(K + 1))
        _chk(load_library().hu_db_get_profile(self.h, _p(out["EM"], C.c_double), _p(out["EI"], C.c_double), _p(out["T"], C.c_double),
                                              _p(out["p2cs"], C.c_int32), _p(out["entry_cost"], C.c_double), _p(out["exit_cost"], C.c_double)))
        return out

    def close(self):
        if self.h:
            load_library().hu_db_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Window(C.Structure):
    _fields_ = [("win_start", C.c_int64), ("win_len", C.c_int64)]


def windows_plan(cs_len: int, n_win: int, overlap: int):
    """hu_windows_plan: n_win column windows [(start, len)] covering [0, cs_len), neighbours overlapping by `overlap` columns"""
    w = (Window * n_win)()
    _chk(load_library().hu_windows_plan(C.c_int64(cs_len), C.c_int(n_win), C.c_int64(overlap), w))
    return [(int(x.win_start), int(x.win_len)) for x in w]


class WindowedDatabase:
    """Column-window sharding (last row; include/hmmufotu_amd.h): one database held as W column windows — one hu_db per window,
    usually one per device — for message sets beyond one GPU's HBM.  PTUnrooted::load keeps every column of every edge (src/PhyloTreeUnrooted.cpp:496-535);
    here a read is routed to the window that holds the columns its seeds point at, and re-routed ONCE by its exact region when it comes back
    HU_READ_OUT_OF_WINDOW.  The windows never exchange anything."""

    def __init__(self, dbs, windows):
        assert len(dbs) == len(windows) >= 1
        self.dbs = list(dbs); self.windows = [(int(a), int(b)) for a, b in windows]
        self._w = (Window * len(windows))(*[Window(a, b) for a, b in self.windows])
        self.last = {}

    @classmethod
    def from_synth(cls, db, n_win, overlap, devices=None):
        md = model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r if db.dg_k > 0 else None)
        wins = windows_plan(db.cs_len, n_win, overlap)
        dbs = [Database.from_arrays(db.hmm, db.parent, db.blen, db.seq, db.up[:, a:a + l], db.down[:, a:a + l], db.height, md, db.anno_id, db.anno_dist,
                                    win_start=a, win_len=l, device=(devices[i] if devices else 0)) for i, (a, l) in enumerate(wins)]
        return cls(dbs, wins)

    @classmethod
    def load(cls, hmm_path, ptu_path, cs_len, n_win, overlap, devices=None):
        wins = windows_plan(cs_len, n_win, overlap)
        return cls([Database.load(hmm_path, ptu_path, devices[i] if devices else 0, a, l) for i, (a, l) in enumerate(wins)], wins)

    def route_by_seeds(self, lens, vpaths, mate_lens=None, mvpaths=None):
        n = len(lens)
        lens = np.ascontiguousarray(lens, np.int32); vp = np.ascontiguousarray(vpaths, np.int32).reshape(n, 12)
        ml = np.ascontiguousarray(mate_lens, np.int32) if mate_lens is not None else None
        mv = np.ascontiguousarray(mvpaths, np.int32).reshape(n, 12) if mvpaths is not None else None
        out = np.zeros(n, np.int32)
        _chk(load_library().hu_route_by_seeds(self.dbs[0].h, C.c_int(len(self.windows)), self._w, C.c_int(n), _p(lens, C.c_int32), _p(vp, C.c_int32),
                                              _p(ml, C.c_int32) if ml is not None else None, _p(mv, C.c_int32) if mv is not None else None, _p(out, C.c_int32)))
        return out

    def route_by_region(self, cs_start, cs_end):
        s_ = np.ascontiguousarray(cs_start, np.int32); e_ = np.ascontiguousarray(cs_end, np.int32)
        out = np.zeros(len(s_), np.int32)
        _chk(load_library().hu_route_by_region(C.c_int(len(self.windows)), self._w, C.c_int(len(s_)), _p(s_, C.c_int32), _p(e_, C.c_int32), _p(out, C.c_int32)))
        return out

    def assign(self, reads, vpaths, opts, mates=None, mvpaths=None, first_window=None, ids=None, annos=None):
        """The whole per-read task over the windows: route by seeds, one batch per window, re-route the reads that came back out of their window by
        their region, merge in read order.  Returns (placements [n], alignment records [n]); self.last holds the routing (window per read, reads
        re-routed, reads no window holds) and, with ids, the TSV lines per read.  first_window: a routing to use instead of the seeds' (tests)."""
        n = len(reads)
        vp = np.ascontiguousarray(vpaths, np.int32).reshape(n, 2, 6)
        mvp = np.ascontiguousarray(mvpaths, np.int32).reshape(n, 2, 6) if mvpaths is not None else None
        win = np.asarray(first_window, np.int32) if first_window is not None else \
            self.route_by_seeds([len(r) for r in reads], vp, [len(r) for r in mates] if mates is not None else None, mvp)
        best = np.zeros(n, PLACE_DTYPE); recs = np.zeros(n, ALIGN_DTYPE); lines = [None] * n
        final = win.copy(); rerouted = np.zeros(n, bool)

        def run(w, idx):
            B = Batch(self.dbs[w], len(idx))
            B.set_reads([reads[i] for i in idx], vp[idx], [mates[i] for i in idx] if mates is not None else None, mvp[idx] if mvp is not None else None)
            B.assign(opts)
            b_, r_ = B.placements().copy(), B.alignments(want_align=False)["recs"].copy()
            t_ = None
            if ids is not None:     # a read that is not placed has no line: keyed by the read id at the head of each line
                txt = B.format_tsv([ids[i] for i in idx], None, annos).strip("\n")
                t_ = {l.split("\t", 1)[0]: l for l in txt.split("\n")} if txt else {}
            B.close()
            return b_, r_, t_

        def take(idx, b_, r_, t_):
            best[idx] = b_; recs[idx] = r_
            if t_ is not None:
                for i in idx:
                    lines[i] = t_.get(ids[i])
        for w in range(len(self.dbs)):
            idx = np.nonzero(win == w)[0]
            if len(idx):
                take(idx, *run(w, idx))
        out = np.nonzero(recs["status"] == READ_OUT_OF_WINDOW)[0]
        if len(out):
            w2 = self.route_by_region(recs["cs_start"][out], recs["cs_end"][out])
            for w in range(len(self.dbs)):
                idx = out[(w2 == w) & (win[out] != w)]
                if len(idx):
                    take(idx, *run(w, idx)); final[idx] = w; rerouted[idx] = True
        self.last = dict(first_window=win, window=final, rerouted=rerouted, unplaceable=(recs["status"] == READ_OUT_OF_WINDOW), lines=lines)
        return best, recs

    def close(self):
        for d in self.dbs:
            d.close()


class Batch:
    """One batch of reads in flight on one HIP stream (hu_batch)."""

    def __init__(self, db: Database, max_reads: int):
        self.db = db
        self.h = C.c_void_p()
        self.n = 0
        _chk(load_library().hu_batch_create(db.h, C.c_int(max_reads), C.byref(self.h)))

    # ---- inputs
    def set_reads(self, reads, vpaths, mates=None, mvpaths=None):
        n = len(reads)
        self.n = n
        cat = "".join(reads).encode("latin1")
        offs = np.zeros(n + 1, np.int64); offs[1:] = np.cumsum([len(r) for r in reads])
        vp = np.ascontiguousarray(vpaths, np.int32).reshape(n, 2, 6) if vpaths is not None else None
        if mates is not None:
            mcat = "".join(mates).encode("latin1")
            moffs = np.zeros(n + 1, np.int64); moffs[1:] = np.cumsum([len(r) for r in mates])
            mvp = np.ascontiguousarray(mvpaths, np.int32).reshape(n, 2, 6) if mvpaths is not None else None
        _chk(load_library().hu_batch_set_reads(self.h, C.c_int(n), cat, _p(offs, C.c_int64), _p(vp, C.c_int32) if vp is not None else None,
                                               mcat if mates is not None else None, _p(moffs, C.c_int64) if mates is not None else None,
                                               _p(mvp, C.c_int32) if mates is not None and mvp is not None else None))

    def set_reads_packed(self, cat, offs, vpaths):
        """SE reads as one uint8 buffer + offsets [n + 1] (relative to the buffer's start): no per-read Python objects"""
        offs = np.ascontiguousarray(offs, np.int64); n = len(offs) - 1
        self.n = n
        vp = np.ascontiguousarray(vpaths, np.int32).reshape(n, 2, 6)
        _chk(load_library().hu_batch_set_reads(self.h, C.c_int(n), cat.ctypes.data_as(C.c_char_p), _p(offs, C.c_int64), _p(vp, C.c_int32), None, None, None))

    def format_tsv_bytes(self, id_array, desc_array=None, anno_array=None) -> int:
        """formats the batch's assignment lines (hu_batch_format_tsv_ptr) from prepared (c_char_p * n) arrays and returns their length;
        the text stays in the batch's buffer"""
        lib = load_library()
        lib.hu_batch_format_tsv_ptr.restype = C.c_int64
        txt = C.c_char_p()
        need = lib.hu_batch_format_tsv_ptr(self.h, id_array, desc_array, anno_array, None, C.c_int(0), C.c_int(0), C.byref(txt))
        if need < 0:
            _chk(int(need))
        return int(need)

    def set_aligned(self, codes, start, end):
        codes = np.ascontiguousarray(codes, np.int8)
        n = codes.shape[0]
        self.n = n
        s = np.ascontiguousarray(start, np.int32); e = np.ascontiguousarray(end, np.int32)
        _chk(load_library().hu_batch_set_aligned(self.h, C.c_int(n), _p(codes, C.c_int8), _p(s, C.c_int32), _p(e, C.c_int32)))

    # ---- stages (reference names in the docstrings of include/hmmufotu_amd.h)
    def align(self, opts): _chk(load_library().hu_align_batch(self.h, C.byref(opts)))
    def get_seed(self, opts): _chk(load_library().hu_seed_batch(self.h, C.byref(opts)))
    def estimate_seq(self, opts): _chk(load_library().hu_estimate_batch(self.h, C.byref(opts)))
    def filter_placements(self, opts): _chk(load_library().hu_filter_batch(self.h, C.byref(opts)))
    def place_seq(self, opts): _chk(load_library().hu_place_batch(self.h, C.byref(opts)))
    def calc_q_values(self, opts): _chk(load_library().hu_finish_batch(self.h, C.byref(opts)))
    def assign(self, opts): _chk(load_library().hu_assign_batch(self.h, C.byref(opts)))
    def sync(self): _chk(load_library().hu_batch_sync(self.h))
    def profile(self, enable=True): _chk(load_library().hu_batch_profile(self.h, C.c_int(int(enable))))

    def set_knob(self, name: str, value: int = 1):
        """hu_batch_set_knob: pick an alternative kernel / diagnostic for this batch (tests force every kernel path with it)."""
        _chk(load_library().hu_batch_set_knob(self.h, name.encode(), C.c_int(int(value))))

    def refsort_stats(self):
        """(reads of the last seed stage in the reference's order that the device sort left to the host, whole batch on the host path?)"""
        a = C.c_int32(0); w = C.c_int32(0)
        _chk(load_library().hu_batch_refsort_stats(self.h, C.byref(a), C.byref(w)))
        return int(a.value), bool(w.value)

    def wall(self):
        ms = np.zeros(4)
        _chk(load_library().hu_batch_wall(self.h, _p(ms, C.c_double)))
        return dict(align=ms[0], seed_estimate_filter=ms[1], place=ms[2], finish=ms[3])

    def timings(self):
        ms = np.zeros(8, np.float32)
        _chk(load_library().hu_batch_timings(self.h, _p(ms, C.c_float)))
        return {T_NAMES[i]: float(ms[i]) for i in range(6)}

    # ---- results
    def alignments(self, want_align=True, want_trace=False, trace_stride=0):
        n, L = self.n, self.db.cs_len
        recs = np.zeros(n, ALIGN_DTYPE)
        rows = np.zeros((n, L), np.uint8) if want_align else None
        tr = np.zeros((n, trace_stride), np.uint8) if want_trace else None
        _chk(load_library().hu_batch_get_alignments(self.h, recs.ctypes.data_as(C.c_void_p),
                                                    rows.ctypes.data_as(C.c_char_p) if rows is not None else None,
                                                    tr.ctypes.data_as(C.c_char_p) if tr is not None else None, C.c_int(trace_stride)))
        out = dict(recs=recs)
        if rows is not None:
            out["align"] = [rows[i].tobytes().decode("latin1") for i in range(n)]
        if tr is not None:
            out["trace"] = [tr[i].tobytes().split(b"\0")[0].decode() for i in range(n)]
        return out

    def codes(self):
        n, L = self.n, self.db.cs_len
        cd = np.zeros((n, L), np.int8); s = np.zeros(n, np.int32); e = np.zeros(n, np.int32)
        _chk(load_library().hu_batch_get_codes(self.h, _p(cd, C.c_int8), _p(s, C.c_int32), _p(e, C.c_int32)))
        return cd, s, e

    def pdist(self, read: int):
        d = np.zeros(self.db.n_nodes, np.int32); N = np.zeros(self.db.n_nodes, np.int32)
        _chk(load_library().hu_batch_get_pdist(self.h, C.c_int(read), _p(d, C.c_int32), _p(N, C.c_int32)))
        return d, N

    def seeds(self):
        n = self.n
        cnt = np.zeros(n, np.int32); ids = np.zeros((n, HU_MAX_SEEDS), np.int32); d = np.zeros_like(ids); N = np.zeros_like(ids)
        _chk(load_library().hu_batch_get_seeds(self.h, _p(cnt, C.c_int32), _p(ids, C.c_int32), _p(d, C.c_int32), _p(N, C.c_int32)))
        return cnt, ids, d, N

    def seeds_strided(self, stride: int, guard: int = 0):
        """hu_batch_get_seeds_strided into [n][stride] buffers followed by `guard` sentinel entries (ABI test)."""
        n = self.n
        cnt = np.zeros(n, np.int32)
        bufs = [np.full(n * stride + guard, -777, np.int32) for _ in range(3)]
        _chk(load_library().hu_batch_get_seeds_strided(self.h, _p(cnt, C.c_int32), *[_p(b, C.c_int32) for b in bufs], C.c_int(stride)))
        return (cnt,) + tuple(bufs)

    def estimates_strided(self, stride: int, guard: int = 0):
        n = self.n
        bufs = [np.full(n * stride + guard, -777.0) for _ in range(3)]
        _chk(load_library().hu_batch_get_estimates_strided(self.h, *[_p(b, C.c_double) for b in bufs], C.c_int(stride)))
        return tuple(bufs)

    def estimates(self):
        n = self.n
        r = np.zeros((n, HU_MAX_SEEDS)); w = np.zeros_like(r); ll = np.zeros_like(r)
        _chk(load_library().hu_batch_get_estimates(self.h, _p(r, C.c_double), _p(w, C.c_double), _p(ll, C.c_double)))
        return r, w, ll

    def candidates(self):
        offs = np.zeros(self.n + 1, np.int64)
        _chk(load_library().hu_batch_get_candidates(self.h, _p(offs, C.c_int64), None, None, None, None, None))
        m = int(offs[-1])
        c = np.zeros(m, np.int32); r = np.zeros(m); w = np.zeros(m); e = np.zeros(m); it = np.zeros(m, np.int32)
        _chk(load_library().hu_batch_get_candidates(self.h, _p(offs, C.c_int64), _p(c, C.c_int32), _p(r, C.c_double), _p(w, C.c_double),
                                                    _p(e, C.c_double), _p(it, C.c_int32)))
        return dict(offs=offs, c_node=c, ratio=r, wnr=w, est_loglik=e, iters=it)

    def placements(self):
        out = np.zeros(self.n, PLACE_DTYPE)
        _chk(load_library().hu_batch_get_placements(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def candidate_places(self):
        """Every candidate's placement record after calc_q_values, in filterPlacements order."""
        offs = np.zeros(self.n + 1, np.int64)
        _chk(load_library().hu_batch_get_candidates(self.h, _p(offs, C.c_int64), None, None, None, None, None))
        out = np.zeros(int(offs[-1]), PLACE_DTYPE)
        _chk(load_library().hu_batch_get_candidate_places(self.h, out.ctypes.data_as(C.c_void_p)))
        return offs, out

    def set_candidates(self, offs, recs, placed=False):
        """hu_batch_set_candidates: the candidates of every read given by the caller (records of PLACE_DTYPE in the order the later stages
        are to see them); placed: they carry placeSeq's results and calc_q_values may follow, else place_seq may follow"""
        offs = np.ascontiguousarray(offs, np.int64); recs = np.ascontiguousarray(recs, PLACE_DTYPE)
        assert len(offs) == self.n + 1 and offs[-1] == len(recs)
        _chk(load_library().hu_batch_set_candidates(self.h, _p(offs, C.c_int64), recs.ctypes.data_as(C.c_void_p), C.c_int(int(placed))))

    def get_seed_given(self, n_seeds, ids, dist_ids=None):
        """Segment form of the seed stage: given node ids, distances over the current regions (src/hmmufotu.cpp:662-665)."""
        n_seeds = np.ascontiguousarray(n_seeds, np.int32); ids = np.ascontiguousarray(ids, np.int32)
        assert ids.ndim == 2 and ids.shape[0] == self.n == len(n_seeds)
        di = None
        if dist_ids is not None:
            di = np.ascontiguousarray(dist_ids, np.int32); assert di.shape == ids.shape
        _chk(load_library().hu_seed_batch_given(self.h, _p(n_seeds, C.c_int32), _p(ids, C.c_int32), _p(di, C.c_int32) if di is not None else None,
                                                C.c_int(ids.shape[1])))

    def check_chimera(self, work: "Batch", opts, num_seg=2, max_chimera_error=None, min_chimera_lod=0.0):
        """-C of the per-read task (src/hmmufotu.cpp:653-691) for a batch that is at least seeded; `work` is a second batch
        on the same database whose contents are overwritten."""
        co = ChimeraOpts(num_seg, 0, opts.max_error / num_seg if max_chimera_error is None else max_chimera_error, min_chimera_lod)
        out = np.zeros(self.n, CHIMERA_DTYPE)
        _chk(load_library().hu_chimera_batch(self.h, work.h, C.byref(opts), C.byref(co), out.ctypes.data_as(C.c_void_p)))
        return out

    def format_tsv(self, ids, descs=None, annos=None) -> str:
        return self.format_tsv_chimera(ids, descs, annos, None, False, 0)

    def format_tsv_chimera(self, ids, descs=None, annos=None, chimera=None, chimera_info=False, which=0) -> str:
        """Assignment lines: which=0 the main file, which=1 --chimera-out, which=2 --align-only (src/hmmufotu.cpp:693-747)."""
        lib = load_library()
        lib.hu_batch_format_tsv_ptr.restype = C.c_int64
        arr = lambda xs: (C.c_char_p * len(xs))(*[x.encode() for x in xs]) if xs is not None else None
        a_ids, a_desc, a_anno = arr(ids), arr(descs), arr(annos)
        cp = None
        if chimera is not None:
            chimera = np.ascontiguousarray(chimera, CHIMERA_DTYPE); assert len(chimera) == self.n
            cp = chimera.ctypes.data_as(C.c_void_p)
        txt = C.c_char_p()
        need = lib.hu_batch_format_tsv_ptr(self.h, a_ids, a_desc, a_anno, cp, C.c_int(int(chimera_info)), C.c_int(which), C.byref(txt))
        if need < 0:
            _chk(int(need))
        return C.string_at(txt, need).decode("latin1") if need else ""

    def format_tsv_copy(self, ids, descs=None, annos=None) -> str:
        """the (buf, cap) form of the ABI: size first, then the copy"""
        lib = load_library()
        lib.hu_batch_format_tsv.restype = C.c_int64
        arr = lambda xs: (C.c_char_p * len(xs))(*[x.encode() for x in xs]) if xs is not None else None
        a_ids, a_desc, a_anno = arr(ids), arr(descs), arr(annos)
        need = lib.hu_batch_format_tsv(self.h, a_ids, a_desc, a_anno, None, C.c_int64(0))
        if need < 0:
            _chk(int(need))
        buf = C.create_string_buffer(int(need) + 1)
        lib.hu_batch_format_tsv(self.h, a_ids, a_desc, a_anno, buf, C.c_int64(need))
        return buf.raw[:need].decode("latin1")

    def close(self):
        if self.h:
            load_library().hu_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
