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
