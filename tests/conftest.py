import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


_DB_CACHE = {}


def get_db(n_leaves=120, cs_len=700, model="GTR", dg_k=4, seed=11, **kw):
    """Small synthetic database, memoised per session."""
    from hmmufotu_amd import synth
    key = (n_leaves, cs_len, model, dg_k, seed, tuple(sorted(kw.items())))
    if key not in _DB_CACHE:
        _DB_CACHE[key] = synth.make_db(n_leaves, cs_len, model, dg_k=dg_k, seed=seed, **kw)
    return _DB_CACHE[key]


def oracle_objects(db, mode=0):
    from oracle import oracle_py as O
    m = O.Model(db.model.type_id, db.model.pi, db.model.par)
    h = db.hmm
    H = O.Hmm(h.K, h.L, h.EM, h.EI, h.T, h.p2cs, mode)
    T = O.Tree(db.parent, db.blen, db.seq, db.up, db.down, db.height, m, db.dg_r if db.dg_k > 0 else None, db.anno_id)
    return m, H, T


def sim_reads(db, n, read_len, seed=1, amplicon=True, cols=None):
    from hmmufotu_amd import synth
    rng = np.random.default_rng(seed)
    if amplicon:
        cols = cols or min(db.cs_len - 120, int(read_len * db.cs_len / max(db.hmm.K, 1) * 1.05))
        reads = synth.simulate_reads(db, n, read_len, rng, amplicon_start=50, amplicon_cols=cols, jitter=20)
    else:
        reads = synth.simulate_reads(db, n, read_len, rng, mean_cols=min(500, db.cs_len // 2), sd_cols=30)
    vps = np.stack([synth.read_vpaths(db.hmm, r) for r in reads])
    return reads, vps


@pytest.fixture(scope="session")
def small_db():
    return get_db()
