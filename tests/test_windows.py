"""Column-window sharding (last row): one database held as several column windows, reads routed by their seeds and re-routed
once by their region — against the same database held whole.  CPU part: the plan and the routing rules.  GPU part: identical records."""
import ctypes as C

import numpy as np
import pytest

from conftest import get_db, sim_reads


def test_windows_plan_covers_the_consensus_with_the_overlap_asked_for():
    from hmmufotu_amd import engine as E
    for cs_len, n, ov in ((7682, 2, 3100), (7682, 8, 3100), (50000, 8, 3200), (1000, 1, 0), (2000, 3, 500), (10, 2, 9)):
        w = E.windows_plan(cs_len, n, ov)
        assert len(w) == n and w[0][0] == 0 and w[-1][0] + w[-1][1] == cs_len
        assert all(a >= 0 and a + l <= cs_len and l == w[0][1] for a, l in w)
        for (a0, l0), (a1, l1) in zip(w, w[1:]):
            assert a1 >= a0 and a0 + l0 - a1 >= ov                    # neighbours share at least `overlap` columns (equal starts: the window is the whole consensus)
        # every region of at most overlap + 1 columns lies inside some window
        for lo in range(0, cs_len - ov, max(1, (cs_len - ov) // 97)):
            assert any(a <= lo and lo + ov < a + l for a, l in w), (cs_len, n, ov, lo)
    with pytest.raises(E.EngineError):
        E.windows_plan(100, 4, 100)                                     # windows that would not advance


def test_route_by_region_prefers_the_widest_margin():
    from hmmufotu_amd import engine as E
    wd = E.WindowedDatabase.__new__(E.WindowedDatabase)
    wd.windows = [(0, 1000), (600, 1000), (1200, 800)]; wd.dbs = [None] * 3
    wd._w = (E.Window * 3)(*[E.Window(a, b) for a, b in wd.windows])
    #                 inside 0 only      inside 0 and 1: margin decides      inside 1 and 2                 in no window
    got = wd.route_by_region([1, 101, 650, 1250, 1900, 400], [200, 950, 980, 1550, 2000, 1700])
    assert list(got) == [0, 0, 1, 1, 2, -1]


@pytest.mark.gpu
@pytest.mark.parametrize("paired", [False, True])
def test_two_windows_give_the_records_of_the_whole_database(paired):
    from hmmufotu_amd import engine as E, synth
    if E.device_count() < 1:
        pytest.fail("no gfx950 device: GPU tests must run on the MI355X box (no CPU fallback exists)")
    db = get_db(300, 2000, "GTR", dg_k=4, seed=7)
    if paired:
        rng = np.random.default_rng(9)
        ins = synth.simulate_reads(db, 96, 100000, rng, mean_cols=900, sd_cols=60)       # inserts scattered over the consensus
        fw, mt = zip(*[synth.split_pair(r, 110) for r in ins])
        reads = [r.seq for r in fw]; mates = [r.seq for r in mt]
        vps = np.stack([synth.read_vpaths(db.hmm, r) for r in fw]); mvps = np.stack([synth.read_vpaths(db.hmm, r) for r in mt])
    else:
        rd, vps = sim_reads(db, 160, 120, amplicon=False)                                # uniform window starts
        reads = [r.seq for r in rd]; mates = mvps = None
        vps = vps.copy(); vps[3] = 0; vps[40] = 0; vps[77, 1] = 0                        # reads without a seed go to window 0 first: some come back out of it
    opts = E.default_opts()
    ids = ["r%d" % i for i in range(len(reads))]
    # the whole database
    D = E.Database.from_synth(db); B = E.Batch(D, len(reads))
    B.set_reads(reads, vps, mates, mvps); B.assign(opts)
    want, wrec = B.placements().copy(), B.alignments(want_align=False)["recs"].copy()
    wtxt = {l.split("\t", 1)[0]: l for l in B.format_tsv(ids, None, db.annos).strip("\n").split("\n")}
    B.close(); D.close()
    span = (wrec["cs_end"] - wrec["cs_start"] + 1)[wrec["status"] == 1]
    overlap = int(span.max()) + 8
    W = E.WindowedDatabase.from_synth(db, 2, overlap)
    assert W.windows[0][1] < db.cs_len and sum(d.hbm_bytes for d in W.dbs) < 1.9 * 2 * db.up.nbytes        # each window keeps its own columns only
    got, grec = W.assign(reads, vps, opts, mates, mvps, ids=ids, annos=db.annos)
    info = W.last
    assert not info["unplaceable"].any()
    for k in wrec.dtype.names:
        assert np.array_equal(grec[k], wrec[k]), k                                       # alignment: bit-equal, status included
    for k in want.dtype.names:
        assert np.array_equal(got[k], want[k], equal_nan=True), k                         # placement records: bit-equal
    assert [info["lines"][i] for i in range(len(reads))] == [wtxt.get(ids[i]) for i in range(len(reads))]
    print("two windows %s: first routing %s, re-routed %d" % (W.windows, np.bincount(info["first_window"], minlength=2), int(info["rerouted"].sum())))
    # a deliberately wrong first routing (everything to window 0): the second pass repairs it, same records
    got2, grec2 = W.assign(reads, vps, opts, mates, mvps, first_window=np.zeros(len(reads), np.int32))
    assert W.last["rerouted"].sum() > 0 and not W.last["unplaceable"].any()
    for k in want.dtype.names:
        assert np.array_equal(got2[k], want[k], equal_nan=True), k
    # both windows really serve reads, and the seeds' routing is mostly right the first time
    assert (np.bincount(info["window"], minlength=2) > 0).all()
    assert info["rerouted"].sum() <= max(4, len(reads) // 10)
    # a region no window holds (overlap too small on purpose) stays HU_READ_OUT_OF_WINDOW and the rest of the batch goes on
    W3 = E.WindowedDatabase.from_synth(db, 3, 10)
    got3, grec3 = W3.assign(reads, vps, opts, mates, mvps)
    un = W3.last["unplaceable"]
    assert un.any() and (grec3["status"][un] == E.READ_OUT_OF_WINDOW).all() and (got3["c_node"][un] < 0).all()
    for k in want.dtype.names:
        assert np.array_equal(got3[k][~un], want[k][~un], equal_nan=True), k
    W.close(); W3.close()


@pytest.mark.gpu
def test_cli_with_column_windows_writes_the_lines_of_the_whole_database(tmp_path):
    """hmmufotu-amd --col-windows 2: the database files loaded as two column windows (hu_db_load_window), every batch split by window, re-routed,
    reassembled in read order — the assignment file, the -a alignment file and the --chimera-info columns are those of the run on the whole database"""
    import os, subprocess, sys
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    import make_tsv_golden as M
    db, pre, samples = M.make_inputs(str(tmp_path))
    cli = os.path.join(os.path.dirname(gold), "..", "hmmufotu_amd", "bin", "hmmufotu-amd")
    fa = samples["A"][0]
    body = lambda t: [l for l in t.split("\n") if l and not l.startswith("#")]
    for extra in ([], ["-C", "--chimera-info"], ["--batch", "7", "--inflight", "2"]):
        a1, a2 = str(tmp_path / "w.aln"), str(tmp_path / "p.aln")
        plain = subprocess.run([cli, pre, fa, "-s", "1", "-a", a2] + extra, capture_output=True, text=True, timeout=300)
        assert plain.returncode == 0, plain.stderr
        win = subprocess.run([cli, pre, fa, "-s", "1", "-a", a1, "--col-windows", "2", "--win-overlap", "600", "-v"] + extra, capture_output=True, text=True, timeout=300)
        assert win.returncode == 0, win.stderr
        assert "2 column windows: [0, 650) [50, 700)" in win.stderr and "0 whose region no window holds" in win.stderr, win.stderr
        assert body(win.stdout) == body(plain.stdout) and len(body(plain.stdout)) == 24          # 23 reads placed + the header
        assert open(a1).read() == open(a2).read()
    # windows too narrow for some regions: those reads have no line, the run goes on and says how many
    nar = subprocess.run([cli, pre, fa, "-s", "1", "--col-windows", "3", "--win-overlap", "100", "-v"], capture_output=True, text=True, timeout=300)
    assert nar.returncode == 0 and " 0 whose region no window holds" not in nar.stderr and len(body(nar.stdout)) < 24, nar.stderr
