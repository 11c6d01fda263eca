"""HU_SEED_ORDER_LIBSTDCXX (hu_opts.seed_order = 1): the seeds getSeed keeps under the reference's own order — the first max_nseed elements of
std::sort(locs) on dist ALONE (src/HmmUFOtu_main.cpp:139, src/hmmufotu.cpp:646-647), i.e. the tie permutation of libstdc++'s introsort.

CPU part: the product's restatement of that algorithm restricted to the first k places (hu_sort_prefix_libstdcxx, hu_host.cpp) against the
LITERAL std::sort of the oracle (orc_std_sort_prefix) on tie-heavy inputs of every size class.  GPU part: the engine in that mode against the
oracle's TIE_LIBSTDCXX task on config 1, where the two orders give different final branches for 19 of 1,000 reads."""
import ctypes as C

import numpy as np
import pytest


def _prefix(dist, k):
    from hmmufotu_amd import engine as E
    lib = E.load_library()
    dist = np.ascontiguousarray(dist, np.float64)
    out = np.zeros(min(len(dist), k), np.int32)
    rc = lib.hu_sort_prefix_libstdcxx(dist.ctypes.data_as(C.c_void_p), C.c_int64(len(dist)), C.c_int64(k), out.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return out


def _cases():
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 34, 50, 51, 64, 65, 100, 257, 1000, 4099, 30011):
        yield "distinct-%d" % n, rng.permutation(n) / max(n, 1)
        yield "two-values-%d" % n, rng.integers(0, 2, n) * 0.5
        yield "ties-%d" % n, rng.integers(0, max(2, n // 7), n) / 251.0
        yield "all-equal-%d" % n, np.full(n, 0.25)
        yield "ascending-%d" % n, np.sort(rng.integers(0, max(2, n // 3), n) / 250.0)
        yield "descending-%d" % n, np.sort(rng.integers(0, max(2, n // 3), n) / 250.0)[::-1]
        a = np.arange(n) / 250.0
        yield "organ-pipe-%d" % n, np.minimum(a, a[::-1])
        # the shape of a real distance row: d / N with N = the read's bases for the inferred ancestors, a little less for leaves
        N = np.where(rng.random(n) < 0.5, 250, 250 - rng.integers(0, 6, n)); d = np.minimum(N, rng.binomial(250, 0.08, n))
        yield "d-over-N-%d" % n, d / N
    # median-of-3 killer (Musser): drives the partitions towards the depth limit, past which the range is heap-sorted
    for n in (2048, 20000):
        k = n // 2; a = np.zeros(n)
        for i in range(1, k + 1):
            if i % 2:
                a[i - 1] = i; a[i] = k + i
            a[k + i - 1] = 2 * i
        yield "musser-%d" % n, a / n
    n = 200000
    N = np.where(rng.random(n) < 0.5, 250, 250 - rng.integers(0, 6, n)); d = np.minimum(N, rng.binomial(250, 0.08, n))
    yield "gg97-scale-row", d / N
    yield "gg97-scale-few-values", rng.integers(0, 40, n) / 250.0


def test_prefix_of_libstdcxx_sort_against_the_literal_std_sort():
    from oracle import oracle_py as O
    worst = 0
    for name, dist in _cases():
        for k in (1, 10, 50, 64, len(dist)):
            if k > 64 and len(dist) > 5000:
                continue
            got = _prefix(dist, k); want = O.std_sort_prefix(dist, k)
            assert np.array_equal(got, want), (name, k, got[:20], want[:20])
        # and the orders really differ from a stable sort on such inputs (else the test would prove nothing)
        st = np.argsort(dist, kind="stable")[:50]
        worst += int(not np.array_equal(st, O.std_sort_prefix(dist, 50)))
    assert worst > 20


def test_heap_sort_branch_of_introsort():
    """past a partition depth of 2 lg n introsort heap-sorts the range: reached with McIlroy's adversary run against std::sort itself"""
    from oracle import oracle_py as O
    for n in (40, 500, 5000, 60000):
        dist = O.antiqsort(n)
        assert len(np.unique(dist)) > n // 2
        for k in (1, 17, 50, n):
            assert np.array_equal(_prefix(dist, k), O.std_sort_prefix(dist, k)), (n, k)
        # ties on top of it: the adversarial values quantised
        q = np.floor(dist * 97) / 97
        for k in (50, n):
            assert np.array_equal(_prefix(q, k), O.std_sort_prefix(q, k)), (n, k, "quantised")


def test_nan_is_refused():
    from hmmufotu_amd import engine as E
    lib = E.load_library()
    d = np.array([0.1, np.nan, 0.2]); out = np.zeros(3, np.int32)
    assert lib.hu_sort_prefix_libstdcxx(d.ctypes.data_as(C.c_void_p), C.c_int64(3), C.c_int64(3), out.ctypes.data_as(C.c_void_p)) == -1


@pytest.mark.gpu
def test_device_sort_of_64_records_against_the_literal_std_sort():
    """hu_sort_desc64 (hu_kern_rank.h): the device routine behind filterPlacements and the final sort — libstdc++'s std::sort(rbegin, rend, less)
    on <= 64 records — against the literal call in the oracle: every n from 0 to 64, all keys equal (the final sort of the reference: every
    placed loglik ties, SURVEY.md F4 — the pick IS the tie permutation), heavy ties, distinct keys, sorted / reversed input, and McIlroy's
    adversary for the heap-sort branch."""
    from hmmufotu_amd import engine as E
    from oracle import oracle_py as O
    lib = E.load_library()
    if E.device_count() < 1:
        pytest.fail("no gfx950 device")
    rng = np.random.default_rng(23)
    for n in range(0, 65):
        rows = []
        rows.append(np.full(n, -1234.5))
        rows.append(rng.standard_normal(n))
        rows.append(rng.integers(0, 3, n).astype(float))
        rows.append(rng.integers(0, max(2, n // 4), n).astype(float))
        rows.append(np.sort(rng.standard_normal(n))); rows.append(np.sort(rng.standard_normal(n))[::-1].copy())
        rows.append(O.antiqsort(n)); rows.append(O.antiqsort(n)[::-1].copy())           # the reverse iterators see it reversed
        rows.append(np.floor(O.antiqsort(n) * 7))
        for _ in range(8):
            rows.append(rng.integers(0, 5, n) + rng.integers(0, 2, n) * 1e-9)
        K = np.ascontiguousarray(np.stack(rows)) if n else np.zeros((len(rows), 0))
        out = np.zeros(K.shape, np.int32)
        rc = lib.hu_sort_desc_device(C.c_int(0), K.ctypes.data_as(C.c_void_p), C.c_int(len(rows)), C.c_int(n), out.ctypes.data_as(C.c_void_p))
        assert rc == 0, lib.hu_last_error()
        for r in range(len(rows)):
            assert np.array_equal(out[r], O.std_sort_desc(K[r])), (n, r, K[r], out[r], O.std_sort_desc(K[r]))
    # the all-equal permutation really is not the identity beyond 16 records (else the test above proves little)
    assert not np.array_equal(O.std_sort_desc(np.zeros(40)), np.arange(40))


@pytest.mark.gpu
def test_device_sort_prefix_against_the_host_restatement():
    """k_seed_refsort (data-parallel Hoare partitions + a sequential finisher in LDS) on rows of (d, N) pairs of every shape of the CPU test —
    sizes below and above the finisher's range, heavy ties, sorted / reversed / organ-pipe rows, a row shaped like a real distance row at
    gg_97 scale — against hu_sort_prefix_libstdcxx (itself checked against the literal std::sort above): identical ids in identical order,
    or the row is handed back (count -1) when it needs introsort's heap-sort branch."""
    from hmmufotu_amd import engine as E
    lib = E.load_library()
    if E.device_count() < 1:
        pytest.fail("no gfx950 device")
    rng = np.random.default_rng(17)

    def run(dm, Nm, k, p16):
        rows, n = dm.shape
        pairs = np.ascontiguousarray((dm.astype(np.uint32) << 16) | Nm.astype(np.uint32))
        out = np.zeros((rows, k), np.int32); cnt = np.zeros(rows, np.int32)
        rc = lib.hu_sort_prefix_device(C.c_int(0), pairs.ctypes.data_as(C.c_void_p), C.c_int(rows), C.c_int64(n), C.c_int(k), C.c_int(int(p16)),
                                       out.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
        assert rc == 0, lib.hu_last_error()
        return out, cnt
    handed_back = 0
    for n in (1, 2, 17, 40, 300, 512, 513, 600, 1025, 2000, 2240, 2241, 4608, 4609, 5000, 33333, 198642):      # 2,240 / 4,608: the places of a range held in LDS (32- / 16-bit pairs)
        rows = 6 if n > 50000 else 24
        for shape in ("row", "few", "equal", "asc", "desc", "pipe", "wideN"):
            p16 = shape != "wideN"
            if shape == "wideN":
                Nm = rng.integers(300, 600, (rows, n)); dm = np.minimum(Nm, rng.binomial(500, 0.1, (rows, n)))
            else:
                Nm = np.where(rng.random((rows, n)) < 0.5, 250, 250 - rng.integers(0, 6, (rows, n)))
                dm = {"row": lambda: rng.binomial(250, 0.08, (rows, n)), "few": lambda: rng.integers(0, 4, (rows, n)), "equal": lambda: np.full((rows, n), 7),
                      "asc": lambda: np.sort(rng.integers(0, 200, (rows, n)), 1), "desc": lambda: np.sort(rng.integers(0, 200, (rows, n)), 1)[:, ::-1],
                      "pipe": lambda: np.minimum(np.arange(n), np.arange(n)[::-1])[None, :].repeat(rows, 0) % 200}[shape]()
                dm = np.minimum(dm, Nm)
                if shape in ("asc", "desc", "pipe", "equal"):
                    Nm = np.full((rows, n), 250)
            for k in (50, 1, 64):
                got, cnt = run(dm, Nm, k, p16)
                for r in range(rows):
                    if cnt[r] < 0:
                        handed_back += 1; continue
                    want = _prefix(dm[r] / Nm[r], k)
                    assert cnt[r] == len(want) and np.array_equal(got[r, :cnt[r]], want), (n, shape, k, r, got[r, :12], want[:12])
    print("rows handed back to the host path:", handed_back)
    assert handed_back == 0
    # more than 2^19 elements with 32-bit pairs (a SILVA-scale tree under paired reads): the kernel's keys are exact integers, no width limit
    n, rows = 700001, 2
    Nm = rng.integers(300, 600, (rows, n)); dm = np.minimum(Nm, rng.binomial(500, 0.1, (rows, n)))
    got, cnt = run(dm, Nm, 50, False)
    for r in range(rows):
        want = _prefix(dm[r] / Nm[r], 50)
        assert cnt[r] == 50 and np.array_equal(got[r], want), (n, r, got[r, :12], want[:12])
    # the root anywhere in node order (the engine's trees have it at node 0, the rows above behind the last node): level 0 skips its entry, and the
    # aligned vectors of the row are one element off from there on — at the start, inside a vector, on a vector / subtile boundary, at the end
    for n in (40000, 198642):
        rows = 4
        Nm = np.where(rng.random((rows, n)) < 0.5, 250, 250 - rng.integers(0, 6, (rows, n))); dm = np.minimum(Nm, rng.binomial(250, 0.08, (rows, n)))
        pairs = np.ascontiguousarray((dm.astype(np.uint32) << 16) | Nm.astype(np.uint32))
        want = [_prefix(dm[r] / Nm[r], 50) for r in range(rows)]
        for root in (0, 1, 5, 8, 63, 64, 1000, n // 2 + 3, n - 1, n):
            for p16 in (True, False):
                out = np.zeros((rows, 50), np.int32); cnt = np.zeros(rows, np.int32)
                rc = lib.hu_sort_prefix_device_at(C.c_int(0), pairs.ctypes.data_as(C.c_void_p), C.c_int(rows), C.c_int64(n), C.c_int(50), C.c_int(int(p16)),
                                                  C.c_int64(root), out.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
                assert rc == 0, lib.hu_last_error()
                for r in range(rows):
                    assert cnt[r] == 50 and np.array_equal(out[r], want[r]), (n, root, p16, r, out[r, :12], want[r][:12])
    # d > N is not a p-distance: refused
    assert lib.hu_sort_prefix_device(C.c_int(0), np.array([(5 << 16) | 3] * 8, np.uint32).ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int64(8), C.c_int(4), C.c_int(0),
                                     np.zeros(4, np.int32).ctypes.data_as(C.c_void_p), np.zeros(1, np.int32).ctypes.data_as(C.c_void_p)) != 0
    # the 16-bit form holds d and N in 8 bits each: a pair beyond that is refused, not cut (it would come back as a NaN row from the host path)
    for bad in ((3 << 16) | 300, (256 << 16) | 300):
        assert lib.hu_sort_prefix_device(C.c_int(0), np.array([bad] * 8, np.uint32).ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int64(8), C.c_int(4), C.c_int(1),
                                         np.zeros(4, np.int32).ctypes.data_as(C.c_void_p), np.zeros(1, np.int32).ctypes.data_as(C.c_void_p)) != 0
        assert b"16-bit form" in lib.hu_last_error()
    # the adversarial input reaches the depth limit: the kernel must hand the row back, not answer wrongly
    from oracle import oracle_py as O
    a = O.antiqsort(60000)
    dm = np.round(a * 60000).astype(np.int64)[None, :]; Nm = np.full_like(dm, 65000)
    got, cnt = run(dm, Nm, 50, False)
    want = _prefix(dm[0] / Nm[0], 50)
    assert cnt[0] == -1 or np.array_equal(got[0], want)


@pytest.mark.gpu
def test_engine_in_reference_seed_order_on_config_1():
    """all 1,000 reads of config 1 (the reference's 70_otus fixture) with hu_opts.seed_order = HU_SEED_ORDER_LIBSTDCXX against the oracle's
    task under TIE_LIBSTDCXX (literal std::sort): the seed lists are identical id by id, in order; candidates and final branches identical
    or a documented near-tie; and the (dist, node id) order differs from it on this fixture (19 final branches), so the mode does something"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_cfg1_golden as G
    from hmmufotu_amd import engine as E
    from oracle import oracle_py as O, parity
    if E.device_count() < 1:
        pytest.fail("no gfx950 device")
    db, reads, vps = G.cfg1_inputs()
    D = E.Database.from_synth(db)
    B = E.Batch(D, len(reads))
    rd = [r.seq for r in reads]
    B.set_reads(rd, vps)
    B.assign(E.default_opts(seed_order=0))
    stable = B.placements().copy()
    opts = E.default_opts(seed_order=1)
    B.assign(opts)
    cnt, ids, sd, sN = B.seeds(); best = B.placements(); cand = B.candidates()
    m = O.Model(db.model.type_id, db.model.pi, db.model.par)
    H = O.Hmm(db.hmm.K, db.hmm.L, db.hmm.EM, db.hmm.EI, db.hmm.T, db.hmm.p2cs, 0)
    T = O.Tree(db.parent, db.blen, db.seq, db.up, db.down, db.height, m, None, db.anno_id)
    p1 = O.pipeline_batch(H, T, rd, vps, mode=1, want_lib=True)
    assert (cnt == p1["seed_cnt"]).all()
    for i in range(len(reads)):
        assert (ids[i, :cnt[i]] == p1["lib_ids"][i, :cnt[i]]).all(), i                 # the reference's list, in the reference's order
    res = O.pipeline_batch(H, T, rd, vps, want_cands=True, mode=2, seeds=(p1["seed_cnt"], p1["lib_ids"]))
    per = []
    for i in range(len(reads)):
        k = int(res["n_cand"][i]); a, b = int(cand["offs"][i]), int(cand["offs"][i + 1])
        per.append(parity.classify_read(res["cand_node"][i, :k], res["cand_est"][i, :k], res["cand_ratio0"][i, :k], cand["c_node"][a:b], db.parent,
                                        pos=int(res["best_pos"][i])))
    tot = parity.summarize(per)
    assert tot["set_differs"] == 0 and tot["swaps_unexplained"] == 0 and tot["best_unexplained"] == 0, tot
    same = best["c_node"] == res["best_nodes"][:, 0]
    assert same.sum() == len(reads) - tot["best_differs"]
    assert (best["a_node"][same] == res["best_nodes"][same, 2]).all()
    ndiff = int((stable["c_node"] != best["c_node"]).sum())
    print("config 1: final branches that differ between the two seed orders:", ndiff, "| near-tie swaps:", tot)
    assert ndiff == 19
    # 70_otus leaves have gaps: some reads have nodes with N = 0 ... none on this fixture (tie report: reads_with_nan_dist == 0)
    B.close(); D.close()


@pytest.mark.gpu
def test_reference_seed_order_paths_agree_and_fall_back_where_the_reference_is_undefined(capfd):
    """The reference-order mode on a mid-size tree: (1) the device kernel and the host restatement (knob refsort_host) give the same seed lists,
    equal to the oracle's literal std::sort; (2) with a height filter (-H) the pair rows are compacted to the nodes that pass it (on the device: k_compact_rows, and on the host path) — same rule; (3) a database
    with partial sequences: reads meet nodes they share no column with (N = 0: dist = 0 / 0), std::sort is undefined on NaN, the oracle falls
    back to (dist, id) with NaN last and so must the engine — the device kernel hands such reads to the host path (trace line)."""
    import copy, re
    from conftest import get_db, oracle_objects, sim_reads
    from hmmufotu_amd import engine as E
    from oracle import oracle_py as O
    if E.device_count() < 1:
        pytest.fail("no gfx950 device")
    db = get_db(2600, 300, "JC69", dg_k=0, seed=11, n_match=200)
    reads, vps = sim_reads(db, 24, 100)
    _, H, T = oracle_objects(db)
    rd = [r.seq for r in reads]

    def engine_lists(dbx, opts, knob=None):
        D = E.Database.from_synth(dbx); B = E.Batch(D, len(rd))
        if knob:
            B.set_knob(*knob)
        B.set_knob("trace", 1)
        B.set_reads(rd, vps); B.align(opts); B.get_seed(opts)
        cnt, ids, sd, sN = B.seeds(); cd, st, en = B.codes()
        B.close(); D.close()
        return cnt, ids, cd, st, en
    opts = E.default_opts(max_nseed=20, seed_order=1)
    capfd.readouterr()
    cnt, ids, cd, st, en = engine_lists(db, opts)
    err = capfd.readouterr().err
    assert re.search(r"k_seed_refsort: 24 reads.* 0 reads left to the host", err), err
    cnt_h, ids_h, _, _, _ = engine_lists(db, opts, ("refsort_host", 1))
    assert "k_seed_refsort" not in capfd.readouterr().err
    assert (cnt == cnt_h).all() and (ids == ids_h).all()
    for i in range(len(rd)):
        oid, _, _, _ = T.get_seed(cd[i], int(st[i]), int(en[i]), tie=1, max_n=20)
        assert cnt[i] == len(oid) and (ids[i, :cnt[i]] == oid).all(), i
    differs = sum(not np.array_equal(ids[i, :cnt[i]], T.get_seed(cd[i], int(st[i]), int(en[i]), tie=0, max_n=20)[0]) for i in range(len(rd)))
    assert differs > 0                                                      # the two orders are not the same thing on this tree
    # (2) height filter
    hmax = float(np.median(db.height))
    opts_h = E.default_opts(max_nseed=20, seed_order=1, max_height=hmax)
    capfd.readouterr()
    cnt2, ids2, _, _, _ = engine_lists(db, opts_h)
    assert re.search(r"k_seed_refsort: 24 reads.* 0 reads left to the host", capfd.readouterr().err)      # on the device too: the rows compacted to the nodes that pass
    cnt2h, ids2h, _, _, _ = engine_lists(db, opts_h, ("refsort_host", 1))
    assert (cnt2 == cnt2h).all() and (ids2 == ids2h).all()
    for i in range(len(rd)):
        oid, _, _, _ = T.get_seed(cd[i], int(st[i]), int(en[i]), max_height=hmax, tie=1, max_n=20)
        assert cnt2[i] == len(oid) and (ids2[i, :cnt2[i]] == oid).all(), i
    # (3) partial sequences: NaN distances
    dbp = copy.copy(db)
    rng = np.random.default_rng(17)
    seq = db.seq.copy()
    for i in rng.choice(db.n_nodes, size=int(db.n_nodes * 0.5), replace=False):
        cut = int(rng.integers(40, db.cs_len - 40))
        if rng.random() < 0.5:
            seq[i, :cut] = -2
        else:
            seq[i, cut:] = -2
    dbp.seq = seq
    _, Hp, Tp = oracle_objects(dbp)
    capfd.readouterr()
    cnt3, ids3, cd3, st3, en3 = engine_lists(dbp, opts)
    err3 = capfd.readouterr().err
    m = re.search(r"k_seed_refsort: 24 reads.* (\d+) reads left to the host", err3)
    assert m and int(m.group(1)) > 0
    # more than a handful of them (a database with partial sequences: nearly every read): the fallback rule runs on the device as well
    assert int(m.group(1)) <= 16 or "take the (dist, node id) selection on the device" in err3, err3
    for i in range(len(rd)):
        oid, _, _, _ = Tp.get_seed(cd3[i], int(st3[i]), int(en3[i]), tie=1, max_n=20)
        assert cnt3[i] == len(oid) and (ids3[i, :cnt3[i]] == oid).all(), i


@pytest.mark.gpu
def test_reference_seed_order_on_a_tree_with_streaming_levels(capfd):
    """A tree of ~24,000 nodes: the device sort runs its streaming levels on the pair row (count pass, right stoppers by rank, fused counts of the
    next level), moves into LDS below ~14 K places, finishes sequentially and traces the survivors back through every level to their nodes —
    the root being node 0, so every aligned vector of the row is one element off.  Against the host restatement on the same pair rows
    (knob refsort_host), all reads, 16-bit and 32-bit pairs, and under a height filter (the rows compacted to the nodes that pass)."""
    import re
    from conftest import get_db, sim_reads
    from hmmufotu_amd import engine as E
    if E.device_count() < 1:
        pytest.fail("no gfx950 device")
    db = get_db(12000, 260, "JC69", dg_k=0, seed=3, n_match=180)
    reads, vps = sim_reads(db, 96, 100)
    rd = [r.seq for r in reads]
    D = E.Database.from_synth(db)
    hmax = float(np.quantile(db.height, 0.9))          # ~21,600 nodes pass: the compacted rows still start with a streaming level
    for wide, maxh in ((0, None), (1, None), (0, hmax)):
        lists = []
        for host in (0, 1, 2):                         # 0: the device sort, level 0 from the scan's stopper masks; 1: the host restatement; 2: the device sort counting level 0 itself
            B = E.Batch(D, len(rd))
            B.set_knob("pairs32", wide); B.set_knob("refsort_host", int(host == 1)); B.set_knob("ref_nofuse", int(host == 2)); B.set_knob("trace", 1)
            opts = E.default_opts(seed_order=1) if maxh is None else E.default_opts(seed_order=1, max_height=maxh)
            capfd.readouterr()
            B.set_reads(rd, vps); B.align(opts); B.get_seed(opts)
            err = capfd.readouterr().err
            if host != 1:
                assert re.search(r"k_seed_refsort: 96 reads.* %s pairs.* 0 reads left to the host" % ("32-bit" if wide else "16-bit"), err), err
                # the scan prepares level 0 whenever the sort runs on the pair rows as they are (a height filter compacts them first)
                assert ("level 0 from the scan" in err) == (host == 0 and maxh is None), err
                assert B.refsort_stats() == (0, False)
            else:
                assert "k_seed_refsort" not in err
            cnt, ids, sd, sN = B.seeds()
            lists.append((cnt.copy(), ids.copy(), sd.copy(), sN.copy()))
            B.close()
        (c0, i0, d0, n0), (c1, i1, d1, n1), (c2, i2, d2, n2) = lists
        assert (c0 == c1).all() and (c0 == 50).all() and (c2 == c0).all()
        assert (i0 == i1).all() and (d0 == d1).all() and (n0 == n1).all()
        assert (i0 == i2).all() and (d0 == d2).all() and (n0 == n2).all()
        if maxh is not None:
            assert (db.height[i0] <= maxh).all()
    D.close()


@pytest.mark.gpu
def test_reference_order_under_a_height_filter_with_more_rows_than_one_grid_dimension():
    """66,000 reads in one batch under -H: the rows compacted to the nodes that pass the filter are walked by a grid whose y dimension ends at 65,535
    (k_compact_rows loops over the rows beyond it).  Device sort against the host restatement on every read."""
    from conftest import get_db, sim_reads
    from hmmufotu_amd import engine as E
    if E.device_count() < 1:
        pytest.fail("no gfx950 device")
    db = get_db(60, 300, "JC69", dg_k=0, seed=5, n_match=200)
    reads, vps = sim_reads(db, 330, 60)
    rd = [r.seq for r in reads] * 200; vp = np.tile(vps, (200, 1, 1))
    assert len(rd) == 66000
    D = E.Database.from_synth(db)
    opts = E.default_opts(seed_order=1, max_nseed=12, max_height=float(np.quantile(db.height, 0.8)))
    lists = []
    for host in (0, 1):
        B = E.Batch(D, len(rd)); B.set_knob("refsort_host", host)
        B.set_reads(rd, vp); B.align(opts); B.get_seed(opts)
        cnt, ids, sd, sN = B.seeds(); lists.append((cnt.copy(), ids[:, :12].copy()))
        B.close()
    assert (lists[0][0] == lists[1][0]).all() and (lists[0][0] > 0).all() and (lists[0][1] == lists[1][1]).all()
    assert (lists[0][1][:330] == lists[0][1][65670:]).all()                       # the rows past 65,535 are the rows of the same reads
    D.close()
