"""Host seed index (no GPU needed): the hit set and the hit choice against a brute-force restatement of the
reference's semantics (src/CSFMIndex.cpp:65-147, 262-273: occurrences of the seed in the concatenated gap-free MSA rows;
locateFirst = the first of the suffix-ordered range), and the ViterbiAlignPath built from the hit against the oracle's
buildAlignPath."""
import numpy as np
import pytest

from conftest import get_db, oracle_objects, sim_reads


def _leaf_rows(db):
    rows = []
    for i in np.nonzero(db.is_leaf)[0]:
        c = np.nonzero(db.seq[i] >= 0)[0]
        rows.append((c, "".join("ACGT"[x] for x in db.seq[i][c])))
    return rows


def _brute_hits(rows, kmer):
    """(suffix-order key, sequence, offset, CS column of first / last base) of every occurrence"""
    out = []
    k = len(kmer)
    for q, (cols, s) in enumerate(rows):
        p = s.find(kmer)
        while p >= 0:
            nxt = s[p:p + 32].ljust(32, "A")                      # zero ("A") padded at the end of the sequence
            out.append((nxt, q, p, int(cols[p]), int(cols[p + k - 1])))
            p = s.find(kmer, p + 1)
    return out


def test_hit_set_and_first_hit_semantics():
    from hmmufotu_amd import engine as E
    db = get_db(120, 700, "GTR", dg_k=4)
    rows = _leaf_rows(db)
    ix = E.SeedIndex(db.parent, db.seq, db.hmm, 20)
    assert ix.positions == sum(max(0, len(s) - 19) for _, s in rows)
    rng = np.random.default_rng(3)
    some = 0
    for t in range(200):
        q = int(rng.integers(len(rows))); cols, s = rows[q]
        p = int(rng.integers(0, len(s) - 20))
        kmer = s[p:p + 20]
        hits = _brute_hits(rows, kmer)
        n, sq, off, col = ix.occurrences(kmer)
        assert n == len(hits) >= 1
        hits.sort(key=lambda h: (h[0], ))                         # suffix order to depth 32 (ties: index keeps text order)
        assert sorted(zip(sq.tolist(), off.tolist())) == sorted((h[1], h[2]) for h in hits)
        assert [h[0] for h in hits] == sorted(h[0] for h in hits)
        first = min(hits, key=lambda h: (h[0], h[1], h[2]))
        assert (int(sq[0]), int(off[0]), int(col[0])) == (first[1], first[2], first[3])
        some += n > 1
    assert some > 20                                              # related leaves share seeds: the choice matters
    assert ix.occurrences("ACGT" * 5)[0] == len(_brute_hits(rows, "ACGT" * 5))


def test_lookup_builds_the_path_of_the_first_hit():
    from hmmufotu_amd import engine as E
    from oracle import oracle_py as O
    db = get_db(120, 700, "GTR", dg_k=4)
    _, H, _ = oracle_objects(db)
    rows = _leaf_rows(db)
    ix = E.SeedIndex(db.parent, db.seq, db.hmm, 20)
    reads, _ = sim_reads(db, 40, 120)
    seqs = [r.seq for r in reads]
    leaves = np.nonzero(db.is_leaf)[0]
    rng = np.random.default_rng(5)
    for i in range(0, 40, 2):                                      # half of the reads: exact leaf substrings (seeds are found)
        cols, s = rows[int(rng.integers(len(rows)))]
        a = int(rng.integers(0, len(s) - 110)); seqs[i] = s[a:a + 110]
    vp = ix.lookup(seqs, 50, 0)
    found = 0
    for r, read in enumerate(seqs):
        want = np.zeros((2, 6), np.int32); k = 0

        def path(sf):
            hits = _brute_hits(rows, read[sf:sf + 20])
            if not hits:
                return None
            h = min(hits, key=lambda h: (h[0], h[1], h[2]))
            cols, s = rows[h[1]]
            cc = cols[h[2]:h[2] + 20]
            cs = ["-"] * (int(cc[-1]) - int(cc[0]) + 1)
            for j, c in enumerate(cc):
                cs[int(c) - int(cc[0])] = read[sf + j]
            v = H.build_align_path(int(cc[0]) + 1, int(cc[-1]) + 1, "".join(cs), sf + 1, sf + 20)
            return v if (v[0] > 0 and v[0] <= v[1] and v[2] > 0 and v[2] <= v[3] and int(cc[0]) < int(cc[-1])) else None
        n = len(read); region = min(50, n)
        for sf in range(0, region - 20 + 1):
            v = path(sf)
            if v is not None:
                want[k] = v; k += 1; break
        if k == 0 or n >= 2 * region:
            st = n - 1
            while st - 19 >= n - region and st - 19 >= 0:
                v = path(st - 19)
                if v is not None:
                    want[k] = v; k += 1; break
                st -= 1
        assert (vp[r] == want).all(), (r, vp[r], want)
        found += k > 0
    assert found >= 20


def test_random_hit_choice_is_a_member_of_the_hit_set_and_repeats():
    """hmmufotu -S <seed> (CSFMIndex::locateOne, src/CSFMIndex.cpp:121-147): a seed takes one of its occurrences at random.  Every path
    the seeded lookup returns must be the path of SOME occurrence of that seed; the draw depends on (seed, number of the read in the input,
    seed position) only — the same reads in other batches get the same hits; another seed gives other hits; and the first-hit lookup is
    one of the possibilities."""
    from hmmufotu_amd import engine as E
    db = get_db(120, 700, "GTR", dg_k=4)
    _, H, _ = oracle_objects(db)
    rows = _leaf_rows(db)
    ix = E.SeedIndex(db.parent, db.seq, db.hmm, 20)
    rng = np.random.default_rng(11)
    seqs = []
    for i in range(60):                                            # exact leaf substrings: seeds are found, many of them in several leaves
        cols, s = rows[int(rng.integers(len(rows)))]
        a = int(rng.integers(0, len(s) - 110)); seqs.append(s[a:a + 110])

    def paths_of(read, sf):
        out = set()
        for h in _brute_hits(rows, read[sf:sf + 20]):
            cols, s = rows[h[1]]
            cc = cols[h[2]:h[2] + 20]
            cs = ["-"] * (int(cc[-1]) - int(cc[0]) + 1)
            for j, c in enumerate(cc):
                cs[int(c) - int(cc[0])] = read[sf + j]
            out.add(tuple(int(x) for x in H.build_align_path(int(cc[0]) + 1, int(cc[-1]) + 1, "".join(cs), sf + 1, sf + 20)))
        return out
    first = ix.lookup(seqs, 50, 0)
    a = ix.lookup_random(seqs, 7, 0); b = ix.lookup_random(seqs, 7, 0); c = ix.lookup_random(seqs, 8, 0)
    assert (a == b).all() and (a != c).any() and (a != first).any()
    # batching does not matter: the second half alone, numbered from 30
    assert (ix.lookup_random(seqs[30:], 7, 30) == a[30:]).all()
    multi = 0
    for r, read in enumerate(seqs):
        v5 = tuple(int(x) for x in a[r, 0])
        assert v5[0] > 0                                           # an exact leaf substring always has a 5' seed
        sf = v5[2] - 1
        ps = paths_of(read, sf)
        assert v5 in ps and tuple(int(x) for x in first[r, 0]) in paths_of(read, int(first[r, 0, 2]) - 1)
        multi += len(ps) > 1
    assert multi > 10


def test_bad_arguments():
    from hmmufotu_amd import engine as E
    db = get_db(120, 700, "GTR", dg_k=4)
    with pytest.raises(E.EngineError):
        E.SeedIndex(db.parent, db.seq, db.hmm, 8)
    ix = E.SeedIndex(db.parent, db.seq, db.hmm, 15)
    assert (ix.lookup(["ACGTNNNNACGT", "AC"], 50, 0) == 0).all()   # shorter than the seed: no path, not an error
    assert ix.bytes > 0 and ix.size <= ix.positions


def test_lookup_is_the_same_under_any_cpu_budget():
    """the helper threads of every host pool come out of one budget (hu_cpu_budget: hardware, affinity mask, cgroup CPU quota; HU_CPU_BUDGET
    overrides): with none left the caller does all the work itself — same paths either way"""
    import os
    import subprocess
    import sys
    code = r'''
import sys, zlib, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import get_db, sim_reads
from hmmufotu_amd import engine as E
db = get_db(120, 700, "GTR", dg_k=4)
ix = E.SeedIndex(db.parent, db.seq, db.hmm, 20)
reads, _ = sim_reads(db, 3000, 150, seed=5)
vp = ix.lookup([r.seq for r in reads], 50, 0)
print(zlib.crc32(np.ascontiguousarray(vp).tobytes()), int(np.asarray(vp).any(axis=(1, 2)).sum()))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for budget in ("1", "3", "64"):
        env = dict(os.environ, HU_CPU_BUDGET=budget)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2] and int(outs[0].split()[1]) > 1000, outs
