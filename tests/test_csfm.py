"""The reference's own seed-index file, <DB>.csfm (no GPU needed).

tests/golden/70_otus.csfm.gz was written by the reference's vendored libcds + libdivsufsort, compiled from /root/reference where they
lie, under a restated CSFMIndex::build / save (oracle/csfm_ref.cpp; CSFMIndex.cpp itself needs Eigen3) from the reference's 70_otus
alignment; tests/golden/csfm_70otus_hits.tsv holds what CSFMIndex::locateFirst (restated over the REAL wavelet tree and RRR bitmaps)
answers for 500 seeds of 12..31 bases (tests/golden/make_csfm_golden.py).  The product reads that file with its own decoder
(hu_seed_index_load_csfm: libcds's BitSequenceRRR and WaveletTreeNoptrs layouts) and must
  * recover every sequence of the alignment and its CS columns,
  * answer every seed with locateFirst's hit and count,
  * build the same ViterbiAlignPaths as the index made from the alignment rows when the first hits coincide."""
import gzip, os
import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rows():
    rows = []
    for l in gzip.open(os.path.join(G, "ref_data", "70_otus.fasta.gz"), "rt"):
        l = l.strip()
        if l.startswith(">"):
            rows.append("")
        elif rows:
            rows[-1] += l
    return [r.upper() for r in rows]


class _Hmm:          # the index needs the profile's column map only (buildAlignPath): every 5th column a match column here
    def __init__(self, cs_len):
        cols = np.arange(3, cs_len, 5)
        self.K = len(cols)
        self.p2cs = np.zeros(self.K + 2, np.int32); self.p2cs[1:self.K + 1] = cols + 1; self.p2cs[self.K + 1] = cs_len + 1


@pytest.fixture(scope="module")
def csfm_path(tmp_path_factory):
    p = tmp_path_factory.mktemp("csfm") / "70_otus.csfm"
    p.write_bytes(gzip.open(os.path.join(G, "70_otus.csfm.gz"), "rb").read())
    return p


def _hits():
    out = []
    for l in open(os.path.join(G, "csfm_70otus_hits.tsv")):
        f = l.rstrip("\n").split("\t")
        out.append((f[0], int(f[1]), int(f[2]), int(f[3])))
    return out


def test_sequences_and_columns_are_recovered(csfm_path):
    from hmmufotu_amd import engine as E
    rows = _rows()
    cs_len = len(rows[0])
    ix = E.SeedIndex(None, None, _Hmm(cs_len), 20, csfm=csfm_path)
    gapfree = [r.replace("-", "").replace(".", "") for r in rows]
    assert ix.positions == sum(max(0, len(g) - 19) for g in gapfree)
    # every 20-mer at a known place: found, and one of its occurrences is that sequence at that offset and CS column
    rng = np.random.default_rng(5)
    code = {"A": "A", "C": "C", "G": "G", "T": "T", "U": "T", "N": "A", "R": "A", "Y": "C", "M": "A", "K": "G", "S": "C", "W": "A", "B": "C", "D": "A", "H": "A", "V": "A"}
    for _ in range(300):
        q = int(rng.integers(len(rows)))
        g = "".join(code[c] for c in gapfree[q])
        p = int(rng.integers(0, len(g) - 20))
        cols = [j for j, c in enumerate(rows[q]) if c not in "-."]
        n, sq, off, col = ix.occurrences(g[p:p + 20])
        assert n >= 1 and (q, p, cols[p]) in set(zip(sq.tolist(), off.tolist(), col.tolist()))


@pytest.mark.parametrize("k", [12, 16, 20, 24, 31])
def test_first_hit_is_locate_first(csfm_path, k):
    from hmmufotu_amd import engine as E
    cs_len = len(_rows()[0])
    ix = E.SeedIndex(None, None, _Hmm(cs_len), k, csfm=csfm_path)
    checked = crossed = 0
    for pat, s, e, n in _hits():
        if len(pat) != k:
            continue
        got = ix.locate_first(pat)
        assert got[2] == n, pat                                   # CSFMIndex::count
        if got[:2] != (s, e):
            # The reference's accessSA walks back from an unsampled row by LF steps (src/CSFMIndex.cpp:251-259).  Separators and the
            # terminator share the symbol 0 and the BWT row of text position 0 also holds 0, so the LF step on 0 is off by one row for
            # part of the rows: a walk that leaves its sequence through the front (hit within the first three bases, position not a
            # multiple of four) lands in another sequence and locateFirst answers with a wrong place.  The golden file records that
            # answer as the real libcds structures give it; the product reports where the seed really is.
            nocc, sq, off, col = ix.occurrences(pat)
            assert off[0] < 4 and col[0] + 1 == got[0], (pat, got, (s, e))
            crossed += 1
        checked += 1
    assert checked >= 40 and crossed <= 60          # only the seeds deliberately taken from the first four bases of sequences can differ


def test_align_paths_equal_the_row_built_index(csfm_path):
    """the same lookups through hu_seed_index_lookup: where the two indexes' first hits are the same occurrence, the ViterbiAlignPaths are"""
    from hmmufotu_amd import engine as E
    rows = _rows()
    cs_len = len(rows[0])
    hmm = _Hmm(cs_len)
    # a star tree whose leaves are the alignment rows in file order
    n = len(rows)
    parent = np.full(n + 1, n, np.int32); parent[n] = -1
    seq = np.full((n + 1, cs_len), -2, np.int8)
    m = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3, "N": 0, "R": 0, "Y": 1, "M": 0, "K": 2, "S": 1, "W": 0, "B": 1, "D": 0, "H": 0, "V": 0}
    for i, r in enumerate(rows):
        seq[i] = [m.get(c, -2) for c in r]
    a = E.SeedIndex(parent, seq, hmm, 20)
    b = E.SeedIndex(None, None, hmm, 20, csfm=csfm_path)
    assert a.positions == b.positions and a.size == b.size
    rng = np.random.default_rng(9)
    reads = []
    for _ in range(200):
        g = rows[int(rng.integers(n))].replace("-", "").replace(".", "")
        p = int(rng.integers(0, len(g) - 150))
        reads.append("".join("ACGT"[m[c]] for c in g[p:p + 150]))
    va, vb = a.lookup(reads), b.lookup(reads)
    same = 0
    for i, r in enumerate(reads):
        for side in range(2):
            if np.array_equal(va[i, side], vb[i, side]):
                same += 1
    assert same >= 2 * len(reads) * 0.9          # repeats whose 32-symbol contexts tie may resolve to different occurrences
    assert (va[:, 0, 0] > 0).all() and (vb[:, 0, 0] > 0).all()


def test_a_damaged_file_is_refused(csfm_path, tmp_path):
    from hmmufotu_amd import engine as E
    raw = bytearray(csfm_path.read_bytes())
    hmm = _Hmm(len(_rows()[0]))
    for cut in (10, 3000, len(raw) // 2, len(raw) - 40):
        p = tmp_path / ("cut%d.csfm" % cut); p.write_bytes(bytes(raw[:cut]))
        with pytest.raises(Exception):
            E.SeedIndex(None, None, hmm, 20, csfm=p)
    bad = bytearray(raw); bad[len(bad) - 2000] ^= 0x5a                 # inside the wavelet tree's last level
    p = tmp_path / "flip.csfm"; p.write_bytes(bytes(bad))
    try:
        ix = E.SeedIndex(None, None, hmm, 20, csfm=p)                  # a flipped offset bit may still decode: then the lookups differ, not crash
        del ix
    except Exception:
        pass


def test_awkward_rows(tmp_path):
    """rows without a base, rows of one to three bases (no sampled suffix-array entry: they cannot hold a seed and are left out), IUPAC
    symbols, lower case, both gap characters — written by the reference's libcds (tests/golden/csfm_awkward.*)"""
    from hmmufotu_amd import engine as E
    p = tmp_path / "awk.csfm"
    p.write_bytes(gzip.open(os.path.join(G, "csfm_awkward.csfm.gz"), "rb").read())
    rows = []
    for l in gzip.open(os.path.join(G, "csfm_awkward.fasta.gz"), "rt"):
        l = l.strip()
        if l.startswith(">"):
            rows.append("")
        elif rows:
            rows[-1] += l
    ix = E.SeedIndex(None, None, _Hmm(len(rows[0])), 12, csfm=p)
    gapfree = [r.replace("-", "").replace(".", "") for r in rows]
    assert ix.positions == sum(max(0, len(g) - 11) for g in gapfree)
    checked = 0
    for l in open(os.path.join(G, "csfm_awkward_hits.tsv")):
        pat, s, e, n = l.rstrip("\n").split("\t")
        got = ix.locate_first(pat)
        assert got[2] == int(n), pat
        if got[:2] != (int(s), int(e)):
            nocc, sq, off, col = ix.occurrences(pat)
            assert off[0] < 4 and col[0] + 1 == got[0], (pat, got, (s, e))      # the reference's accessSA walk left its sequence (see above)
        checked += 1
    assert checked >= 100
