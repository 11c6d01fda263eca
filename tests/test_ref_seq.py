"""The oracle's restatements against the REFERENCE'S OWN CODE where that compiles without Eigen3 / Boost (no GPU needed).

oracle/_ref/libref_seq.so (`make -C oracle ref`) is the reference's alphabet, DigitalSeq, PrimarySeq and SeqUtils::pDist compiled from
/root/reference/src where they lie, behind a C shim (oracle/ref_seq_shim.cpp, which holds no reference code).  Pinned here:
  * DigitalSeq(abc, name, str)    == oracle digitize, character by character over the whole IUPAC / gap / junk range
  * SeqUtils::pDist             (the inner function of getSeed) == the oracle's (d, N) counts for every node of a tree
  * PrimarySeq::revcom          == the read-side reverse complement used by the data tooling
  * the head of a `.ptu` (saveProgInfo / loadProgInfo, StringUtils::saveString / loadString, DigitalSeq::save / load) in
    both directions against the product's writer and reader
The library is built in the container that has /root/reference and travels with the tree (git-ignored, not gpurun-ignored)."""
import ctypes as C
import os
import numpy as np
import pytest

from conftest import get_db, oracle_objects

SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_seq.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref/libref_seq.so not built (needs /root/reference: make -C oracle ref)")


def _ref():
    L = C.CDLL(SO)
    L.ref_pdist.restype = C.c_double
    return L


def test_digitize_is_the_reference_digitalseq():
    from oracle import oracle_py as O
    L = _ref()
    rng = np.random.default_rng(1)
    alphabet = "ACGTUNRYMKSWBDHVacgtunrymkswbdhv-._~ *xzXJ01"
    for c in alphabet:                                            # every single character
        out = np.zeros(4, np.int8); n = L.ref_digitize(c.encode(), out.ctypes.data_as(C.POINTER(C.c_int8)), 4)
        mine = O.digitize(c)
        assert n == len(mine) and (n == 0 or out[0] == mine[0]), (c, n, out[:n], mine)
    for _ in range(200):
        s = "".join(rng.choice(list(alphabet), size=int(rng.integers(1, 400))))
        out = np.zeros(len(s) + 1, np.int8)
        n = L.ref_digitize(s.encode(), out.ctypes.data_as(C.POINTER(C.c_int8)), len(out))
        mine = O.digitize(s)
        assert n == len(mine) and np.array_equal(out[:n], mine), s


def test_pdist_is_the_reference_sequtils():
    L = _ref()
    db = get_db(60, 400, "GTR", dg_k=0, seed=4)
    _, H, T = oracle_objects(db)
    rng = np.random.default_rng(2)
    p8 = C.POINTER(C.c_int8)
    for _ in range(12):
        q = db.seq[int(rng.integers(db.n_nodes))].copy()
        flip = rng.random(db.cs_len) < 0.1; q[flip & (q >= 0)] = (q[flip & (q >= 0)] + 1) % 4
        gap = rng.random(db.cs_len) < 0.15; q[gap] = -2
        s = int(rng.integers(0, db.cs_len - 50)); e = int(rng.integers(s, db.cs_len))
        if _ % 4 == 3:
            q[s:e + 1] = -2                                      # no base at all in the region: 0 / 0
        q = np.ascontiguousarray(q, np.int8)
        d, N = T.pdist_all(q, s, e)
        for node in range(db.n_nodes):
            row = np.ascontiguousarray(db.seq[node], np.int8)
            ref = L.ref_pdist(q.ctypes.data_as(p8), row.ctypes.data_as(p8), db.cs_len, s, e)
            if N[node] == 0:
                assert np.isnan(ref)
            else:
                assert ref == d[node] / N[node], (node, s, e)      # the same two integers divided as doubles


def test_revcom_is_the_reference_primaryseq():
    L = _ref()
    from hmmufotu_amd import synth
    rng = np.random.default_rng(3)
    for _ in range(50):
        s = "".join(rng.choice(list("ACGTNRYMKSWBDHV"), size=int(rng.integers(1, 300))))
        out = C.create_string_buffer(len(s) + 2)
        assert L.ref_revcom(s.encode(), out, len(s) + 2) == len(s)
        assert out.value.decode() == synth.revcom(s), s


@pytest.mark.gpu
def test_engine_pairs_against_the_reference_pdist():
    """the HIP seed scan's (d, N) for every node, divided, == SeqUtils::pDist of the reference on the same aligned read (through the C ABI)"""
    from hmmufotu_amd import engine as E
    from conftest import sim_reads
    L = _ref()
    db = get_db(150, 700, "GTR", dg_k=0)
    reads, vps = sim_reads(db, 6, 150)
    D = E.Database.from_synth(db)
    B = E.Batch(D, 8)
    opts = E.default_opts()
    B.set_reads([r.seq for r in reads], vps); B.align(opts); B.get_seed(opts)
    cd, st, en = B.codes()
    p8 = C.POINTER(C.c_int8)
    for i in range(len(reads)):
        d, N = B.pdist(i)
        q = np.ascontiguousarray(cd[i], np.int8)
        for node in range(db.n_nodes):
            row = np.ascontiguousarray(db.seq[node], np.int8)
            ref = L.ref_pdist(q.ctypes.data_as(p8), row.ctypes.data_as(p8), db.cs_len, int(st[i]), int(en[i]))
            assert (np.isnan(ref) and N[node] == 0) or ref == d[node] / N[node]
    B.close(); D.close()


def _ptu_head_api(L):
    L.ref_ptu_head_write.restype = C.c_long
    L.ref_ptu_head_write.argtypes = [C.c_long, C.c_int, C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_long]
    L.ref_ptu_head_read.restype = C.c_long
    L.ref_ptu_head_read.argtypes = [C.c_char_p, C.POINTER(C.c_long), C.POINTER(C.c_int), C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                                    C.c_void_p, C.c_void_p]
    return L


def _small_db():
    from hmmufotu_amd import synth
    db = synth.make_db(24, 160, "GTR", dg_k=4, seed=5)
    db.names[3] = ""; db.annos[5] = ""; db.names[7] = "a name with spaces\tand a tab"       # empty and awkward strings
    db.anno_dist = np.linspace(0.0, 0.3, db.n_nodes)
    return db


def test_ptu_head_written_by_the_product_is_read_by_the_reference_code(tmp_path):
    """SURVEY §8 a16.  The head of a `.ptu` — program name + version, node count, csLen and every node record (id, name, DigitalSeq,
    annotation, annotation distance) — written by the PRODUCT (hu_ptu_write) and read back by the reference's own loadProgInfo /
    StringUtils::loadString / DigitalSeq::load (compiled from /root/reference, oracle/ref_seq_shim.cpp): every field equal, and the
    reference code stops reading exactly where the edge block starts."""
    from hmmufotu_amd import engine as E
    L = _ptu_head_api(_ref())
    db = _small_db()
    n, cs = db.n_nodes, db.cs_len
    p = str(tmp_path / "product.ptu")
    md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r)
    E.write_ptu(p, db.parent, db.blen, db.seq, db.up, db.down, db.height, md, model_text=db.model.text, names=db.names, annos=db.annos,
                anno_dist=db.anno_dist, dg_alpha=db.dg_alpha, dg_breaks=db.dg_b)
    nn, cl = C.c_long(0), C.c_int(0)
    codes = np.full((n, cs), 99, np.int8); ad = np.zeros(n); sl = np.zeros(n, np.int64)
    names = C.create_string_buffer(1 << 16); annos = C.create_string_buffer(1 << 16)
    off = L.ref_ptu_head_read(p.encode(), C.byref(nn), C.byref(cl), codes.ctypes.data, codes.size, names, len(names), annos, len(annos),
                              ad.ctypes.data, sl.ctypes.data)
    assert off > 0, off                                                  # loadProgInfo accepted name and version
    assert (nn.value, cl.value) == (n, cs) and (sl == cs).all()
    assert np.array_equal(codes, db.seq) and np.array_equal(ad, db.anno_dist)
    assert names.value.decode().split("\n")[:-1] == list(db.names) and annos.value.decode().split("\n")[:-1] == list(db.annos)
    raw = open(p, "rb").read()
    assert int.from_bytes(raw[off:off + 8], "little") == 2 * (n - 1)      # the edge count follows the last node record


def test_ptu_head_written_by_the_reference_code_is_read_by_the_product(tmp_path):
    """The other direction: saveProgInfo + StringUtils::saveString + DigitalSeq::save (the reference's code) write the head, the tail
    (edges, root, heights, index, models: raw longs / doubles and the model text, whose format the reference's own data/*.sm files pin)
    is taken from the Python writer; the product's reader (hu_db parse path, host only) must return every field.  The bytes of the two
    heads are also compared: the Python writer, the native writer and the reference code agree byte for byte."""
    from hmmufotu_amd import engine as E, synth
    L = _ptu_head_api(_ref())
    db = _small_db()
    n, cs = db.n_nodes, db.cs_len
    py = str(tmp_path / "py.ptu"); synth.write_ptu(db, py)
    raw = open(py, "rb").read()
    buf = C.create_string_buffer(len(raw))
    names = (C.c_char_p * n)(*[s.encode() for s in db.names]); annos = (C.c_char_p * n)(*[s.encode() for s in db.annos])
    seq = np.ascontiguousarray(db.seq, np.int8); ad = np.ascontiguousarray(db.anno_dist, np.float64)
    k = L.ref_ptu_head_write(n, cs, seq.ctypes.data, names, annos, ad.ctypes.data, buf, len(raw))
    assert k > 0
    head = buf.raw[:k]
    assert raw[:k] == head                                                # byte for byte
    assert int.from_bytes(raw[k:k + 8], "little") == 2 * (n - 1)
    mixed = str(tmp_path / "mixed.ptu")
    open(mixed, "wb").write(head + raw[k:])
    got = E.parse_files(None, mixed)
    assert np.array_equal(got["seq"], db.seq) and np.array_equal(got["parent"], db.parent) and np.array_equal(got["up"], db.up)
    # a node without a sequence (DigitalSeq of length 0): the record shrinks, the product reads the rest unharmed
    seq2 = seq.copy(); seq2[4, :] = -128
    k2 = L.ref_ptu_head_write(n, cs, seq2.ctypes.data, names, annos, ad.ctypes.data, buf, len(raw))
    assert k2 == k - cs
    open(mixed, "wb").write(buf.raw[:k2] + raw[k:])
    got = E.parse_files(None, mixed)
    assert np.array_equal(np.delete(got["seq"], 4, 0), np.delete(db.seq, 4, 0)) and np.array_equal(got["up"], db.up)


def test_assignment_file_header_against_the_reference_readproginfo():
    """SURVEY §8 f4.  The first line of the CLI's TSV (committed sample written by hmmufotu-amd on the GPU box) and the Python restatement
    of the consumers' header check (tests/tsv_consumers.py) against the reference's own readProgInfo / writeProgInfo (src/util/ProgEnv.cpp)."""
    import tsv_consumers as T
    L = _ref()
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    first = open(os.path.join(gold, "cli_sampleA.tsv")).readline()
    assert L.ref_read_prog_info(first.encode()) == 1                                   # hmmufotu-sum / -jplace would read on
    buf = C.create_string_buffer(4096)
    k = L.ref_write_prog_info(b" taxonomy assignment generated by ", buf, len(buf))     # hmmufotu.cpp:590 + argv[0]
    assert k > 0 and first.startswith(buf.value.decode().rstrip("\n"))
    for line in ("# HmmUFOtu v1.5.1 taxonomy assignment generated by hmmufotu", "# HmmUFOtu v1.4.0 x", "# HmmUFOtu v1.5.0", "# HmmUFOtu v1.5.2 x",
                 "# hmmufotu_amd v0.1.0 x", "# HmmUFOtu v1.6.0 x", "# HmmUFOtu v2.0.0 x", "id\tdescription", "# HmmUFOtu", "#HmmUFOtu v1.5.1",
                 "# HmmUFOtu v1.5 x", "# HmmUFOtu 1.5.1 x", "# HmmUFOtu v0.0.0 x", ""):
        try:
            T.read_prog_info(line); mine = 1
        except ValueError:
            mine = 0
        try:
            ref = L.ref_read_prog_info(line.encode())
        except Exception:
            ref = -1
        assert mine == ref, (line, mine, ref)
    rng = np.random.default_rng(4)
    for _ in range(400):                                       # random program names, separators and version strings
        name = rng.choice(["HmmUFOtu", "HmmUFOtu", "hmmufotu", "HmmUFOtu2", "X"])
        ver = rng.choice(["v%d.%d.%d", "v%d.%d.%d", "V%d.%d.%d", "%d.%d.%d", "v%d.%d", "v%d.%d.%d-rc1", "v%d.%d.%d.7"])
        ver = ver % tuple(int(x) for x in rng.integers(0, 12, size=ver.count("%d")))
        line = "#" + str(rng.choice(["", " ", "  ", "\t"])) + name + str(rng.choice([" ", "  ", "\t"])) + ver + str(rng.choice(["", " tail", "\ttail"]))
        try:
            T.read_prog_info(line); mine = 1
        except ValueError:
            mine = 0
        assert mine == L.ref_read_prog_info(line.encode()), line


@pytest.fixture(scope="module")
def reads_driver(tmp_path_factory):
    import shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path_factory.mktemp("rd") / "reads_driver")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I" + os.path.join(root, "hmmufotu_amd", "csrc"), "-o", exe, os.path.join(root, "tests", "san", "reads_driver.cpp"), "-lz", "-ldl"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def _product_reads(exe, fmt, path):
    import subprocess
    r = subprocess.run([exe, fmt, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return [tuple(l.split("\x1f")) for l in r.stdout.split("\n")[:-1]]


def _reference_reads(L, fmt, path):
    L.ref_seqio_read.restype = C.c_long
    buf = C.create_string_buffer(1 << 20)
    n = L.ref_seqio_read(str(path).encode(), fmt.encode(), buf, len(buf))
    if n < 0:
        return n
    recs = [tuple(l.split("\x1f")) for l in buf.value.decode().split("\n")[:-1]]
    assert len(recs) == n
    return recs


def test_read_files_against_the_reference_seqio(tmp_path, reads_driver):
    """The CLI's FASTA / FASTQ reader (hu_reads_io.h, under the sanitizers) and the reference's SeqIO on the same files: ids, descriptions
    and sequences equal (the product upper-cases; the reference upper-cases later, when it encodes).  Then the documented differences:
    where the reference ends the program (a '\\r' or a blank inside a sequence: PrimarySeq throws) or stops reading (a blank line between
    records), the product reads on."""
    L = _ref()
    rng = np.random.default_rng(3)
    seq = lambda n: "".join(rng.choice(list("ACGTacgtNRYKMSWBDHVU"), size=n))
    fa = tmp_path / "a.fasta"
    recs = [("r1", "", seq(70)), ("r2", "a description  with   blanks", seq(150)), ("r3|x:1", "tab\tin desc", seq(1)), ("r4", "", seq(300)), ("r5", "d", "")]
    with open(fa, "w") as f:
        f.write(">r1\n%s\n" % recs[0][2])
        f.write(">r2 a description  with   blanks\n" + "".join(recs[1][2][i:i + 60] + "\n" for i in range(0, 150, 60)))     # wrapped at 60
        f.write(">r3|x:1\t tab\tin desc\n%s\n" % recs[2][2])                                                               # tab + blank after the id
        f.write(">r4   \n" + "".join(recs[3][2][i:i + 7] + "\n" for i in range(0, 300, 7)))                                 # blanks after the id, no description
        f.write(">r5 d\n")                                                                                                 # a record without a sequence, last line
    ref = _reference_reads(L, "fasta", fa)
    assert ref == [(i, d, s) for i, d, s in recs]
    assert _product_reads(reads_driver, "fasta", fa) == [(i, d, s.upper()) for i, d, s in ref]
    fq = tmp_path / "a.fastq"
    qrecs = [("q1", "1:N:0:1", seq(100)), ("q2", "", seq(33)), ("@odd", "x", seq(5))]
    with open(fq, "w") as f:
        for i, d, s in qrecs:
            f.write("@%s%s\n%s\n+%s\n%s\n" % (i, " " + d if d else "", s, i if i == "q1" else "", "I" * len(s)))
    ref = _reference_reads(L, "fastq", fq)
    assert ref == qrecs
    assert _product_reads(reads_driver, "fastq", fq) == [(i, d, s.upper()) for i, d, s in ref]
    # no newline at the end of the file
    nf = tmp_path / "nonl.fasta"; nf.write_text(">x y\nACGT\nAC")
    assert _reference_reads(L, "fasta", nf) == [("x", "y", "ACGTAC")] == _product_reads(reads_driver, "fasta", nf)
    # ---- documented differences ----
    cr = tmp_path / "crlf.fasta"; cr.write_bytes(b">w1 dos file\r\nACGT\r\nAC\r\n>w2\r\nGG\r\n")
    assert _reference_reads(L, "fasta", cr) == -1                                    # PrimarySeq throws on '\r': the reference program ends
    assert _product_reads(reads_driver, "fasta", cr) == [("w1", "dos file", "ACGTAC"), ("w2", "", "GG")]
    bl = tmp_path / "blank.fasta"; bl.write_text(">b1\nACGT\n\n>b2\nGG\n")
    assert _reference_reads(L, "fasta", bl) == [("b1", "", "ACGT"), ("b2", "", "GG")]  # a blank line INSIDE a record is an empty line of sequence
    assert _product_reads(reads_driver, "fasta", bl) == [("b1", "", "ACGT"), ("b2", "", "GG")]
    lead = tmp_path / "lead.fasta"; lead.write_text("\n>c1\nAC\n")
    assert _reference_reads(L, "fasta", lead) == []                                  # hasNext() sees '\n', not '>': nothing is read
    assert _product_reads(reads_driver, "fasta", lead) == [("c1", "", "AC")]
    gz = tmp_path / "a.fasta.gz"
    import gzip
    gz.write_bytes(gzip.compress(fa.read_bytes()))
    assert _product_reads(reads_driver, "fasta", gz) == _product_reads(reads_driver, "fasta", fa)


def test_cli_revcom_is_the_reference_revcom(reads_driver):
    """the mate of a pair is reverse-complemented before it is aligned (src/hmmufotu.cpp:609: revSeqI.nextSeq().revcom())"""
    import subprocess
    L = _ref()
    rng = np.random.default_rng(9)
    for s in ["ACGTUNRYMKSWBDHV", "A", "".join(rng.choice(list("ACGTUNRYMKSWBDHV"), size=301))]:
        out = C.create_string_buffer(len(s) + 8)
        assert L.ref_revcom(s.encode(), out, len(out)) == len(s)
        mine = subprocess.run([reads_driver, "revcom", s], capture_output=True, text=True)
        assert mine.returncode == 0 and mine.stdout.strip() == out.value.decode(), s


def test_read_files_damaged(tmp_path, reads_driver):
    """the reader under the sanitizers on input that is not a read file at all: random bytes, a gzip stream cut in the middle, a FASTQ
    record cut after its second line, lines of 1 MB — it returns what it could read and never touches memory it does not own"""
    import gzip, subprocess
    rng = np.random.default_rng(12)
    junk = tmp_path / "junk.fasta"; junk.write_bytes(rng.integers(0, 256, size=200000, dtype=np.uint8).tobytes())
    cut = tmp_path / "cut.fasta.gz"
    whole = gzip.compress(b"".join(b">r%d d\n%s\n" % (i, b"ACGT" * 60) for i in range(2000)))
    cut.write_bytes(whole[:len(whole) // 2])
    fq = tmp_path / "cut.fastq"; fq.write_text("@a\nACGT\n+\nIIII\n@b\nACG")
    longl = tmp_path / "long.fasta"; longl.write_text(">" + "x" * (1 << 20) + " " + "y" * (1 << 20) + "\n" + "A" * (3 << 20) + "\n>z\nAC\n")
    for fmt, path in (("fasta", junk), ("fastq", junk), ("fasta", cut), ("fastq", fq), ("fasta", longl), ("fasta", tmp_path / "missing")):
        r = subprocess.run([reads_driver, fmt, str(path)], capture_output=True)
        assert r.returncode in (0, 3), (fmt, path, r.stderr[-1500:])      # 3 = could not open
    recs = _product_reads(reads_driver, "fastq", fq)
    assert recs == [("a", "", "ACGT")]                                      # the record cut short is dropped, not half-read
    recs = _product_reads(reads_driver, "fasta", longl)
    assert len(recs) == 2 and len(recs[0][0]) == 1 << 20 and len(recs[0][2]) == 3 << 20 and recs[1] == ("z", "", "AC")
    assert len(_product_reads(reads_driver, "fasta", cut)) > 100            # what the intact part of the stream holds
