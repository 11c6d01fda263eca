"""TEST INFRASTRUCTURE: the way the reference's own downstream tools READ an assignment file, restated so that the engine's
TSV output is checked against the wire contract its consumers impose — not against itself.

  readProgInfo            src/util/ProgEnv.cpp:106-134   first line "# <progName> <vN.N.N> ...": name must be HmmUFOtu, version <= v1.5.1
  TSVScanner / header     src/util/TSVScanner.cpp:18-43   comment lines skipped up to the header line; fields by NAME, tab separated
  hmmufotu-sum            src/hmmufotu-sum.cpp:369-400    id, CS_start, CS_end, alignment, taxon_id, Q_taxon -> per-OTU read counts and
                                                          per-column base / gap frequencies
  hmmufotu-jplace         src/hmmufotu-jplace.cpp:229-249 + branch_id ("%d->%d"), branch_ratio, anno_dist, loglik, Q_placement
atoi / atol / atof semantics are C's: leading number, 0 on garbage, "nan" -> NaN (so `q >= minQ` is false for an unplaced read)."""
import math
import re

PROG_NAME = "HmmUFOtu"
PROG_VER = (1, 5, 1)


def c_atol(s):
    m = re.match(r"\s*([+-]?\d+)", s)
    return int(m.group(1)) if m else 0


def c_atof(s):
    m = re.match(r"\s*([+-]?(?:nan|inf(?:inity)?|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?))", s, re.I)
    return float(m.group(1)) if m else 0.0


def read_prog_info(line):
    """sscanf(header, "# %s %s", pname, ver); progName == pname; progVer >= VersionSequence(ver) ("v%d.%d.%d")"""
    m = re.match(r"#\s*(\S+)\s+(\S+)", line)          # a blank in a scanf format matches any run of white space, an empty one included
    if not m:
        raise ValueError("Unrecognized input file for " + PROG_NAME)
    if m.group(1) != PROG_NAME:
        raise ValueError("Not an valid input file of " + PROG_NAME)
    # VersionSequence::parseString (src/util/VersionSequence.cpp:56-58): sscanf(str, "v%d.%d.%d") into a (0, 0, 0) object — the
    # numbers converted before the first mismatch stay ("v10.2" is 10.2.0, "1.5.1" is 0.0.0); %d takes a sign
    ver = [0, 0, 0]
    rest = m.group(2)
    if rest[:1] == "v":
        rest = rest[1:]
        for k in range(3):
            d = re.match(r"[+-]?\d+", rest)
            if not d:
                break
            ver[k] = int(d.group(0)); rest = rest[d.end():]
            if k < 2:
                if rest[:1] != ".":
                    break
                rest = rest[1:]
    ver = tuple(ver)
    if not PROG_VER >= ver:
        raise ValueError("file written by a newer version")
    return ver


def scan(text):
    """readProgInfo on the first line, then TSVScanner(in, hasHeader=true): records as dicts keyed by the header's names"""
    lines = text.split("\n")
    read_prog_info(lines[0])
    header = None
    recs = []
    for line in lines[1:]:
        if header is None:
            if line == "" or line[0] == "#":
                continue
            header = line.split("\t")
            continue
        if line == "":
            continue
        f = line.split("\t")
        recs.append({h: (f[i] if i < len(f) else "") for i, h in enumerate(header)})
    if header is None:
        raise ValueError("no header line")
    return header, recs


_ENC = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3, "M": 0, "R": 0, "W": 0, "S": 1, "Y": 1, "K": 2, "V": 0, "H": 0, "D": 0, "B": 1, "N": 0}


def sum_otus(samples, cs_len, min_q=0.0):
    """hmmufotu-sum's accumulation (minAlnIden = minHmmIden = 0): {taxon_id: dict(count=[per sample], freq=[4][L], gap=[L], reads=[ids])}"""
    otus = {}
    for s, text in enumerate(samples):
        _, recs = scan(text)
        for r in recs:
            taxon_id = c_atol(r["taxon_id"]); q_taxon = c_atof(r["Q_taxon"])
            aln = r["alignment"]
            assert c_atol(r["CS_start"]) >= 1 and c_atol(r["CS_end"]) <= cs_len
            if taxon_id >= 0 and q_taxon >= min_q:
                o = otus.setdefault(taxon_id, dict(count=[0] * len(samples), freq=[[0] * cs_len for _ in range(4)], gap=[0] * cs_len, reads=[]))
                o["count"][s] += 1
                o["reads"].append(r["id"])
                assert len(aln) == cs_len, "the alignment column must span the whole consensus (hmmufotu-sum indexes aln[j] for j < L)"
                for j in range(cs_len):
                    b = _ENC.get(aln[j].upper(), -1)
                    if b >= 0:
                        o["freq"][b][j] += 1
                    else:
                        o["gap"][j] += 1
    return otus


def jplace_rows(text, blen, min_q=0.0):
    """hmmufotu-jplace's per-record placement: (cNode, pNode, readName, branch length, ratio, loglik, annoDist, q)"""
    _, recs = scan(text)
    out = []
    for r in recs:
        taxon_id = c_atol(r["taxon_id"]); q = c_atof(r["Q_placement"])
        if taxon_id >= 0 and q >= min_q:
            m = re.match(r"(-?\d+)->(-?\d+)", r["branch_id"])
            assert m, "branch_id must scan as %d->%d"
            c, p = int(m.group(1)), int(m.group(2))
            out.append(dict(c=c, p=p, read=r["id"], length=float(blen[c]), ratio=c_atof(r["branch_ratio"]), loglik=c_atof(r["loglik"]),
                            anno_dist=c_atof(r["anno_dist"]), q=q))
            assert not math.isnan(out[-1]["ratio"]) and not math.isnan(out[-1]["loglik"])
    return out
