#!/usr/bin/env python3
"""Golden `.csfm` of the reference's 70_otus alignment + locateFirst answers for a set of seeds.
Made by oracle/_ref/csfm_ref (`make -C oracle ref`): the reference's vendored libcds and libdivsufsort compiled from /root/reference
where they lie, under a restated CSFMIndex build / save / locateFirst (oracle/csfm_ref.cpp) — CSFMIndex.cpp itself needs Eigen3.
Outputs: tests/golden/70_otus.csfm.gz, tests/golden/csfm_70otus_hits.tsv (pattern, csStart, csEnd, occurrences; 1-based, 0 = no hit).
Run here (the container with /root/reference); the GPU box only reads the committed files."""
import gzip, os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(ROOT, "tests", "golden")
rows = []
for l in gzip.open(os.path.join(G, "ref_data", "70_otus.fasta.gz"), "rt"):
    l = l.strip()
    if l.startswith(">"):
        rows.append("")
    elif rows:
        rows[-1] += l
random.seed(7)
pats = []
for i in range(400):
    r = rows[random.randrange(len(rows))].replace("-", "").replace(".", "").upper()
    L = random.choice([12, 16, 20, 20, 20, 24, 31])
    p = random.randrange(0, len(r) - L)
    s = r[p:p + L]
    if i % 10 == 0:                         # a mutated seed: most of these have no hit
        s = s[:L // 2] + ("A" if s[L // 2] != "A" else "C") + s[L // 2 + 1:]
    pats.append(s)
for r in rows[:20]:                         # seeds at the very start and end of sequences (sampled-SA walks that stop at a separator)
    g = r.replace("-", "").replace(".", "").upper()
    pats += [g[:20], g[1:21], g[2:22], g[3:23], g[-20:]]
with tempfile.TemporaryDirectory() as t:
    fa = os.path.join(t, "msa.fasta")
    with open(fa, "w") as f:
        for i, r in enumerate(rows):
            f.write(">s%d\n%s\n" % (i, r))
    open(os.path.join(t, "pats.txt"), "w").write("\n".join(pats) + "\n")
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "csfm_ref"), fa, os.path.join(t, "x.csfm"), os.path.join(t, "pats.txt"), os.path.join(G, "csfm_70otus_hits.tsv")])
    with open(os.path.join(t, "x.csfm"), "rb") as f, gzip.GzipFile(os.path.join(G, "70_otus.csfm.gz"), "wb", mtime=0) as g:
        g.write(f.read())
print("wrote", os.path.getsize(os.path.join(G, "70_otus.csfm.gz")), "bytes;", len(pats), "patterns")

# ---- a small alignment of awkward rows: empty rows, rows of one to three bases, IUPAC symbols, lower case, '.' gaps
random.seed(11)
rows2 = []
for i in range(48):
    r = []
    kind = i % 8
    for j in range(240):
        if kind == 0:
            r.append("-")                                                    # no base at all
        elif kind == 1:
            r.append(random.choice("ACGT") if j in (5 + i, 100, 200)[:1 + i % 3] else ".")   # one to three bases
        else:
            r.append(random.choice("ACGTacgtNRYK") if random.random() < 0.75 else random.choice("-."))
    rows2.append("".join(r))
pats2 = []
for i in range(200):
    r = rows2[random.randrange(len(rows2))]
    g = "".join(c for c in r if c not in "-.").upper()
    if len(g) < 14:
        continue
    p = random.randrange(0, len(g) - 12)
    pats2.append(g[p:p + 12])
with tempfile.TemporaryDirectory() as t:
    fa = os.path.join(t, "msa.fasta")
    with open(fa, "w") as f:
        for i, r in enumerate(rows2):
            f.write(">s%d\n%s\n" % (i, r))
    open(os.path.join(t, "pats.txt"), "w").write("\n".join(pats2) + "\n")
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "csfm_ref"), fa, os.path.join(t, "x.csfm"), os.path.join(t, "pats.txt"), os.path.join(G, "csfm_awkward_hits.tsv")])
    with open(os.path.join(t, "x.csfm"), "rb") as f, gzip.GzipFile(os.path.join(G, "csfm_awkward.csfm.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    with gzip.GzipFile(os.path.join(G, "csfm_awkward.fasta.gz"), "wb", mtime=0) as g:
        g.write(open(fa, "rb").read())
print("awkward alignment:", len(pats2), "patterns")
