#!/usr/bin/env python3
"""Generates the assignment-file fixtures of tests/test_tsv_contract.py ON THE GPU BOX: hmmufotu-amd (the product CLI) on a small
synthetic database in the reference's file formats and two samples of reads -> tests/golden/cli_sampleA.tsv, cli_sampleB.tsv,
cli_sampleA_chimera.tsv (-C --chimera-info) and cli_records.npz (the same reads' placement records through the C ABI).
Usage: python tests/golden/make_tsv_golden.py <outdir>"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_inputs(tmp):
    from conftest import get_db
    from hmmufotu_amd import synth
    db = get_db(120, 700, "GTR", dg_k=4)
    pre = os.path.join(tmp, "db")
    synth.write_hmm(db.hmm, pre + ".hmm"); synth.write_ptu(db, pre + ".ptu")
    leaves = np.nonzero(db.is_leaf)[0]
    samples = {}
    for name, seed in (("A", 8), ("B", 9)):
        rng = np.random.default_rng(seed)
        reads = []
        for i in range(24):                                       # leaf substrings with two substitutions
            u = int(rng.choice(leaves)); s = db.seq[u]; c = np.nonzero(s >= 0)[0]
            a = int(rng.integers(0, max(1, len(c) - 108))); c = c[a:a + 104]
            b = s[c].copy(); k = rng.integers(25, 80, size=2); b[k] = (b[k] + 1) % 4
            reads.append("".join("ACGT"[x] for x in b))
        reads[5] = reads[5][:40] + "?" + reads[5][41:]             # an invalid read: no line in the assignment file
        fa = os.path.join(tmp, "sample%s.fasta" % name)
        with open(fa, "w") as f:
            for i, r in enumerate(reads):
                f.write(">%s_read%d sample=%s\n%s\n" % (name, i, name, r))
        samples[name] = (fa, reads)
    return db, pre, samples


if __name__ == "__main__":
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    tmp = os.path.join(out, "_tmp"); os.makedirs(tmp, exist_ok=True)
    db, pre, samples = make_inputs(tmp)
    cli = os.path.join(ROOT, "hmmufotu_amd", "bin", "hmmufotu-amd")
    for name, (fa, reads) in samples.items():
        p = subprocess.run([cli, pre, fa, "-s", "1"], capture_output=True, text=True, check=True)
        open(os.path.join(out, "cli_sample%s.tsv" % name), "w").write(p.stdout)
    p = subprocess.run([cli, pre, samples["A"][0], "-s", "1", "-C", "--chimera-info"], capture_output=True, text=True, check=True)
    open(os.path.join(out, "cli_sampleA_chimera.tsv"), "w").write(p.stdout)
    from hmmufotu_amd import engine as E
    D = E.Database.load(pre + ".hmm", pre + ".ptu")
    ix = E.SeedIndex(db.parent, db.seq, db.hmm, 20)
    rec = {}
    for name, (fa, reads) in samples.items():
        B = E.Batch(D, 32)
        B.set_reads(reads, ix.lookup(reads, 50, 0)); B.assign(E.default_opts())
        rec["best" + name] = B.placements().copy(); rec["aln" + name] = B.alignments(want_align=False)["recs"].copy()
        B.close()
    np.savez_compressed(os.path.join(out, "cli_records.npz"), blen=db.blen, cs_len=db.cs_len, **rec)
    print("fixtures written to", out)
