"""The reference's seed-index file at gg_97 scale, on the GPU box.
A <leaves>-leaf x 7,682-column alignment is evolved on the device, its leaf rows written as FASTA, indexed into a `.csfm` by
oracle/_ref/csfm_ref (the reference's vendored libcds + libdivsufsort under a restated CSFMIndex::build / save: the binary is built
in the container that has /root/reference and travels here), and read back by the product (hu_seed_index_load_csfm).  Reported:
file size, load time, resident bytes, lookups/s, and how the ViterbiAlignPaths compare with the index built from the leaf rows.
Usage: python profiles/measure_csfm.py [leaves] [reads] -> JSON line"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from hmmufotu_amd import synth_gpu, engine as E

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 99322
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
tmp = os.environ.get("HU_CSFM_TMP", "/dev/shm/hu_csfm")
os.makedirs(tmp, exist_ok=True)
log = lambda *a: print("[csfm]", *a, file=sys.stderr, flush=True)
db, up, down = synth_gpu.make_db_gpu(leaves, 7682, "GTR", dg_k=0, seed=97, win=(0, 64), device="cuda:0", log=log)   # the messages are not needed here
leaf = np.nonzero(db.is_leaf)[0]
fa = os.path.join(tmp, "msa.fasta")
t0 = time.time()
lut = np.frombuffer(b"ACGT", np.uint8)
with open(fa, "wb") as f:
    for i in leaf:
        row = db.seq[i]
        s = np.where(row >= 0, lut[np.clip(row, 0, 3)], ord("-")).astype(np.uint8)
        f.write(b">s%d\n" % i); f.write(s.tobytes()); f.write(b"\n")
log("alignment written: %d rows, %.0f MB (%.0fs)" % (len(leaf), os.path.getsize(fa) / 1e6, time.time() - t0))
# reads = windows of leaf sequences (the lookup only sees bases)
rng = np.random.default_rng(3)
rd = []
for i in rng.choice(leaf, size=nreads):
    g = db.seq[i][db.seq[i] >= 0]
    p = int(rng.integers(0, len(g) - 250))
    rd.append(lut[g[p:p + 250]].tobytes().decode())
csfm = os.path.join(tmp, "db.csfm")
t1 = time.time()
subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "csfm_ref"), fa, csfm])
t_write = time.time() - t1
os.remove(fa)
res = dict(leaves=leaves, rows=int(len(leaf)), csfm_bytes=os.path.getsize(csfm), reference_writer_s=round(t_write, 1), host_cpus=os.cpu_count())
t2 = time.time(); a = E.SeedIndex(None, None, db.hmm, 20, csfm=csfm); res["load_csfm_s"] = round(time.time() - t2, 1)
t3 = time.time(); b = E.SeedIndex(db.parent, db.seq, db.hmm, 20); res["build_from_rows_s"] = round(time.time() - t3, 1)
res.update(csfm_index=dict(distinct=a.size, positions=a.positions, bytes=a.bytes), rows_index=dict(distinct=b.size, positions=b.positions, bytes=b.bytes))
for name, ix in (("csfm_index", a), ("rows_index", b)):
    ix.lookup(rd[:2000])
    t = time.time(); v = ix.lookup(rd); dt = time.time() - t
    res[name]["lookups_per_s"] = round(len(rd) / dt); res[name]["reads_with_a_seed"] = int((v[:, 0, 0] > 0).sum())
    res[name + "_vp"] = v
va, vb = res.pop("csfm_index_vp"), res.pop("rows_index_vp")
res["reads"] = len(rd)
res["same_align_paths"] = int((va == vb).all(axis=(1, 2)).sum())
os.remove(csfm)
print(json.dumps(res))
