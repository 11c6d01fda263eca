"""Host seed index at gg_97 scale: build time, resident bytes, lookups/s.  CPU only.
99,322 leaf rows x 7,682 CS columns with ~1,400 residues each (1,400 dense columns at 2 % gaps + sparse columns), rows derived
from one another with 3 % substitutions so that relatives share seeds as real OTU sequences do.
Usage: python profiles/measure_seed_index.py [leaves] [reads]  -> one JSON line"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hmmufotu_amd import engine as E, synth

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 99322
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
L, K = 7682, 1400
rng = np.random.default_rng(97)
match = np.zeros(L, bool); match[np.sort(rng.choice(L, K, replace=False))] = True
base = np.empty((leaves, K), np.int8)
base[0] = rng.integers(0, 4, K)
for i in range(1, leaves):                       # every row: a copy of an earlier row with 3 % substitutions
    src = base[int(rng.integers(max(0, i - 64), i))]
    row = src.copy()
    m = rng.random(K) < 0.03
    row[m] = (row[m] + rng.integers(1, 4, int(m.sum()))) % 4
    base[i] = row
seq = np.full((leaves, L), -2, np.int8)
seq[:, match] = np.where(rng.random((leaves, K)) < 0.02, -2, base)
parent = np.zeros(leaves + 1, np.int32); parent[0] = -1          # a star: node 0 is the root, every other node a leaf
allseq = np.vstack([np.zeros((1, L), np.int8), seq])

class H: pass
h = H(); h.K = K; h.p2cs = np.concatenate([[0], np.nonzero(match)[0] + 1]).astype(np.int32)
t0 = time.time()
ix = E.SeedIndex(parent, allseq, h, 20)
t_build = time.time() - t0
rows = rng.integers(1, leaves + 1, nreads)
reads = []
for r in rows:
    s = allseq[r]; c = np.nonzero(s >= 0)[0]
    a = int(rng.integers(0, len(c) - 260)); b = s[c[a:a + 250]].copy()
    k = rng.integers(0, 250, 5); b[k] = (b[k] + 1) % 4           # 2 % errors: some first seeds miss
    reads.append("".join("ACGT"[x] for x in b))
ix.lookup(reads[:2000], 50, 0)
t0 = time.time()
vp = ix.lookup(reads, 50, 0)
dt = time.time() - t0
found = int((vp[:, 0, 0] > 0).sum()); both = int((vp[:, 1, 0] > 0).sum())
print(json.dumps(dict(leaves=leaves, residues=int((seq >= 0).sum()), indexed_positions=ix.positions, distinct_20mers=ix.size, resident_gb=ix.bytes / 1e9,
                      build_s=round(t_build, 1), reads=nreads, lookup_reads_per_s=round(nreads / dt), host_threads=min(16, os.cpu_count()),
                      reads_with_5p_seed=found, reads_with_two_seeds=both)))
