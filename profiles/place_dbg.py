"""Diagnostic: per-phase s_memtime stamps of the placement kernel on the benchmark workload (HU_PLACE_VAR=97: k_place_w1,
98: k_place_blk).  Usage on the GPU box: python profiles/place_dbg.py [var]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hmmufotu_amd import engine as E, synth, synth_gpu

var = int(sys.argv[1]) if len(sys.argv) > 1 else 97
leaves = int(os.environ.get("HU_BENCH_LEAVES", 99322))
db, up, down = synth_gpu.make_db_gpu(leaves, 7682, "GTR", dg_k=4, seed=97, device="cuda:0", log=lambda *a: None)
reads = synth_gpu.simulate_reads_gpu(db, up, down, 8192, 250, seed=1, amplicon_start=1000, amplicon_cols=1372, device="cuda:0")
vps = np.stack([synth.read_vpaths(db.hmm, r) for r in reads])
md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r)
D = E.Database.from_arrays(db.hmm, db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, db.anno_id, msgs_on_device=True)
B = E.Batch(D, 8192)
B.set_reads([r.seq for r in reads], vps)
opts = E.default_opts()
B.assign(opts)
B.profile(True)
for v in (0, var, 0, var):
    B.set_knob("place_var", v)
    B.place_seq(opts)
    print("place_var", v, "ms", B.timings()["place"], flush=True)
