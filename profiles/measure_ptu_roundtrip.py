"""The reference's database file at gg_97 scale, written and read by the library: a 198,643-node x
7,682-column database is evaluated on the device (hu_tree_evaluate), written to a .ptu from HBM (hu_ptu_write, 98 GB), loaded
back (hu_db_load: file -> device edge by edge), and one batch of reads is placed against both copies.
Usage (GPU box): python profiles/measure_ptu_roundtrip.py [leaves] -> JSON line.  Needs ~100 GB in /dev/shm."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from hmmufotu_amd import engine as E, synth, synth_gpu

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 99322
tmp = os.environ.get("HU_PTU_TMP", "/dev/shm/hu_ptu"); os.makedirs(tmp, exist_ok=True)
log = lambda *a: print("[ptu]", *a, file=sys.stderr, flush=True)
db, up, down = synth_gpu.make_db_gpu(leaves, 7682, "GTR", dg_k=4, seed=97, device="cuda:0", log=log)
reads = synth_gpu.simulate_reads_gpu(db, up, down, 2048, 250, seed=1, amplicon_start=1000, amplicon_cols=1372, device="cuda:0")
vps = np.stack([synth.read_vpaths(db.hmm, r) for r in reads])
md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r)
pre = os.path.join(tmp, "db")
synth.write_hmm(db.hmm, pre + ".hmm")
t0 = time.time()
E.write_ptu(pre + ".ptu", db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, model_text=db.model.text,
            dg_alpha=db.dg_alpha, dg_breaks=db.dg_b, msgs_on_device=True)
t_write = time.time() - t0
size = os.path.getsize(pre + ".ptu")
log("written %.1f GB in %.0f s" % (size / 1e9, t_write))


def place(D):
    B = E.Batch(D, len(reads)); B.set_reads([r.seq for r in reads], vps); B.assign(E.default_opts())
    out = (B.placements().copy(), B.alignments(want_align=False)["recs"].copy()); B.close()
    return out


D1 = E.Database.from_arrays(db.hmm, db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, db.anno_id, msgs_on_device=True)
p1, a1 = place(D1)
D1.close(); del up, down; torch.cuda.empty_cache()
t0 = time.time()
D2 = E.Database.load(pre + ".hmm", pre + ".ptu")
t_load = time.time() - t0
log("loaded in %.0f s, %.1f GB in HBM" % (t_load, D2.hbm_bytes / 1e9))
p2, a2 = place(D2)
same = all(np.array_equal(p1[k], p2[k], equal_nan=True) for k in ("c_node", "p_node", "a_node", "n_cand", "ratio", "wnr", "loglik", "q_place", "est_loglik")) \
    and np.array_equal(a1["cost"], a2["cost"]) and np.array_equal(a1["cs_start"], a2["cs_start"])
D2.close()
os.remove(pre + ".ptu"); os.remove(pre + ".hmm")
print(json.dumps(dict(nodes=int(db.n_nodes), cs_len=7682, ptu_bytes=size, write_s=round(t_write, 1), write_gbps=round(size / t_write / 1e9, 2),
                      load_s=round(t_load, 1), load_gbps=round(size / t_load / 1e9, 2), reads_placed=len(reads), placements_identical=bool(same))))
