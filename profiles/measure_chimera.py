import sys, os, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from hmmufotu_amd import engine as E, synth, synth_gpu
dev="cuda:0"; torch.cuda.set_device(0)
NL = int(os.environ.get("NL", "99322"))
t0=time.time()
db, up, down = synth_gpu.make_db_gpu(NL, 7682, "GTR", dg_k=4, seed=97, device=dev, log=lambda *a: None)
reads = synth_gpu.simulate_reads_gpu(db, up, down, 8192, 250, seed=1, amplicon_start=1000, amplicon_cols=1372, device=dev)
vps = np.stack([synth.read_vpaths(db.hmm, r) for r in reads])
md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r)
D = E.Database.from_arrays(db.hmm, db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, db.anno_id, device=0, msgs_on_device=True)
print("db ready %.0f s" % (time.time()-t0), flush=True)
B = E.Batch(D, 8192); W = E.Batch(D, 8192); B.set_reads([r.seq for r in reads], vps)
opts = E.default_opts()
for rep in range(3):
    t=time.time(); B.assign(opts); B.sync(); ta=time.time()-t
    t=time.time(); B.align(opts); B.get_seed(opts); B.sync(); t1=time.time()-t
    t=time.time(); res = B.check_chimera(W, opts, num_seg=int(os.environ.get("NSEG","2"))); t2=time.time()-t
    t=time.time(); B.estimate_seq(opts); B.filter_placements(opts); B.place_seq(opts); B.calc_q_values(opts); t3=time.time()-t
    print("assign %.1f ms | align+seed %.1f, chimera %.1f, est..finish %.1f ms -> %.1f k reads/s with -C" % (ta*1e3, t1*1e3, t2*1e3, t3*1e3, 8.192/(t1+t2+t3)), flush=True)
print("checked", int(res["checked"].sum()), "flagged", int(res["is_chimera"].sum()), "taxa differ", int((res["seg5"]["a_node"] != res["seg3"]["a_node"]).sum()),
      "mean pool", res["n_seg5"].mean(), res["n_seg3"].mean(), "lod!=0", int((res["lod"] != 0).sum()))
