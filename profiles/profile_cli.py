"""Writes a gg_97-scale (or smaller) database + FASTA to /dev/shm for a rocprofv3 run of the CLI binary itself:
   python profiles/profile_cli.py prepare [leaves] [reads]   ->  /dev/shm/hu_cli/{db.hmm,db.ptu,reads.fasta}
   (then: rocprofv3 --kernel-trace --stats ... -- hmmufotu_amd/bin/hmmufotu-amd /dev/shm/hu_cli/db /dev/shm/hu_cli/reads.fasta -s 1 -v --inflight 6 -o /dev/null)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hmmufotu_amd import synth, synth_gpu, engine as E
leaves = int(sys.argv[2]) if len(sys.argv) > 2 else 99322
nreads = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
tmp = "/dev/shm/hu_cli"; os.makedirs(tmp, exist_ok=True)
db, up, down = synth_gpu.make_db_gpu(leaves, 7682, "GTR", dg_k=4, seed=97, device="cuda:0", log=lambda *a: None)
reads = []
for i in range(0, nreads, 65536):
    n = min(65536, nreads - i)
    reads += [r.seq for r in synth_gpu.simulate_reads_gpu(db, up, down, n, 250, seed=1 + i, amplicon_start=1000, amplicon_cols=1372, device="cuda:0")]
names = ["n%d" % i for i in range(db.n_nodes)]; annos = ["k__Synth;p__clade%d" % db.anno_id[i] for i in range(db.n_nodes)]
pre = os.path.join(tmp, "db")
synth.write_hmm(db.hmm, pre + ".hmm")
md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r)
E.write_ptu(pre + ".ptu", db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, names=names, annos=annos, model_text=db.model.text,
            dg_alpha=db.dg_alpha, dg_breaks=db.dg_b, msgs_on_device=True)
with open(os.path.join(tmp, "reads.fasta"), "w") as f:
    for i, r in enumerate(reads):
        f.write(">r%d\n%s\n" % (i, r))
print("prepared", pre, len(reads))
