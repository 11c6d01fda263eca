"""End-to-end throughput of the product CLI (hmmufotu-amd: parse -> seed lookup -> engine -> TSV) on the GPU box.
A <leaves>-leaf x 7,682-column GTR+dGamma(4) database (default 99,322 = gg_97 scale: a 99 GB .ptu) is built on the device and written
in the reference's file formats to /dev/shm straight from HBM (hu_ptu_write), 1 M SE 250 bp reads as FASTA.  Usage: python profiles/measure_cli.py [leaves] [reads] -> JSON line"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from hmmufotu_amd import synth, synth_gpu

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 99322
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
tmp = os.environ.get("HU_CLI_TMP", "/dev/shm/hu_cli")
os.makedirs(tmp, exist_ok=True)
log = lambda *a: print("[cli]", *a, file=sys.stderr, flush=True)
t0 = time.time()
db, up, down = synth_gpu.make_db_gpu(leaves, 7682, "GTR", dg_k=4, seed=97, device="cuda:0", log=log)
reads = []
for i in range(0, nreads, 65536):
    n = min(65536, nreads - i)
    reads += [r.seq for r in synth_gpu.simulate_reads_gpu(db, up, down, n, 250, seed=1 + i, amplicon_start=1000, amplicon_cols=1372, device="cuda:0")]
from hmmufotu_amd import engine as E
names = ["n%d" % i for i in range(db.n_nodes)]; annos = ["k__Synth;p__clade%d" % db.anno_id[i] for i in range(db.n_nodes)]
pre = os.path.join(tmp, "db")
synth.write_hmm(db.hmm, pre + ".hmm")
md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r)
E.write_ptu(pre + ".ptu", db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, names=names, annos=annos, model_text=db.model.text,
            dg_alpha=db.dg_alpha, dg_breaks=db.dg_b, msgs_on_device=True)          # straight from HBM: no host copy of the messages
del up, down; torch.cuda.empty_cache()
fa = os.path.join(tmp, "reads.fasta")
with open(fa, "w") as f:
    for i, r in enumerate(reads):
        f.write(">r%d\n%s\n" % (i, r))
log("inputs written: ptu %.1f GB, fasta %.0f MB (%.0fs)" % (os.path.getsize(pre + ".ptu") / 1e9, os.path.getsize(fa) / 1e6, time.time() - t0))
del db
cli = os.path.join(ROOT, "hmmufotu_amd", "bin", "hmmufotu-amd")
res = {}
flights = os.environ.get("HU_CLI_INFLIGHT", "6").split(",")          # several values: the /dev/null run is repeated for each
runs = ([] if os.environ.get("HU_CLI_SKIP_FILE") else [("shm_file", ["-o", os.path.join(tmp, "out.tsv")], flights[0])]) + [("devnull" if f == flights[0] else "devnull_inflight%s" % f, ["-o", "/dev/null"], f) for f in flights]
# round 4: the reference binary's own seed order (k_seed_refsort) is the CLI's default; the (dist, node id) order and the column-window mode beside it
runs.append(("devnull_stable_seed_order", ["-o", "/dev/null", "--seed-order", "stable"], flights[0]))
if os.environ.get("HU_CLI_WINDOWS", "2") != "0":      # ONE database held as column windows on this one device (each window keeps its own columns' messages only)
    runs.append(("devnull_col_windows", ["-o", "/dev/null", "--col-windows", os.environ.get("HU_CLI_WINDOWS", "2"), "--win-overlap", "3100"], flights[0]))
for name, extra, fl in runs:
    t1 = time.time()
    p = subprocess.run([cli, pre, fa, "-s", "1", "-v", "--inflight", fl] + extra + os.environ.get("HU_CLI_EXTRA", "").split(), capture_output=True, text=True)
    wall = time.time() - t1
    line = [l for l in p.stderr.splitlines() if l.startswith("read loop:")]
    log(name, "rc", p.returncode, "wall %.1fs" % wall, line)
    rate = float(line[0].split(" = ")[1].split()[0]) if line else None
    res[name] = dict(rc=p.returncode, wall_s=round(wall, 1), read_loop_reads_per_s=rate, stderr_tail=p.stderr.splitlines()[-6:])
    if name == "shm_file" and p.returncode == 0:
        res[name]["output_gb"] = os.path.getsize(os.path.join(tmp, "out.tsv")) / 1e9
        os.remove(os.path.join(tmp, "out.tsv"))
for f in (pre + ".hmm", pre + ".ptu", fa):
    os.remove(f)
print(json.dumps(dict(leaves=leaves, reads=nreads, host_cpus=os.cpu_count(), runs=res)))
