#!/usr/bin/env python3
"""Headline benchmark: reads placed / s on a gg_97_otus-scale synthetic DB (GTR + dGamma(4), SE 250 bp).

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the whole per-read task (banded Viterbi -> alignment -> seed scan -> top-k ->
estimate -> filter -> place -> q-values) over one batch of reads that is already resident in HBM.
Reads shard across ranks (weak scaling: every rank processes its own batches against its own
replica of the database); the only collective is the final gather of fixed-size result records.
One JSON line is printed by rank 0.

N > 1: one process per GPU.  Under a launcher (torch.distributed.run sets WORLD_SIZE) this process is one
rank; started plainly with --gpus N > 1 it launches the N ranks itself (python -m torch.distributed.run on
127.0.0.1) BEFORE anything touches the GPU, forwards rank 0's JSON line and exits with the launcher's code.
A WORLD_SIZE that disagrees with --gpus is an error, never a silent one-rank run.

Other workloads of BASELINE.json (parity cases, kept out of the headline line): --read-len 150 --dg-k 0 (cfg2),
--paired --read-len 250 (cfg4 shape), --leaves 200000 --paired --read-len 300 (cfg5), --uniform-starts (SURVEY
§8d: read starts uniform over the resident window instead of one amplicon window).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--leaves", type=int, default=int(os.environ.get("HU_BENCH_LEAVES", 99322)))
    ap.add_argument("--cs-len", type=int, default=7682)
    ap.add_argument("--read-len", type=int, default=250)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("HU_BENCH_BATCH", 8192)))
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("HU_BENCH_INFLIGHT", 6)), help="batches in flight per GPU")
    ap.add_argument("--dg-k", type=int, default=4)
    ap.add_argument("--win", type=int, default=int(os.environ.get("HU_BENCH_WIN", 0)), help="message window columns (0 = all)")
    ap.add_argument("--paired", action="store_true", help="paired-end: two mates of --read-len from the ends of the amplicon")
    ap.add_argument("--amplicon-cols", type=int, default=0, help="CS columns of the simulated amplicon (0: from the read length)")
    ap.add_argument("--uniform-starts", action="store_true", help="read starts uniform over the resident window (SURVEY §8d second run)")
    ap.add_argument("--partial-frac", type=float, default=0.0, help="fraction of the leaves that lose a prefix or suffix (partial reference sequences)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (0 = skip, -1 = about 15 s worth)")
    ap.add_argument("--rehearse", action="store_true",
                    help="control-flow rehearsal WITHOUT the engine (CPU, gloo): launch, rendezvous, barrier, gather, max-over-ranks, one JSON line; value is null")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start the N ranks as children (nothing in this process has touched the
    GPU or imported torch), forward rank 0's line, exit non-zero when any rank failed."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["HU_BENCH_SELF_LAUNCHED"] = "1"
    print("[bench] launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        sys.stdout.write(p.stdout)
        print("[bench] launcher exit code %d, %d JSON lines" % (p.returncode, len(lines)), file=sys.stderr, flush=True)
        sys.exit(p.returncode or 1)
    print(lines[0], flush=True)
    sys.exit(0)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("[bench] WORLD_SIZE=%d but --gpus %d: refusing to measure a different rank count than asked for" % (world, args.gpus),
              file=sys.stderr, flush=True)
        sys.exit(2)

    import numpy as np
    import torch
    backend = os.environ.get("HU_BENCH_BACKEND", "gloo" if args.rehearse else "nccl")   # "gloo" + HU_BENCH_SHARE_GPU=1: N > 1 on one GPU
    if os.environ.get("HU_BENCH_SHARE_GPU"):
        local = 0
    dev = "cuda:%d" % local
    if not args.rehearse:
        torch.cuda.set_device(local)
    rccl_ranks = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
            rccl_ranks = dist.get_world_size()
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == "nccl" else "cpu"              # where collective payloads live

    from hmmufotu_amd.shard import gather_records

    def barrier():
        if world > 1:
            dist.barrier()
        if not args.rehearse:
            torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return dt

    metric = "reads placed/sec (whole node), gg_97_otus GTR+dGamma 250bp; HBM GB/s %peak"
    if args.rehearse:
        # the control flow of the N > 1 run with the engine replaced by a sleep: what the CPU test of the launch path runs
        from hmmufotu_amd.engine import PLACE_DTYPE
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.002)
        recs = np.zeros(args.batch, PLACE_DTYPE)
        recs["c_node"] = rank * args.batch + np.arange(args.batch)
        gathered = gather_records(recs, cdev) if world > 1 else recs
        barrier()
        dt = max_over_ranks(time.perf_counter() - t1)
        ok = len(gathered) == world * args.batch and (np.sort(gathered["c_node"]) == np.arange(world * args.batch)).all()
        if rank == 0:
            print(json.dumps(dict(metric=metric, value=None, unit="reads/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                                  ms_per_step=dt / max(1, args.steps) * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                                  dtype="f64", data="control-flow rehearsal: no engine, no GPU, nothing measured",
                                  config=dict(workload="rehearsal", parallelism="read-sharded x%d" % world), backend=backend,
                                  rccl_ranks=rccl_ranks, gathered_records=int(len(gathered)), gather_ok=bool(ok),
                                  self_launched=bool(os.environ.get("HU_BENCH_SELF_LAUNCHED")))), flush=True)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(0 if ok else 1)

    from hmmufotu_amd import engine as E, synth, synth_gpu

    # amplicon window: 250 bp reads over ~1,372 CS columns; PE: an insert of ~1.84 read lengths
    # (460-bp amplicons for 2 x 250, 550-bp for 2 x 300)
    ins_len = int(round(args.read_len * 1.84)) if args.paired else args.read_len
    amp_cols = args.amplicon_cols or int(round(ins_len * args.cs_len / 1400.0))
    amp_start = 1000 if args.cs_len > 4000 else 30
    win = None
    if args.win > 0:
        win = (max(0, amp_start - 200), min(args.cs_len, args.win))
    t0 = time.time()
    db, up, down = synth_gpu.make_db_gpu(args.leaves, args.cs_len, "GTR", dg_k=args.dg_k, seed=97, win=win, device=dev, log=log, partial_frac=args.partial_frac)
    # reads are drawn from the log-space messages BEFORE the engine adopts (and repacks) them
    nb = args.inflight                                  # batches in flight per GPU (one host thread + one HIP stream each)
    all_reads, all_vps, all_mates, all_mvps = [], [], [], []
    for i in range(nb):
        reads = synth_gpu.simulate_reads_gpu(db, up, down, args.batch, 100000 if args.paired else args.read_len, seed=1 + 1000 * rank + i,
                                             amplicon_start=amp_start, amplicon_cols=amp_cols, device=dev, uniform=args.uniform_starts)
        if args.paired:
            fw, mt = zip(*[synth.split_pair(r, args.read_len) for r in reads])
            all_reads.append(list(fw)); all_vps.append(np.stack([synth.read_vpaths(db.hmm, r) for r in fw]))
            all_mates.append(list(mt)); all_mvps.append(np.stack([synth.read_vpaths(db.hmm, r) for r in mt]))
        else:
            all_reads.append(reads); all_vps.append(np.stack([synth.read_vpaths(db.hmm, r) for r in reads]))
    # host copy of the columns the reads of batch 0 touch, for the CPU baseline (oracle), log space
    cpu_win = None
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        rs = all_reads[0] + (all_mates[0] if args.paired else [])
        lo = max(db.win[0], min(r.cs_start for r in rs) - 40)
        hi = min(db.win[0] + db.win[1], max(r.cs_end for r in rs) + 41)
        if (hi - lo) * db.n_nodes * 64 < 40e9:
            cpu_win = (lo, hi, up[:, lo - db.win[0]:hi - db.win[0]].contiguous().cpu().numpy(),
                       down[:, lo - db.win[0]:hi - db.win[0]].contiguous().cpu().numpy())
        else:
            log("CPU baseline skipped: the reads span %d columns, too many for a host copy of the messages" % (hi - lo))
    md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r if db.dg_k > 0 else None)
    D = E.Database.from_arrays(db.hmm, db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, db.anno_id,
                               win_start=db.win[0], win_len=db.win[1] if win else 0, device=local, msgs_on_device=True)
    log("database resident: %.1f GB in HBM, K=%d, nodes=%d, build %.0fs" % (D.hbm_bytes / 1e9, D.K, D.n_nodes, time.time() - t0))
    batches = []
    opts = E.default_opts()
    for i in range(nb):
        B = E.Batch(D, args.batch)
        if args.paired:
            B.set_reads([r.seq for r in all_reads[i]], all_vps[i], [r.seq for r in all_mates[i]], all_mvps[i])
        else:
            B.set_reads([r.seq for r in all_reads[i]], all_vps[i])     # inputs resident in HBM before the timed region
        B.sync()
        batches.append(B)
    log("reads simulated and uploaded: %d batches of %d (%.0fs)" % (nb, args.batch, time.time() - t0))

    # nb batches in flight, each driven by its own host thread on its own HIP stream (the C ABI's
    # threading model): the host stages of one batch overlap the kernels of the others
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(nb)

    timed_ms = [dict() for _ in range(nb)]
    timed_n = [0] * nb

    def run_steps(i, steps):
        out = None
        for _ in range(steps):
            batches[i].assign(opts)
            out = batches[i].placements()
            for k, v in batches[i].timings().items():       # HIP events on this batch's stream, every timed step
                timed_ms[i][k] = timed_ms[i].get(k, 0.0) + v
            timed_n[i] += 1
        return out

    def run(total):
        share = [total // nb + (1 if i < total % nb else 0) for i in range(nb)]
        futs = [pool.submit(run_steps, i, share[i]) for i in range(nb) if share[i]]
        return [f.result() for f in futs]

    for B in batches:
        B.profile(True)
    # setup, not warm-up: every batch object runs once so that its device / pinned host buffers exist (first-use
    # hipMalloc of several GB per batch) before the W untimed warm-up steps and the K timed steps
    list(pool.map(lambda i: run_steps(i, 1), range(nb)))
    run(args.warmup)
    for i in range(nb):
        timed_ms[i].clear(); timed_n[i] = 0
    barrier()
    cpu0 = os.times()
    t1 = time.perf_counter()
    recs = run(args.steps)[-1]
    if world > 1:                                           # the one collective: final result gather over RCCL
        gathered = gather_records(recs, cdev)
    barrier()
    dt_local = time.perf_counter() - t1
    cpu1 = os.times()
    host_cores_busy = ((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / dt_local      # this rank's host threads, averaged over the timed region
    dt = max_over_ranks(dt_local)
    total_reads = args.batch * args.steps * world
    value = total_reads / dt

    # ---- per-kernel device times (HIP events on the batch's stream)
    # (a) inside the timed region: the kernels of the other batches in flight share the GPU (elapsed-with-contention)
    acc = {}
    for i in range(nb):
        for k, v in timed_ms[i].items():
            acc[k] = acc.get(k, 0.0) + v
    acc = {k: v / max(1, sum(timed_n)) for k, v in acc.items()}
    # (b) one batch alone on the GPU: the kernel's own cost, what the roofline fraction is quoted on
    B = batches[0]
    iso = {}
    nprof = 3
    wall = {}
    for _ in range(nprof):
        B.assign(opts)
        for k, v in B.timings().items():
            iso[k] = iso.get(k, 0.0) + v / nprof
        for k, v in B.wall().items():
            wall[k] = wall.get(k, 0.0) + v / nprof
    cd, st, en = B.codes()
    ok = en >= st
    R = float((en[ok] - st[ok] + 1).mean())
    best = B.placements()
    cand = B.candidates()
    C = float(np.diff(cand["offs"]).mean())
    place_iters = dict(outer_mean=float((cand["iters"] & 255).mean()), outer_max=int((cand["iters"] & 255).max()),
                       em_mean=float((cand["iters"] >> 8).mean()), em_max=int((cand["iters"] >> 8).max()))
    S = 50
    peak = 8000.0
    nread = int(ok.sum())
    nseq = nread * (2 if args.paired else 1)
    Rsum = float((en[ok] - st[ok] + 1).sum())
    Wp = args.read_len + 60
    # algorithmic bytes per launch (SURVEY.md §8d per-unit figures x units of one launch)
    alg = dict(viterbi=nseq * (args.read_len + 136.0 * Wp), seed_pdist=(D.n_nodes - 1) * Rsum, seed_topk=4.0 * D.n_nodes * nread,
               estimate=S * 65.0 * Rsum, place=C * 64.0 * Rsum)
    # measured HBM traffic + VALU issue per kernel: from the newest committed PMC summary that was taken on THIS workload
    # (rocprofv3 --pmc passes of this same command, profiles/make_pmc_summary.py); tagged with its source, null otherwise
    stages = ("viterbi", "seed_pdist", "seed_topk", "estimate", "place")
    pmc_prefix = dict(viterbi=("k_viterbi_wave", "k_viterbi_dec2", "k_viterbi_dec", "k_viterbi_lds", "k_viterbi"), seed_pdist=("k_seed_dscan4", "k_seed_dscan", "k_seed_pdist2", "k_seed_pdist"),
                      seed_topk=("k_seed_topk_straight", "k_seed_topk_d", "k_seed_topk"), estimate=("k_estimate_prod", "k_estimate_blk", "k_estimate"), place=("k_place_blk", "k_place_pair", "k_place"))
    workload_key = dict(leaves=args.leaves, cs_len=args.cs_len, read_len=args.read_len, batch=args.batch, dg_k=args.dg_k, paired=bool(args.paired),
                        uniform_starts=bool(args.uniform_starts), win=args.win)
    if args.partial_frac:
        workload_key["partial_frac"] = args.partial_frac
    pmc, pmc_src = {}, None
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted((x for x in os.listdir(pdir) if x.endswith("_pmc_summary.json")), reverse=True):
        try:
            j = json.load(open(os.path.join(pdir, f)))
            if j.get("workload") == workload_key:
                pmc, pmc_src = j["kernels"], "profiles/" + f
                break
        except Exception:
            pass

    def pmc_entry(stage):
        for pre in pmc_prefix[stage]:
            for name in sorted(pmc):
                if name.startswith(pre) and not name[len(pre):len(pre) + 1].isalnum() and name[len(pre):len(pre) + 1] != "_":
                    return name, pmc[name]
        return pmc_prefix[stage][0], {}

    kern = []
    for k in stages:
        name, ent = pmc_entry(k)
        tr = ent.get("hbm_bytes_per_launch")
        e = dict(stage=k, kernel=name, ms_isolated=round(iso[k], 3), ms_in_timed_region=round(acc[k], 3), algorithmic_bytes=alg[k],
                 achieved=alg[k] / (iso[k] * 1e-3) / 1e9, unit="GB/s", frac=alg[k] / (iso[k] * 1e-3) / 1e9 / peak,
                 traffic=tr, hbm_measured_frac=(tr / (iso[k] * 1e-3) / 1e9 / peak if tr else None))
        if ent.get("valu_issue_cycles_per_launch"):       # typed instruction counts x measured issue cycles (profiles/isa_cost.py), over 1,024 SIMDs
            e["valu_issue_frac"] = ent["valu_issue_cycles_per_launch"] / 1024.0 / (iso[k] * 1e-3 * ent.get("clock_hz", 2.4e9))
        kern.append(e)
    dom = max(kern, key=lambda x: x["ms_isolated"])
    roof = dict(bound="hbm", kernel=dom["kernel"], achieved=dom["achieved"], peak=peak, unit="GB/s", frac=dom["frac"], traffic=dom["traffic"],
                ms=dom["ms_isolated"], timing="HIP events on the batch's stream, one batch in flight (the kernel's own cost)",
                ms_in_timed_region=dom["ms_in_timed_region"], frac_in_timed_region=dom["algorithmic_bytes"] / (dom["ms_in_timed_region"] * 1e-3) / 1e9 / peak,
                hbm_measured_frac=dom["hbm_measured_frac"], valu_issue_frac=dom.get("valu_issue_frac"), traffic_source=pmc_src,
                binding_resource="FP64 VALU issue (see DESIGN.md §8); the algorithmic-bytes fraction is reported as the contract asks")
    bytes_per_read = (D.n_nodes - 1) * R + S * 65 * R + C * 64 * R + (2 if args.paired else 1) * (args.read_len + 136 * Wp) + args.cs_len + 128
    step_traffic = sum(k["traffic"] for k in kern) if all(k["traffic"] for k in kern) else None
    path = dict(bytes_per_read=bytes_per_read, achieved=bytes_per_read * value / world / 1e9, unit="GB/s per GPU",
                frac=bytes_per_read * value / world / 1e9 / peak, mean_R=R, mean_candidates=C,
                hbm_measured_bytes_per_step=step_traffic,
                hbm_measured_frac=(step_traffic / (dt / args.steps) / 1e9 / peak if step_traffic else None), traffic_source=pmc_src)

    shape = "%s %d bp" % ("PE 2 x" if args.paired else "SE", args.read_len)
    out = dict(metric=metric, value=value, unit="reads/s" if not args.paired else "pairs/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload="%s synthetic DB (%d nodes x %d CS columns, K=%d), GTR%s, %s %s reads, "
                                    "batch %d reads/step/GPU, %d batches in flight" % ("SILVA-scale" if args.leaves >= 150000 else "gg_97_otus-scale" if args.leaves >= 90000 else "reduced-scale", D.n_nodes, args.cs_len, D.K, "+dGamma(%d)" % args.dg_k if args.dg_k else "",
                                                                                         shape, "uniform-start" if args.uniform_starts else "amplicon", args.batch, nb),
                           db_hbm_gb=D.hbm_bytes / 1e9, message_window_cols=db.win[1], parallelism="read-sharded x%d" % world),
               timed_region="the engine's whole per-read task on reads and seed paths already resident (hu_assign_batch + result fetch per step); the host seed "
                            "lookup, FASTA parsing, the read upload (< 1 KB per read) and the TSV are outside it: the product CLI's end-to-end rate is "
                            "profiles/r02_cli_throughput.json",
               host_cores_busy_per_rank=round(host_cores_busy, 1), host_cpus=os.cpu_count(),
               rccl_ranks=rccl_ranks, backend=backend if world > 1 else None, gathered_records=(int(len(gathered)) if world > 1 else None),
               roofline=roof, roofline_kernels=kern, roofline_path=path,
               kernel_ms={k: round(v, 3) for k, v in acc.items()}, kernel_ms_one_batch_in_flight={k: round(v, 3) for k, v in iso.items()},
               host_wall_ms={k: round(float(v), 2) for k, v in wall.items()}, place_iterations=place_iters)

    # ---- CPU baseline: the oracle (line-faithful port) on this box's host cores, rank 0, N=1 only; doubles as the
    # full-scale parity check: every difference of the final pick must be a documented near-tie (oracle/parity.py)
    if rank == 0 and world == 1 and args.cpu_sample != 0 and cpu_win is not None:
        try:
            from oracle import oracle_py as O, parity
            lo, hi, up_h, down_h = cpu_win
            assert lo <= int(st[ok].min()) and int(en[ok].max()) < hi
            m = O.Model(db.model.type_id, db.model.pi, db.model.par)
            H = O.Hmm(db.hmm.K, db.hmm.L, db.hmm.EM, db.hmm.EI, db.hmm.T, db.hmm.p2cs, 0)
            T = O.Tree(db.parent, db.blen, db.seq, up_h, down_h, db.height, m, db.dg_r if db.dg_k > 0 else None, db.anno_id,
                       win_start=lo, win_len=hi - lo)
            cores = O.max_threads()
            reads = [r.seq for r in all_reads[0]]
            mates = [r.seq for r in all_mates[0]] if args.paired else None
            mv = all_mvps[0] if args.paired else None

            def cpu(n):
                return O.pipeline_batch(H, T, reads[:n], all_vps[0][:n], mates=mates[:n] if mates else None,
                                        mvpaths=mv[:n] if mates else None, threads=cores, want_cands=True)
            n0 = min(len(reads), max(cores, 16))
            tc = time.perf_counter()
            cpu(n0)
            d0 = time.perf_counter() - tc
            ns = args.cpu_sample if args.cpu_sample > 0 else int(min(len(reads), max(n0, 15.0 / max(d0 / n0, 1e-6))))
            tc = time.perf_counter()
            r1 = cpu(ns)
            d1 = time.perf_counter() - tc
            per = []
            for i in range(ns):
                k = int(r1["n_cand"][i]); a, b = int(cand["offs"][i]), int(cand["offs"][i + 1])
                per.append(parity.classify_read(r1["cand_node"][i, :k], r1["cand_est"][i, :k], r1["cand_ratio0"][i, :k],
                                                cand["c_node"][a:b], db.parent, pos=int(r1["best_pos"][i]) if k else None))
            tot = parity.summarize(per)
            agree = float((r1["best_nodes"][:ns, 0] == best["c_node"][:ns]).mean())
            out["cpu_baseline"] = dict(value=ns / d1, unit=out["unit"], cores=cores, kind="port",
                                       sample="first %d reads of batch 0 (same DB, same reads), OpenMP one read per task" % ns,
                                       stage_cpu_sec=dict(zip(["align", "seed", "estimate", "place"], [round(float(x), 2) for x in r1["stage_sec"]])),
                                       best_branch_agreement_with_gpu=agree,
                                       best_branch_diffs=tot["best_differs"], unexplained_best_branch_diffs=tot["best_unexplained"],
                                       candidate_order=dict(swaps_explained_near_tie=tot["swaps_explained"], swaps_unexplained=tot["swaps_unexplained"],
                                                            candidate_set_differs=tot["set_differs"]),
                                       unexplained_detail=[repr(d) for r_ in per for d in r_["detail"]][:20])
        except Exception as ex:                             # the baseline must never sink the measurement
            import traceback
            traceback.print_exc()
            out["cpu_baseline"] = dict(value=None, unit="reads/s", cores=0, kind="port", sample="failed: %r" % (ex,))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
