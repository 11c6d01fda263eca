#!/usr/bin/env python3
"""Headline benchmark: reads placed / s on a gg_97_otus-scale synthetic DB (GTR + dGamma(4), SE 250 bp).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step is one pass of the whole per-read task (banded Viterbi -> alignment -> seed scan -> top-k ->
estimate -> filter -> place -> q-values) over one batch of reads that is already resident in HBM.
Reads shard across ranks (weak scaling: every rank processes its own batches against its own
replica of the database); the only collective is the final gather of fixed-size result records.
One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--leaves", type=int, default=int(os.environ.get("HU_BENCH_LEAVES", 99322)))
    ap.add_argument("--cs-len", type=int, default=7682)
    ap.add_argument("--read-len", type=int, default=250)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("HU_BENCH_BATCH", 8192)))
    ap.add_argument("--dg-k", type=int, default=4)
    ap.add_argument("--win", type=int, default=int(os.environ.get("HU_BENCH_WIN", 0)), help="message window columns (0 = all)")
    ap.add_argument("--cpu-sample", type=int, default=-1)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    backend = os.environ.get("HU_BENCH_BACKEND", "nccl")     # "gloo" + HU_BENCH_SHARE_GPU=1: rehearsal of the N > 1 path on one GPU
    if os.environ.get("HU_BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == "nccl" else "cpu"              # where collective payloads live

    from hmmufotu_amd import engine as E, synth, synth_gpu
    from hmmufotu_amd.shard import gather_records

    # amplicon window: 250 bp reads over ~1,372 CS columns
    amp_cols = int(round(args.read_len * args.cs_len / 1400.0))
    amp_start = 1000 if args.cs_len > 4000 else 30
    win = None
    if args.win > 0:
        win = (max(0, amp_start - 200), min(args.cs_len, args.win))
    t0 = time.time()
    db, up, down = synth_gpu.make_db_gpu(args.leaves, args.cs_len, "GTR", dg_k=args.dg_k, seed=97, win=win, device=dev, log=log)
    # reads are drawn from the log-space messages BEFORE the engine adopts (and repacks) them
    nb = int(os.environ.get("HU_BENCH_INFLIGHT", 6))    # batches in flight per GPU (one host thread + one HIP stream each)
    all_reads, all_vps = [], []
    for i in range(nb):
        reads = synth_gpu.simulate_reads_gpu(db, up, down, args.batch, args.read_len, seed=1 + 1000 * rank + i,
                                             amplicon_start=amp_start, amplicon_cols=amp_cols, device=dev)
        all_reads.append(reads); all_vps.append(np.stack([synth.read_vpaths(db.hmm, r) for r in reads]))
    # host copy of the amplicon window of the messages for the CPU baseline (oracle), log space
    cpu_win = None
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        lo = max(db.win[0], min(r.cs_start for r in all_reads[0]) - 40)
        hi = min(db.win[0] + db.win[1], max(r.cs_end for r in all_reads[0]) + 41)
        cpu_win = (lo, hi, up[:, lo - db.win[0]:hi - db.win[0]].contiguous().cpu().numpy(),
                   down[:, lo - db.win[0]:hi - db.win[0]].contiguous().cpu().numpy())
    md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r if db.dg_k > 0 else None)
    D = E.Database.from_arrays(db.hmm, db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, db.anno_id,
                               win_start=db.win[0], win_len=db.win[1] if win else 0, device=local, msgs_on_device=True)
    log("database resident: %.1f GB in HBM, K=%d, nodes=%d, build %.0fs" % (D.hbm_bytes / 1e9, D.K, D.n_nodes, time.time() - t0))
    batches = []
    opts = E.default_opts()
    for i in range(nb):
        B = E.Batch(D, args.batch)
        B.set_reads([r.seq for r in all_reads[i]], all_vps[i])     # inputs resident in HBM before the timed region
        B.sync()
        batches.append(B)
    log("reads simulated and uploaded: %d batches of %d (%.0fs)" % (nb, args.batch, time.time() - t0))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # two batches in flight, each driven by its own host thread on its own HIP stream (the C ABI's
    # threading model): the host stages of one batch overlap the kernels of the other
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
    t1 = time.perf_counter()
    recs = run(args.steps)[-1]
    if world > 1:                                           # the one collective: final result gather over RCCL
        gathered = gather_records(recs, cdev)
    barrier()
    dt = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_reads = args.batch * args.steps * world
    value = total_reads / dt

    # ---- per-kernel device times (HIP events on the batch's stream) + roofline of the scan kernel
    # kernel time inside the timed region (two batches in flight: kernels of both streams share the GPU)
    acc = {}
    for i in range(nb):
        for k, v in timed_ms[i].items():
            acc[k] = acc.get(k, 0.0) + v
    acc = {k: v / max(1, sum(timed_n)) for k, v in acc.items()}
    # and with one batch alone on the GPU
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
    Rsum = float((en[ok] - st[ok] + 1).sum())
    Wp = args.read_len + 60
    # algorithmic bytes per launch (SURVEY.md §8d per-unit figures x units of one launch)
    alg = dict(viterbi=nread * (args.read_len + 136.0 * Wp), seed_pdist=(D.n_nodes - 1) * Rsum, seed_topk=4.0 * D.n_nodes * nread,
               estimate=S * 65.0 * Rsum, place=C * 64.0 * Rsum)
    # kernel of each stage as rocprofv3 names it (prefix match: template arguments vary with the read length)
    pmc_prefix = dict(viterbi=("k_viterbi_wave", "k_viterbi_dec2", "k_viterbi_dec", "k_viterbi_lds", "k_viterbi"), seed_pdist=("k_seed_pdist",), seed_topk=("k_seed_topk",),
                      estimate=("k_estimate_prod", "k_estimate_blk", "k_estimate"), place=("k_place_blk", "k_place"))
    pmc = {}
    tfile = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            pmc = json.load(open(tfile))["kernels"]
        except Exception:
            pmc = {}

    def pmc_entry(stage):
        for pre in pmc_prefix[stage]:
            for name in sorted(pmc):
                if name.startswith(pre) and not name.startswith(pre + "_"):
                    return name, pmc[name]
        return pmc_prefix[stage][0], {}

    kern = []
    for k in ("viterbi", "seed_pdist", "seed_topk", "estimate", "place"):
        ms = acc[k]
        name, ent = pmc_entry(k)
        tr = ent.get("hbm_bytes_per_launch") if args.batch == 8192 and args.leaves == 99322 else None
        kern.append(dict(kernel=name, ms=round(ms, 3), algorithmic_bytes=alg[k], achieved=alg[k] / (ms * 1e-3) / 1e9, unit="GB/s",
                         frac=alg[k] / (ms * 1e-3) / 1e9 / peak, traffic=tr, traffic_gbps=(tr / (ms * 1e-3) / 1e9 if tr else None)))
    dom = max(kern, key=lambda x: x["ms"])
    roof = dict(bound="hbm", kernel=dom["kernel"], achieved=dom["achieved"], peak=peak, unit="GB/s", frac=dom["frac"], traffic=dom["traffic"])
    # the same kernel with one batch alone on the GPU (in the timed region the kernels of the other batches in flight share it)
    dk = [k for k in ("viterbi", "seed_pdist", "seed_topk", "estimate", "place") if pmc_entry(k)[0] == dom["kernel"]][0]
    roof_iso = dict(bound="hbm", kernel=dom["kernel"], ms=round(iso[dk], 3), achieved=alg[dk] / (iso[dk] * 1e-3) / 1e9, peak=peak, unit="GB/s",
                    frac=alg[dk] / (iso[dk] * 1e-3) / 1e9 / peak, traffic=dom["traffic"])
    bytes_per_read = (D.n_nodes - 1) * R + S * 65 * R + C * 64 * R + args.read_len + 136 * Wp + args.cs_len + 128
    path = dict(bytes_per_read=bytes_per_read, achieved=bytes_per_read * value / world / 1e9, unit="GB/s per GPU",
                frac=bytes_per_read * value / world / 1e9 / peak, mean_R=R, mean_candidates=C)

    out = dict(metric="reads placed/sec (whole node), gg_97_otus GTR+dGamma 250bp; HBM GB/s %peak", value=value, unit="reads/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload="gg_97_otus-scale synthetic DB (%d nodes x %d CS columns, K=%d), GTR+dGamma(%d), SE %d bp amplicon reads, "
                                    "batch %d reads/step/GPU" % (D.n_nodes, args.cs_len, D.K, args.dg_k, args.read_len, args.batch),
                           db_hbm_gb=D.hbm_bytes / 1e9, message_window_cols=db.win[1], parallelism="read-sharded x%d" % world),
               roofline=roof, roofline_one_batch_in_flight=roof_iso, roofline_kernels=kern, roofline_path=path, kernel_ms={k: round(v, 3) for k, v in acc.items()},
               kernel_ms_one_batch_in_flight={k: round(v, 3) for k, v in iso.items()},
               host_wall_ms={k: round(float(v), 2) for k, v in wall.items()}, place_iterations=place_iters)

    # ---- CPU baseline: the oracle (line-faithful port) on this box's host cores, rank 0, N=1 only
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        try:
            from oracle import oracle_py as O
            lo, hi, up_h, down_h = cpu_win
            assert lo <= int(st[ok].min()) and int(en[ok].max()) < hi
            m = O.Model(db.model.type_id, db.model.pi, db.model.par)
            H = O.Hmm(db.hmm.K, db.hmm.L, db.hmm.EM, db.hmm.EI, db.hmm.T, db.hmm.p2cs, 0)
            T = O.Tree(db.parent, db.blen, db.seq, up_h, down_h, db.height, m, db.dg_r if db.dg_k > 0 else None, db.anno_id,
                       win_start=lo, win_len=hi - lo)
            cores = O.max_threads()
            reads = [r.seq for r in all_reads[0]]
            n0 = min(len(reads), max(cores, 16))
            tc = time.perf_counter()
            r0 = O.pipeline_batch(H, T, reads[:n0], all_vps[0][:n0], threads=cores)
            d0 = time.perf_counter() - tc
            ns = args.cpu_sample if args.cpu_sample > 0 else int(min(len(reads), max(n0, 15.0 / max(d0 / n0, 1e-6))))
            tc = time.perf_counter()
            r1 = O.pipeline_batch(H, T, reads[:ns], all_vps[0][:ns], threads=cores)
            d1 = time.perf_counter() - tc
            agree = float((r1["best_nodes"][:ns, 0] == best["c_node"][:ns]).mean())
            out["cpu_baseline"] = dict(value=ns / d1, unit="reads/s", cores=cores, kind="port",
                                       sample="first %d reads of batch 0 (same DB, same reads), OpenMP one read per task" % ns,
                                       stage_cpu_sec=dict(zip(["align", "seed", "estimate", "place"], [round(float(x), 2) for x in r1["stage_sec"]])),
                                       best_branch_agreement_with_gpu=agree)
        except Exception as ex:                             # the baseline must never sink the measurement
            out["cpu_baseline"] = dict(value=None, unit="reads/s", cores=0, kind="port", sample="failed: %r" % (ex,))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
