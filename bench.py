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
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (0 = skip, -1 = about 30 s worth on the CPUs the process may use)")
    ap.add_argument("--seed-order", choices=["stable", "reference"], default="reference",
                    help="hu_opts.seed_order of the timed region: 'reference' [default] = the first max_nseed of libstdc++'s std::sort on dist alone "
                         "(HU_SEED_ORDER_LIBSTDCXX: the reference binary's own tie permutation, src/HmmUFOtu_main.cpp:139 + src/hmmufotu.cpp:646-647, reproduced on the "
                         "device by the pair scan + k_seed_refsort); 'stable' = (dist, node id) (distance-only scan + top-k).  The other order is measured after the "
                         "timed region and reported as a side block (seed_order_stable / seed_order_reference)")
    ap.add_argument("--e2e-reads", type=int, default=int(os.environ.get("HU_BENCH_E2E_READS", 1 << 20)),
                    help="distinct reads of the end-to-end block (host seed lookup + upload + engine + TSV formatting, measured after the timed region; 0 = skip)")
    ap.add_argument("--rehearse", action="store_true",
                    help="control-flow rehearsal WITHOUT the engine (CPU, gloo): launch, rendezvous, barrier, gather, max-over-ranks, one JSON line; value is null")
    return ap.parse_args(argv)


def cpu_quota():
    """CPUs this process may really use: min(os.cpu_count(), affinity, cgroup quota) — the GPU boxes show 256 CPUs under a quota of 16, and threads beyond
    the quota only run the container into its CFS throttle"""
    c = os.cpu_count() or 1
    try:
        c = min(c, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            c = min(c, max(1, -(-int(q) // int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                c = min(c, max(1, -(-q // per)))
        except Exception:
            pass
    w = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    return max(1, c // w) if w > 1 else c


def kernel_source_hash():
    """hash of the sources every kernel is compiled from: ties a PMC summary to the kernels that were running when it was taken"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hmmufotu_amd", "csrc")
    for f in sorted(x for x in os.listdir(d) if x == "hu_engine.hip" or x == "hu_common.h" or (x.startswith("hu_kern_") and x.endswith(".h"))):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start the N ranks as children (nothing in this process has touched the
    GPU or imported torch), forward rank 0's line, exit non-zero when any rank failed."""
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_TOOLS", "ROCP_TOOL_LIB")):
        # the profiler's preloaded library has initialised the GPU in this process: starting the launcher from here is the launcher hop this
        # pool forbids under rocprofv3.  Profile a multi-rank run with the rank program directly after `--`.
        print("[bench] --gpus %d under rocprofv3: refusing to start the launcher from a profiled process" % args.gpus, file=sys.stderr, flush=True)
        sys.exit(2)
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


_REAL_STDOUT = None


def claim_stdout():
    """ONE JSON line on stdout is the contract; libraries this process loads write there too (RCCL prints its version banner to stdout when a process
    group is created).  File descriptor 1 is pointed at stderr for the whole run and the line goes to a duplicate of the original descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _REAL_STDOUT


def emit(line):
    out = claim_stdout()
    out.write(line + "\n"); out.flush()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    claim_stdout()
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
    rccl_error = None
    if world == 1 and not args.rehearse and backend == "nccl" and not os.environ.get("HU_BENCH_NO_RCCL"):
        # one rank is a process group too: RCCL is initialised on this GPU and the final gather of the result records runs through it exactly as at N > 1
        # (an all_gather over one rank), so that the collective path has run on the hardware the line is measured on.  A failure here must not sink the line.
        try:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
            rccl_ranks = dist.get_world_size()
        except Exception as ex:
            rccl_error = repr(ex); dist = None
            log("RCCL at world size 1 failed to initialise: %s" % rccl_error)
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
            emit(json.dumps(dict(metric=metric, value=None, unit="reads/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                                  ms_per_step=dt / max(1, args.steps) * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                                  dtype="f64", data="control-flow rehearsal: no engine, no GPU, nothing measured",
                                  config=dict(workload="rehearsal", parallelism="read-sharded x%d" % world), backend=backend,
                                  rccl_ranks=rccl_ranks, gathered_records=int(len(gathered)), gather_ok=bool(ok),
                                  self_launched=bool(os.environ.get("HU_BENCH_SELF_LAUNCHED")))))
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
    # ---- end-to-end pool: --e2e-reads DISTINCT reads of the headline shape as one byte buffer (drawn from the log-space messages, like the batches)
    e2e_pool = None
    if rank == 0 and world == 1 and args.e2e_reads > 0 and not args.paired:
        tp = time.time()
        e2e_pool = synth_gpu.simulate_pool_gpu(db, up, down, args.e2e_reads, args.read_len, seed=777, amplicon_start=amp_start, amplicon_cols=amp_cols, device=dev)
        log("end-to-end pool: %d distinct reads, %.1f MB (%.0fs)" % (len(e2e_pool[1]) - 1, e2e_pool[0].nbytes / 1e6, time.time() - tp))

    # ---- CPU baseline, phase A (rank 0, N = 1): the oracle's alignment + getSeed on a bounded sample of batch 0, BEFORE the engine adopts the
    # messages (it rewrites them in place into its packed form).  getSeed reads no message; estimateSeq / placeSeq read those of a read's <= 50
    # seed nodes only, so the rows of the sample's seed nodes are gathered from the device here (a few GB) instead of a host copy of the
    # window (40-200 GB) — which is what lets the leg run on the 400 k-node database of config 5.  Phase B (after the timed region) runs the
    # rest of the task on those rows.  The same scan also yields the seed list under the reference's literal std::sort (tie-mode report).
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        try:
            from oracle import oracle_py as O
            cores = max(1, min(O.max_threads(), cpu_quota()))
            m_o = O.Model(db.model.type_id, db.model.pi, db.model.par)
            H_o = O.Hmm(db.hmm.K, db.hmm.L, db.hmm.EM, db.hmm.EI, db.hmm.T, db.hmm.p2cs, 0)
            dummy = np.zeros((1, 1, 4))
            T_a = O.Tree(db.parent, db.blen, db.seq, dummy, dummy, db.height, m_o, db.dg_r if db.dg_k > 0 else None, db.anno_id)
            reads_c = [r.seq for r in all_reads[0]]
            mates_c = [r.seq for r in all_mates[0]] if args.paired else None
            mv_c = all_mvps[0] if args.paired else None

            def phase_a(n, lib):
                # seed_ids: the (dist, node id) list; lib_ids (want_lib): the literal std::sort's, from the same scan
                return O.pipeline_batch(H_o, T_a, reads_c[:n], all_vps[0][:n], mates=mates_c[:n] if mates_c else None, mvpaths=mv_c[:n] if mates_c else None,
                                        opts=O.default_opts(tieMode=0), threads=cores, mode=1, want_lib=lib)
            n0 = min(len(reads_c), max(cores, 16))
            tc = time.perf_counter(); phase_a(n0, False); d0 = time.perf_counter() - tc
            ns = args.cpu_sample if args.cpu_sample > 0 else int(min(len(reads_c), max(n0, 24.0 / max(d0 / n0, 1e-6))))
            ns = min(ns, len(reads_c))
            tc = time.perf_counter(); p1 = phase_a(ns, True); dA = time.perf_counter() - tc
            dA -= p1["extra_thread_sec"] / cores               # the libstdc++ order of the same scan is not part of the task
            okA = p1["aln_ints"][:, 7] == 1
            nodes = np.unique(np.concatenate([p1["seed_ids"].ravel(), p1["lib_ids"].ravel()])); nodes = nodes[nodes >= 0]
            lo = max(db.win[0], int(p1["aln_ints"][okA, 4].min()) - 1); hi = min(db.win[0] + db.win[1], int(p1["aln_ints"][okA, 5].max()))
            rows_up = np.empty((len(nodes), hi - lo, 4)); rows_dn = np.empty_like(rows_up)
            nd_d = torch.tensor(nodes, device=dev)
            for a0 in range(0, len(nodes), 1024):
                ix = nd_d[a0:a0 + 1024]
                rows_up[a0:a0 + 1024] = up[ix, lo - db.win[0]:hi - db.win[0]].cpu().numpy()
                rows_dn[a0:a0 + 1024] = down[ix, lo - db.win[0]:hi - db.win[0]].cpu().numpy()
            row_of = np.full(db.n_nodes, -1, np.int32); row_of[nodes] = np.arange(len(nodes))
            cpu = dict(O=O, H=H_o, m=m_o, T_a=T_a, p1=p1, ns=ns, dA=dA, cores=cores, lo=lo, hi=hi, rows_up=rows_up, rows_dn=rows_dn, row_of=row_of,
                       reads=reads_c, mates=mates_c, mv=mv_c)
            log("CPU leg, phase A: %d reads aligned + seeded on %d threads in %.1fs; message rows of %d seed nodes x %d columns gathered (%.2f GB)"
                % (ns, cores, dA, len(nodes), hi - lo, 2 * rows_up.nbytes / 1e9))
        except Exception as ex:
            import traceback
            traceback.print_exc()
            cpu = dict(failed=repr(ex))
    md = E.model_desc(db.model.type_id, db.model.pi, db.model.par, db.dg_r if db.dg_k > 0 else None)
    D = E.Database.from_arrays(db.hmm, db.parent, db.blen, db.seq, up.data_ptr(), down.data_ptr(), db.height, md, db.anno_id,
                               win_start=db.win[0], win_len=db.win[1] if win else 0, device=local, msgs_on_device=True)
    log("database resident: %.1f GB in HBM, K=%d, nodes=%d, build %.0fs" % (D.hbm_bytes / 1e9, D.K, D.n_nodes, time.time() - t0))
    batches = []
    opts = E.default_opts(seed_order=1 if args.seed_order == "reference" else 0)
    # as many batches in flight as fit beside the database: the first one runs once (its device buffers exist then — in the reference's seed order a
    # (d, N) pair row per read + the sort's scratch, 13 + 5 GB per 8,192 pairs at config 5) and what it took is asked of the others
    per_batch = None
    for i in range(nb):
        free_now, _ = torch.cuda.mem_get_info(local)
        if per_batch is not None and free_now < 1.1 * per_batch + (2 << 30):
            log("%d batches in flight instead of %d: %.1f GB free, %.1f GB per batch" % (i, nb, free_now / 1e9, per_batch / 1e9))
            break
        B = E.Batch(D, args.batch)
        for kv in filter(None, os.environ.get("HU_BENCH_KNOBS", "").split(",")):      # development: engine knobs for an A/B run, "name=value,..." (noted in the line)
            B.set_knob(kv.split("=")[0], int(kv.split("=")[1]))
        if args.paired:
            B.set_reads([r.seq for r in all_reads[i]], all_vps[i], [r.seq for r in all_mates[i]], all_mvps[i])
        else:
            B.set_reads([r.seq for r in all_reads[i]], all_vps[i])     # inputs resident in HBM before the timed region
        B.sync()
        B.assign(opts); B.sync()        # every batch runs once here: its buffers exist, and the free memory the next one is checked against is what is really left
        if i == 0:
            per_batch = free_now - torch.cuda.mem_get_info(local)[0]
        batches.append(B)
    nb = len(batches)
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
    gathered = None
    if world > 1 or rccl_ranks:                             # the one collective: final result gather over RCCL
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
    _, cplaces = B.candidate_places()        # fetched now: the end-to-end block below reuses the batch objects
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
    ref_order = args.seed_order == "reference"
    pair_bytes = 4 if (args.paired or args.read_len > 255) else 2       # the (d, N) pair row of the reference-order mode: 16-bit pairs while no read has more than 255 bases
    alg = dict(viterbi=nseq * (args.read_len + 136.0 * Wp), seed_pdist=(D.n_nodes - 1) * Rsum,
               seed_topk=(float(pair_bytes) * (D.n_nodes - 1) * nread if ref_order else 4.0 * D.n_nodes * nread),      # k_seed_refsort: the pair row once
               estimate=S * 65.0 * Rsum, place=C * 64.0 * Rsum)
    # measured HBM traffic + VALU issue per kernel: from the newest committed PMC summary that was taken on THIS workload
    # (rocprofv3 --pmc passes of this same command, profiles/make_pmc_summary.py); tagged with its source, null otherwise
    stages = ("viterbi", "seed_pdist", "seed_topk", "estimate", "place")
    pmc_prefix = dict(viterbi=("k_viterbi_wave", "k_viterbi_dec2", "k_viterbi_dec", "k_viterbi_lds", "k_viterbi"),
                      seed_pdist=("k_seed_pdist2", "k_seed_pdist") if ref_order else ("k_seed_dscan4", "k_seed_dscan", "k_seed_pdist2", "k_seed_pdist"),
                      seed_topk=("k_seed_refsort",) if ref_order else ("k_seed_topk_straight", "k_seed_topk_d", "k_seed_topk"),
                      estimate=("k_estimate_prod", "k_estimate_blk", "k_estimate"), place=("k_place_blk", "k_place_pair", "k_place"))
    workload_key = dict(leaves=args.leaves, cs_len=args.cs_len, read_len=args.read_len, batch=args.batch, dg_k=args.dg_k, paired=bool(args.paired),
                        uniform_starts=bool(args.uniform_starts), win=args.win)
    if not ref_order:
        workload_key["seed_order"] = "stable"
    if args.partial_frac:
        workload_key["partial_frac"] = args.partial_frac
    pmc, pmc_src, pmc_hash = {}, None, None
    src_hash = kernel_source_hash()
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted((x for x in os.listdir(pdir) if x.endswith("_pmc_summary.json")), reverse=True):
        try:
            j = json.load(open(os.path.join(pdir, f)))
            if j.get("workload") == workload_key:
                pmc, pmc_src, pmc_hash = j["kernels"], "profiles/" + f, j.get("kernel_source_hash")
                break
        except Exception:
            pass

    pmc_stale = bool(pmc) and pmc_hash != src_hash       # counters of other kernel sources than the ones running: quoted, but tagged

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
        # algorithmic_over_hbm_peak: the contract's figure (SURVEY.md section 8d bytes / isolated time / 8 TB/s).  Above 1 it is NOT a bandwidth anybody measured: the kernel does the
        # reference's work on fewer bytes (bit-planes, sixteen reads per node tile, L2 sharing); what the chip really moved and issued are the two fields after it
        e = dict(stage=k, kernel=name, ms_isolated=round(iso[k], 3), ms_in_timed_region=round(acc[k], 3), algorithmic_bytes=alg[k],
                 algorithmic_GBps=alg[k] / (iso[k] * 1e-3) / 1e9, algorithmic_over_hbm_peak=alg[k] / (iso[k] * 1e-3) / 1e9 / peak,
                 traffic=tr, hbm_measured_frac=(tr / (iso[k] * 1e-3) / 1e9 / peak if tr else None))
        if ent.get("valu_issue_cycles_per_launch"):       # typed instruction counts x measured issue cycles (profiles/isa_cost.py), over 1,024 SIMDs
            e["valu_issue_frac"] = ent["valu_issue_cycles_per_launch"] / 1024.0 / (iso[k] * 1e-3 * ent.get("clock_hz", 2.37e9))   # 2.37 GHz: measured, profiles/r02d_clocks.json
        kern.append(e)
    dom = max(kern, key=lambda x: x["ms_isolated"])
    # What binds the dominant kernel is FP64 VALU issue at the occupancy its per-site state allows, not HBM: `bound` says so.
    # `frac` stays the contract's number (algorithmic bytes / isolated time / HBM peak); the two measured roofs sit beside it.
    dfrac = dom["algorithmic_over_hbm_peak"]
    roof = dict(bound="valu_fp64_issue", kernel=dom["kernel"], achieved=dom["algorithmic_GBps"], peak=peak, unit="GB/s", frac=dfrac, traffic=dom["traffic"],
                ms=dom["ms_isolated"], timing="HIP events on the batch's stream, one batch in flight (the kernel's own cost)",
                ms_in_timed_region=dom["ms_in_timed_region"], frac_in_timed_region=dom["algorithmic_bytes"] / (dom["ms_in_timed_region"] * 1e-3) / 1e9 / peak,
                hbm_measured_frac=dom["hbm_measured_frac"], valu_issue_frac=dom.get("valu_issue_frac"), traffic_source=pmc_src,
                pmc_kernel_source_hash=pmc_hash, kernel_source_hash=src_hash, pmc_stale=pmc_stale,
                hbm_roof=dict(achieved=dom["algorithmic_GBps"], peak=peak, unit="GB/s", frac=dfrac, measured_frac=dom["hbm_measured_frac"]),
                valu_roof=dict(frac=dom.get("valu_issue_frac"), unit="issue cycles of the kernel's vector instructions / (1,024 SIMDs x clock x isolated time)"),
                note="frac = ALGORITHMIC bytes of the dominant kernel (SURVEY.md section 8d per-unit figure x the launch's units: FP64 messages once per candidate) / its isolated "
                     "time / 8 TB/s; traffic = PMC bytes of the same launch.  The kernel is bound by FP64 vector issue at the occupancy its per-site state allows (valu_roof), not "
                     "by HBM.  BASELINE.json's '>= 50 % HBM-bandwidth utilisation' is NOT met at configs 2-4 and cannot be by this formulation: after the bit-plane / tiling / "
                     "eigenbasis reformulations every kernel of the path is VALU-issue-bound (roofline_path.path_valu_issue_frac), and the path moves ~a fifth of the HBM peak "
                     "(roofline_path.hbm_measured_frac).  pmc_stale = the PMC summary was taken on other kernel sources than are running now.")
    if dfrac > 1:      # (a seed-scan kernel dominates, e.g. config 5 in the reference's seed order)
        roof["frac_above_1"] = ("the dominant kernel here is the node scan, whose ALGORITHMIC bytes (one int8 site per node, column and read, no reuse credited: SURVEY.md section 8d) are "
                                "not bytes anybody moves: it reads three bit-planes per 32 sites once per tile of sixteen reads.  frac is kept as the contract defines it; "
                                "hbm_measured_frac and valu_issue_frac say how busy the chip is")
    bytes_per_read = (D.n_nodes - 1) * R + S * 65 * R + C * 64 * R + (2 if args.paired else 1) * (args.read_len + 136 * Wp) + args.cs_len + 128
    step_traffic = sum(k["traffic"] for k in kern) if all(k["traffic"] for k in kern) else None
    # every vector instruction of a step, priced at its measured issue cost, over what 1,024 SIMDs could issue in the step's time: how busy the chip's VALUs are, the roof
    # that binds the path.  The per-step kernels of this seed order from the PMC summary (one-time kernels and the other order's seed kernels left out)
    one_time = ("k_alleq_init", "k_node_cover", "k_pack_msgs", "k_col_planes", "k_tree_", "k_pairs_of_read", "k_sort_desc_test")
    other_seed = ("k_seed_dscan", "k_seed_topk_straight", "k_seed_topk_d") if ref_order else ("k_seed_pdist2", "k_seed_refsort", "k_ref_pivots", "k_take_nan_rows")
    issue_cycles = sum(v.get("valu_issue_cycles_per_launch", 0.0) for name_, v in pmc.items() if not name_.startswith(one_time) and not name_.startswith(other_seed))
    clock_hz = next((v.get("clock_hz") for v in pmc.values() if v.get("clock_hz")), 2.37e9)
    path = dict(algorithmic_bytes_per_read=bytes_per_read, algorithmic_GBps_equivalent=bytes_per_read * value / world / 1e9, mean_R=R, mean_candidates=C,
                what="algorithmic bytes (SURVEY.md section 8d: int8 node sites streamed once per read, FP64 messages, no reuse credited) x reads/s.  An EQUIVALENT rate — "
                     "%.0f x the HBM peak — not a bandwidth: the scan works on 3 bit-planes x 16 reads per node tile.  The measured roofs follow" % (bytes_per_read * value / world / 1e9 / peak),
                hbm_measured_bytes_per_step=step_traffic,
                hbm_measured_frac=(step_traffic / (dt / args.steps) / 1e9 / peak if step_traffic else None),
                path_valu_issue_frac=(issue_cycles / 1024.0 / ((dt / args.steps) * clock_hz) if issue_cycles else None),
                traffic_source=pmc_src, pmc_stale=pmc_stale)

    shape = "%s %d bp" % ("PE 2 x" if args.paired else "SE", args.read_len)
    free_b, total_b = torch.cuda.mem_get_info(local)
    out = dict(metric=metric, value=value, unit="reads/s" if not args.paired else "pairs/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload="%s synthetic DB (%d nodes x %d CS columns, K=%d), GTR%s, %s %s reads, "
                                    "batch %d reads/step/GPU, %d batches in flight" % ("SILVA-scale" if args.leaves >= 150000 else "gg_97_otus-scale" if args.leaves >= 90000 else "reduced-scale", D.n_nodes, args.cs_len, D.K, "+dGamma(%d)" % args.dg_k if args.dg_k else "",
                                                                                         shape, "uniform-start" if args.uniform_starts else "amplicon", args.batch, nb) +
                                    (", seeds in the reference's std::sort order (k_seed_refsort)" if args.seed_order == "reference" else ""),
                           seed_order=args.seed_order,
                           db_hbm_gb=D.hbm_bytes / 1e9, message_window_cols=db.win[1], parallelism="read-sharded x%d" % world,
                           hbm_in_use_gb=(total_b - free_b) / 1e9, hbm_total_gb=total_b / 1e9),
               timed_region="the engine's whole per-read task on reads and seed paths already resident (hu_assign_batch + result fetch per step); the host seed "
                            "lookup, FASTA parsing, the read upload (< 1 KB per read) and the TSV are outside it: `end_to_end` below measures them in this same process",
               **({"engine_knobs": os.environ["HU_BENCH_KNOBS"]} if os.environ.get("HU_BENCH_KNOBS") else {}),
               host_cores_busy_per_rank=round(host_cores_busy, 1), host_cpus=os.cpu_count(), host_cpu_quota=cpu_quota(),
               rccl_ranks=rccl_ranks, rccl_error=rccl_error, backend=backend if (world > 1 or rccl_ranks) else None, gathered_records=(int(len(gathered)) if gathered is not None else None),
               roofline=roof, roofline_kernels=kern, roofline_path=path,
               kernel_ms={k: round(v, 3) for k, v in acc.items()}, kernel_ms_one_batch_in_flight={k: round(v, 3) for k, v in iso.items()},
               host_wall_ms={k: round(float(v), 2) for k, v in wall.items()}, place_iterations=place_iters)

    # ---- the same task in the OTHER seed order, after the timed region, on the same resident batches.  The timed region runs the reference's own order
    # (HU_SEED_ORDER_LIBSTDCXX: what a g++-built reference binary outputs wherever nodes tie at the cut-off distance; cpu_baseline.tie_mode counts how often
    # that matters); (dist, node id) — HU_SEED_ORDER_STABLE, the distance-only scan + top-k — is the faster, library-independent mode and is reported beside it
    other = "stable" if ref_order else "reference"
    other_key = "seed_order_" + other
    best_oth = cand_oth = None
    if rank == 0 and world == 1:
        try:
            opts_oth = E.default_opts(seed_order=0 if ref_order else 1)
            nbr = nb
            if not ref_order:     # the reference-order mode asks for more memory per batch (pair rows + sort scratch): as many in flight as fit
                free_ref, _ = torch.cuda.mem_get_info(local)
                need_ref = args.batch * int(D.n_nodes) * 4 + (5 << 30)
                nbr = max(1, min(nb, int(free_ref * 0.9) // need_ref))
            steps_oth = 4 * nbr
            for i in range(nbr):
                batches[i].assign(opts_oth)                     # setup: the mode's buffers exist before the clock starts
            def oth_steps(i):
                for _ in range(steps_oth // nbr):
                    batches[i].assign(opts_oth)
                    batches[i].placements()
            torch.cuda.synchronize()
            tr0 = time.perf_counter()
            list(pool.map(oth_steps, range(nbr)))
            torch.cuda.synchronize()
            dtr = time.perf_counter() - tr0
            best_oth = batches[0].placements().copy(); cand_oth = batches[0].candidates()
            tms = batches[0].timings()
            out[other_key] = dict(value=args.batch * steps_oth / dtr, unit=out["unit"], ms_per_step=dtr / steps_oth * 1e3, steps=steps_oth, batches_in_flight=nbr,
                                  seed_stage_ms=round(tms["seed_pdist"] + tms["seed_topk"], 2),
                                  what=("hu_opts.seed_order = HU_SEED_ORDER_STABLE: ascending (dist, node id) by the distance-only scan + top-k instead of the full (d, N) pair scan + "
                                        "k_seed_refsort; everything else as in the timed region.  NOT the reference binary's tie permutation (cpu_baseline.tie_mode)" if ref_order else
                                        "hu_opts.seed_order = HU_SEED_ORDER_LIBSTDCXX: full (d, N) pair scan + k_seed_refsort (libstdc++'s introsort restricted to the first "
                                        "max_nseed places, as data-parallel Hoare partitions) instead of the distance-only scan + top-k; everything else as in the timed region"))
            log("seed order '%s': %.0f %s (%.1f ms/step, %d batches in flight)" % (other, args.batch * steps_oth / dtr, out["unit"], dtr / steps_oth * 1e3, nbr))
        except Exception as ex:
            import traceback
            traceback.print_exc()
            out[other_key] = dict(value=None, failed=repr(ex))

    # ---- end to end in this process: host seed lookup (hu_seed_index_lookup over the leaf rows' own index) -> read upload -> engine -> TSV lines,
    # over the pool of distinct reads drawn before the database was packed; one worker thread per batch object, chunks dealt in order of completion
    if e2e_pool is not None:
        try:
            import ctypes as Ct
            import queue
            import threading
            cat, offs = e2e_pool
            nr = len(offs) - 1
            tb = time.perf_counter()
            ix = E.SeedIndex(db.parent, db.seq, db.hmm, 20)
            t_index = time.perf_counter() - tb
            chunks = [(a0, min(nr, a0 + args.batch)) for a0 in range(0, nr, args.batch)]
            id_arrays = [(Ct.c_char_p * (b0 - a0))(*[b"r%d" % i for i in range(a0, b0)]) for a0, b0 in chunks]
            anno_strs = [b"taxon%d" % int(x) for x in db.anno_id]
            anno_arr = (Ct.c_char_p * len(anno_strs))(*anno_strs)
            e2e_diag = bool(os.environ.get("HU_BENCH_E2E_PROFILE"))      # diagnostic: per-kernel HIP events stay on and the last batch's stage times are reported
            for B_ in batches:
                B_.profile(e2e_diag)
            seeded = [[0, 0] for _ in range(nb)]
            stage = [dict(lookup=0.0, upload=0.0, engine=0.0, tsv=0.0) for _ in range(nb)]
            done = [0] * nb; tsv_bytes = [0] * nb; placed = [0] * nb
            q = queue.Queue()
            for c_ in range(len(chunks)):
                q.put(c_)

            def worker(w):
                B_ = batches[w]
                while True:
                    try:
                        c_ = q.get_nowait()
                    except queue.Empty:
                        return
                    a0, b0 = chunks[c_]
                    sub = cat[offs[a0]:offs[b0]]; so = offs[a0:b0 + 1] - offs[a0]
                    x0 = time.perf_counter(); vp = pre_vp[c_] if pre_vp is not None else ix.lookup_packed(sub, so, 50, 0)
                    seeded[w][0] += int(vp[:, 0].any(axis=1).sum()); seeded[w][1] += int(vp[:, 1].any(axis=1).sum())
                    x1 = time.perf_counter(); B_.set_reads_packed(sub, so, vp)
                    x2 = time.perf_counter(); B_.assign(opts)
                    if e2e_diag and os.environ.get("HU_BENCH_E2E_REPEAT") and len(rep_log) < 12:      # diagnostic: the same reads assigned again at once
                        t1_ = B_.timings(); w1_ = time.perf_counter() - x2
                        y0 = time.perf_counter(); B_.assign(opts); w2_ = time.perf_counter() - y0
                        rc_ = B_.alignments(want_align=False)["recs"]; R_ = np.sort(rc_["cs_end"] - rc_["cs_start"] + 1)
                        rep_log.append(dict(region_cols=dict(p50=int(R_[len(R_) // 2]), p99=int(R_[int(len(R_) * 0.99)]), p999=int(R_[int(len(R_) * 0.999)]), max=int(R_[-1]), over_1024=int((R_ > 1024).sum())), first={k: round(v, 2) for k, v in t1_.items()}, first_wall_ms=round(w1_ * 1e3, 1),
                                            again={k: round(v, 2) for k, v in B_.timings().items()}, again_wall_ms=round(w2_ * 1e3, 1)))
                    x3 = time.perf_counter(); tsv_bytes[w] += 0 if pre_vp is not None else B_.format_tsv_bytes(id_arrays[c_], None, anno_arr)
                    x4 = time.perf_counter()
                    st_ = stage[w]; st_["lookup"] += x1 - x0; st_["upload"] += x2 - x1; st_["engine"] += x3 - x2; st_["tsv"] += x4 - x3
                    done[w] += b0 - a0
                    placed[w] += int((B_.placements()["c_node"] >= 0).sum())
            pre_vp = None; rep_log = []
            if os.environ.get("HU_BENCH_E2E_NOHOST"):      # diagnostic: lookups done before the clock, no TSV — what is left is upload + engine on fresh reads
                pre_vp = [ix.lookup_packed(cat[offs[a0]:offs[b0]], offs[a0:b0 + 1] - offs[a0], 50, 0) for a0, b0 in chunks]
            te = time.perf_counter()
            th = [threading.Thread(target=worker, args=(w,)) for w in range(nb)]
            [t.start() for t in th]; [t.join() for t in th]
            torch.cuda.synchronize()
            de = time.perf_counter() - te
            out["end_to_end"] = dict(value=sum(done) / de, unit="reads/s", reads=int(sum(done)), distinct_reads=int(nr), seconds=de, placed=int(sum(placed)),
                                     tsv_mb=sum(tsv_bytes) / 1e6, workers=nb,
                                     stages="hu_seed_index_lookup (5' + 3' seeds, depth-32 suffix index over the leaf rows) -> hu_batch_set_reads (upload) -> hu_assign_batch -> "
                                            "hu_batch_format_tsv_ptr (one line per read incl. the csLen-character alignment)",
                                     stage_busy_sec={k: round(sum(s_[k] for s_ in stage), 2) for k in stage[0]},
                                     seed_index_build_sec=round(t_index, 1), seed_index_gb=ix.bytes / 1e9,
                                     reads_with_5p_seed=sum(x[0] for x in seeded), reads_with_3p_seed=sum(x[1] for x in seeded),
                                     excluded="FASTA parsing and the file write (profiles/measure_cli.py times the product CLI with both)")
            if e2e_diag and rep_log:
                out["end_to_end"]["fresh_then_again"] = rep_log
            if e2e_diag:
                out["end_to_end"]["last_batch_kernel_ms"] = [{k: round(v, 2) for k, v in B_.timings().items()} for B_ in batches]
                out["end_to_end"]["last_batch_host_wall_ms"] = [{k: round(float(v), 2) for k, v in B_.wall().items()} for B_ in batches]
                out["end_to_end"]["last_batch_full_dp_reads"] = [int(B_.alignments(want_align=False)["recs"]["used_full"].sum()) for B_ in batches]
                out["end_to_end"]["last_batch_mean_candidates"] = [round(float(B_.placements()["n_cand"].mean()), 2) for B_ in batches]
                out["end_to_end"]["last_batch_mean_region_cols"] = [round(float((B_.alignments(want_align=False)["recs"]["cs_end"] - B_.alignments(want_align=False)["recs"]["cs_start"] + 1).mean()), 1) for B_ in batches]
            log("end to end: %.0f reads/s over %d distinct reads (%.1fs; index build %.1fs)" % (sum(done) / de, nr, de, t_index))
            del ix
        except Exception as ex:
            import traceback
            traceback.print_exc()
            out["end_to_end"] = dict(value=None, failed=repr(ex))

    # ---- CPU baseline, phase B: estimateSeq / filterPlacements / placeSeq / calcQValues of the oracle on the seeds of phase A, on this box's host
    # cores; doubles as the full-scale parity check — ids: every difference of the final pick must be a documented near-tie (oracle/parity.py);
    # numbers: relative error of every candidate's estimated loglik and placed ratio / wnr / height, and iteration counts — and carries the
    # tie-mode report (SURVEY.md H1 ii): the same reads under the reference's literal std::sort seed order.
    if cpu is not None and "failed" in cpu:
        out["cpu_baseline"] = dict(value=None, unit=out["unit"], cores=0, kind="port", sample="failed: " + cpu["failed"])
    elif cpu is not None:
        try:
            from oracle import parity
            O = cpu["O"]; ns = cpu["ns"]; cores = cpu["cores"]; p1 = cpu["p1"]
            T_b = O.Tree(db.parent, db.blen, db.seq, cpu["rows_up"], cpu["rows_dn"], db.height, cpu["m"], db.dg_r if db.dg_k > 0 else None, db.anno_id,
                         win_start=cpu["lo"], win_len=cpu["hi"] - cpu["lo"])
            T_b.set_rows(cpu["row_of"])
            reads, mates, mv = cpu["reads"], cpu["mates"], cpu["mv"]
            tc = time.perf_counter()
            r1 = O.pipeline_batch(cpu["H"], T_b, reads[:ns], all_vps[0][:ns], mates=mates[:ns] if mates else None, mvpaths=mv[:ns] if mates else None,
                                  threads=cores, want_cands=True, mode=2, seeds=(p1["seed_cnt"], p1["lib_ids"] if args.seed_order == "reference" else p1["seed_ids"]))
            dB = time.perf_counter() - tc
            per = []
            for i in range(ns):
                k = int(r1["n_cand"][i]); a, b = int(cand["offs"][i]), int(cand["offs"][i + 1])
                per.append(parity.classify_read(r1["cand_node"][i, :k], r1["cand_est"][i, :k], r1["cand_ratio0"][i, :k],
                                                cand["c_node"][a:b], db.parent, pos=int(r1["best_pos"][i]) if k else None))
            tot = parity.summarize(per)
            agree = float((r1["best_nodes"][:ns, 0] == best["c_node"][:ns]).mean())
            # numbers: candidate by candidate (matched by branch inside a read)
            rel = dict(est_loglik=[], ratio=[], wnr=[], height=[]); it_out = it_em = ncmp = nan_mismatch = 0
            for i in range(ns):
                k = int(r1["n_cand"][i]); a = int(cand["offs"][i]); b = int(cand["offs"][i + 1])
                pos = {int(nn): j for j, nn in enumerate(r1["cand_node"][i, :k])}
                for c in range(a, b):
                    j = pos.get(int(cand["c_node"][c]))
                    if j is None:
                        continue
                    ncmp += 1
                    pairs = (("est_loglik", cand["est_loglik"][c], r1["cand_est"][i, j]), ("ratio", cand["ratio"][c], r1["cand_placed"][i, j, 0]),
                             ("wnr", cand["wnr"][c], r1["cand_placed"][i, j, 1]), ("height", cplaces["height"][c], r1["cand_placed"][i, j, 2]))
                    for nm, g, o_ in pairs:
                        if np.isnan(g) or np.isnan(o_):
                            nan_mismatch += int(np.isnan(g) != np.isnan(o_))
                        else:
                            rel[nm].append(abs(g - o_) / max(abs(o_), 1e-3))
                    it_out += int((int(cand["iters"][c]) & 255) != int(r1["cand_iters"][i, j, 0]))
                    it_em += int((int(cand["iters"][c]) >> 8) != int(r1["cand_iters"][i, j, 1]))
            num = {nm: dict(max_rel=float(np.max(v)) if v else None, p99_rel=float(np.percentile(v, 99)) if v else None) for nm, v in rel.items()}
            # tie-mode report on the same sample
            tt = time.perf_counter()
            tper, tsum = O.tie_report(cpu["H"], cpu["T_a"], reads[:ns], all_vps[0][:ns], mates[:ns] if mates else None, mv[:ns] if mates else None,
                                      threads=cores, phase1=p1, tree2=T_b)
            tsum["seconds"] = round(time.perf_counter() - tt, 1)
            tsum["what"] = ("the same reads with the seeds kept under the reference's literal std::sort on dist alone (libstdc++ introsort's tie permutation, "
                            "src/HmmUFOtu_main.cpp:139 + src/hmmufotu.cpp:646-647) against the product's (dist, node id) order; both from one scan per read")
            out["cpu_baseline"] = dict(value=ns / (cpu["dA"] + dB), unit=out["unit"], cores=cores, kind="port",
                                       sample="first %d reads of batch 0 (same DB, same reads), OpenMP one read per task; alignment + getSeed timed before the engine packs "
                                              "the messages (%.1f s), estimate / filter / place / q-values after the GPU run on the gathered rows of the seed nodes (%.1f s)" % (ns, cpu["dA"], dB),
                                       stage_cpu_sec=dict(zip(["align", "seed", "estimate", "place"],
                                                              [round(float(x), 2) for x in (p1["stage_sec"][0], p1["stage_sec"][1], r1["stage_sec"][2], r1["stage_sec"][3])])),
                                       oracle_seed_order="TIE_LIBSTDCXX (literal std::sort)" if args.seed_order == "reference" else "TIE_STABLE (dist, node id)",
                                       best_branch_agreement_with_gpu=agree,
                                       best_branch_diffs=tot["best_differs"], unexplained_best_branch_diffs=tot["best_unexplained"],
                                       candidate_order=dict(swaps_explained_near_tie=tot["swaps_explained"], swaps_unexplained=tot["swaps_unexplained"],
                                                            candidate_set_differs=tot["set_differs"]),
                                       unexplained_detail=[repr(d) for r_ in per for d in r_["detail"]][:20],
                                       max_rel=dict(candidates_compared=ncmp, tolerance=1e-6, **num, nan_placement_mismatches=nan_mismatch,
                                                    outer_iteration_mismatches=it_out, em_iteration_mismatches=it_em,
                                                    note="relative to max(|oracle|, 1e-3), every candidate of every sampled read matched by branch; est_loglik = estimateSeq's, "
                                                         "ratio / wnr / height = placeSeq's; iterations = outer loops of the joint optimisation and passes of the 2-node EM"),
                                       tie_mode=tsum)
            if best_oth is not None:    # the engine in the other seed order against the oracle in that order (tie_report ran both lists wherever they differ), read by read
                col = 0 if ref_order else 1                 # tie_report's picks[:, 0] = under (dist, node id), [:, 1] = under the literal std::sort
                want_c = np.where(tper["order_differs"][:ns], tper["picks"][:ns, col, 0], r1["best_nodes"][:ns, 0])
                row_of = {int(i): k for k, i in enumerate(tper["lib_idx"])}; ro = tper["stable_run"] if ref_order else tper["lib_run"]
                perr = []
                for i in range(ns):                           # the same classification as for the timed region's order
                    src_, k_ = (ro, row_of[i]) if i in row_of else (r1, i)
                    kc = int(src_["n_cand"][k_]); a, b = int(cand_oth["offs"][i]), int(cand_oth["offs"][i + 1])
                    perr.append(parity.classify_read(src_["cand_node"][k_, :kc], src_["cand_est"][k_, :kc], src_["cand_ratio0"][k_, :kc],
                                                     cand_oth["c_node"][a:b], db.parent, pos=int(src_["best_pos"][k_]) if kc else None))
                totr = parity.summarize(perr)
                out[other_key].update(reads_compared=int(ns), oracle_seed_order="TIE_STABLE (dist, node id)" if ref_order else "TIE_LIBSTDCXX (literal std::sort)",
                                      final_branch_differs_from_oracle=int((best_oth["c_node"][:ns] != want_c).sum()),
                                      best_branch_diffs_explained_near_tie=totr["best_differs"] - totr["best_unexplained"], unexplained_best_branch_diffs=totr["best_unexplained"],
                                      candidate_set_differs=totr["set_differs"], swaps_unexplained=totr["swaps_unexplained"])
        except Exception as ex:                             # the baseline must never sink the measurement
            import traceback
            traceback.print_exc()
            out["cpu_baseline"] = dict(value=None, unit="reads/s", cores=0, kind="port", sample="failed: %r" % (ex,))
    if rank == 0:
        emit(json.dumps(out))
    if world > 1 or rccl_ranks:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
