#!/usr/bin/env python3
"""gpurun_out/<tag>/ (profiles/collect_pmc.sh) -> the tracked summaries:
     profiles/<tag>_pmc_summary.json    per kernel: HBM bytes per launch, SQ / L2 counters, VALU instructions by type and the
                                        issue cycles they cost; `workload` = the key bench.py matches before quoting it;
                                        `kernel_source_hash` = bench.py's hash of the kernel sources of the profiled run (a bench run
                                        on other sources tags the quoted counters pmc_stale)
     profiles/<tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats rows of the engine's kernels
     profiles/<tag>_summary.md          one table
   HBM bytes: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: gfx950 tallies the 128-B requests
   of wide coalesced reads at 64 B), WRITE_SIZE is taken as is.
   VALU issue cycles per launch = 4 x (ADD + MUL + FMA)_F64 + 16 x TRANS_F64 + 8 x TRANS_F32 + c x every other VALU instruction,
   c = the mean issue cost of those instructions in the kernel's inner loops (profiles/isa_cost.py over the ISA listing with the rates
   measured by profiles/ubench/valu_rate.hip: 2 cycles only for a handful of operations on vector registers alone, 4 for the rest and
   for anything with a scalar operand; profiles/<tag>_isa_costs.json), 4 when the kernel is not in that file.
   Usage: python3 profiles/make_pmc_summary.py gpurun_out/<tag> <tag> '<workload json>'"""
import collections, csv, json, os, shutil, sys

d, tag = sys.argv[1], sys.argv[2]
workload = json.loads(sys.argv[3]) if len(sys.argv) > 3 else None
clean = lambda n: n.split('(')[0].replace('void ', '')


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        a[clean(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())} for k, cs in a.items()}


fe, wr, sq, va = (agg(os.path.join(d, x + "_engine.csv")) for x in ("fetch", "write", "sq", "valu"))
isa = {}
if os.path.exists("profiles/%s_isa_costs.json" % tag):
    isa = json.load(open("profiles/%s_isa_costs.json" % tag))["kernels"]
bench = json.load(open(os.path.join(d, "bench_profiled.json")))
iso = {}
for k in bench.get("roofline_kernels", []):
    iso[k["kernel"]] = k
out = {}
for k in sorted(set(fe) | set(sq)):
    e = dict(launches=int(sq.get(k, fe.get(k, {})).get("launches", 0)))
    if k in fe:
        e["fetch_size_kb"] = fe[k]["FETCH_SIZE"]; e["hbm_read_bytes"] = fe[k]["FETCH_SIZE"] * 1024 * 2
    if k in wr:
        e["write_size_kb"] = wr[k]["WRITE_SIZE"]; e["hbm_write_bytes"] = wr[k]["WRITE_SIZE"] * 1024
    if "hbm_read_bytes" in e:
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes"] + e.get("hbm_write_bytes", 0.0)
    for c, v in sq.get(k, {}).items():
        if c != "launches":
            e[c] = v
    if k in va and k in sq:
        v = va[k]
        f64 = v.get("SQ_INSTS_VALU_ADD_F64", 0) + v.get("SQ_INSTS_VALU_MUL_F64", 0) + v.get("SQ_INSTS_VALU_FMA_F64", 0)
        t64, t32 = v.get("SQ_INSTS_VALU_TRANS_F64", 0), v.get("SQ_INSTS_VALU_TRANS_F32", 0)
        tot = sq[k].get("SQ_INSTS_VALU", 0)
        e.update({c: x for c, x in v.items() if c != "launches"})
        e["valu_fp64_frac"] = (f64 + t64) / tot if tot else None
        rest_cost = (isa.get(k) or {}).get("rest_mean_cycles") or 4.0
        e["valu_rest_mean_cycles"] = rest_cost
        e["valu_issue_cycles_per_launch"] = 4 * f64 + 16 * t64 + 8 * t32 + rest_cost * max(0.0, tot - f64 - t64 - t32)
    if sq.get(k, {}).get("SQ_WAVE_CYCLES"):
        e["valu_active_per_wave_cycle"] = sq[k].get("SQ_ACTIVE_INST_VALU", 0) / sq[k]["SQ_WAVE_CYCLES"]
        e["waiting_frac"] = sq[k].get("SQ_WAIT_ANY", 0) / sq[k]["SQ_WAVE_CYCLES"]
        e["issue_stall_frac"] = sq[k].get("SQ_WAIT_INST_ANY", 0) / sq[k]["SQ_WAVE_CYCLES"]
    if sq.get(k, {}).get("TCC_HIT_sum") is not None:
        h, m = sq[k].get("TCC_HIT_sum", 0), sq[k].get("TCC_MISS_sum", 0)
        e["l2_hit_rate"] = h / (h + m) if h + m else None
    out[k] = e
note = ("rocprofv3 --pmc passes of `python3 bench.py --steps 2 --warmup 0 --cpu-sample 0 --inflight 1 <workload flags>` (one batch in flight: every dispatch "
        "alone on the GPU), counters averaged over a kernel's launches; see the header of profiles/make_pmc_summary.py for the corrections")
json.dump(dict(note=note, workload=workload, bench_config=bench.get("config"), kernel_source_hash=(bench.get("roofline") or {}).get("kernel_source_hash"), kernels=out), open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)
rows = list(csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))))
mine = [r for r in rows if clean(r['Name']).startswith('k_')]
with open("profiles/%s_kernel_stats.csv" % tag, "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); [w.writerow(r) for r in mine]
shutil.copy(os.path.join(d, "bench_profiled.json"), "profiles/bench_%s_profiled.json" % tag)
one = {}
p1 = os.path.join(d, "kernel_stats_inflight1.csv")
if os.path.exists(p1):
    r1 = [r for r in csv.DictReader(open(p1)) if clean(r['Name']).startswith('k_')]
    one = {clean(r['Name']): r for r in r1}
    with open("profiles/%s_kernel_stats_inflight1.csv" % tag, "w") as f:
        w = csv.DictWriter(f, fieldnames=list(r1[0].keys())); w.writeheader(); [w.writerow(r) for r in r1]
    shutil.copy(os.path.join(d, "bench_profiled_inflight1.json"), "profiles/bench_%s_profiled_inflight1.json" % tag)
for x in ("fetch", "write", "sq", "valu"):
    shutil.copy(os.path.join(d, x + "_engine.csv"), "profiles/%s_pmc_%s.csv" % (tag, x))
onetime = lambda nm: nm.startswith(('k_pack', 'k_tree', 'k_model', 'k_col'))
stage_prefix = [("k_viterbi", "viterbi"), ("k_seed_dscan", "seed_pdist"), ("k_seed_pdist", "seed_pdist"), ("k_seed_topk", "seed_topk"), ("k_seed_refsort", "seed_topk"), ("k_estimate", "estimate"), ("k_place", "place")]
by_stage = {k["stage"]: k for k in bench.get("roofline_kernels", [])}
def stage_entry(nm):
    for pre, st in stage_prefix:
        if nm.startswith(pre):
            return by_stage.get(st)
    return None
class _KM(dict):
    def get(self, nm, default=None):
        return stage_entry(nm) or default
km = _KM()
with open("profiles/%s_summary.md" % tag, "w") as f:
    f.write("# %s: rocprofv3 --kernel-trace --stats + PMC passes\n\n" % tag)
    f.write("Stats command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --cpu-sample 0 <workload flags>`; bench line of that (profiled) run: "
            "%.0f %s, %.2f ms/step.  Workload: %s\n\n" % (bench["value"], bench["unit"], bench["ms_per_step"], bench["config"]["workload"]))
    f.write("| kernel | calls | rocprof avg ms, six batches in flight (+ the 3 isolated steps at the end) | rocprof avg ms, one batch in flight (`--inflight 1`) | bench.py HIP events: in the timed region / one batch alone | HBM GB per launch (PMC) | "
            "HBM GB/s alone (% of 8 TB/s) | VALU wave-instr per launch (FP64 share) | VALU issue: % of the SIMDs' cycles alone | VALU active / wave cycle | L2 hit |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in sorted(mine, key=lambda r: -int(r['TotalDurationNs'])):
        nm = clean(r['Name'])
        if onetime(nm):
            continue
        e = out.get(nm, {}); b = km.get(nm)
        if b and clean(b["kernel"]) != nm:      # the stage's HIP-event time belongs to the kernel bench.py names for it; the others are rated on their own rocprof time alone
            b = None
        ms = b["ms_isolated"] if b else (float(one[nm]['AverageNs']) / 1e6 if nm in one else None)
        hb = e.get("hbm_bytes_per_launch")
        f.write("| %s | %s | %.3f | %s | %s | %s | %s | %s | %s | %s | %s |\n" % (
            nm, r['Calls'], float(r['AverageNs']) / 1e6, "%.3f" % (float(one[nm]['AverageNs']) / 1e6) if nm in one else "",
            "%.2f / %.2f" % (b["ms_in_timed_region"], b["ms_isolated"]) if b else "",
            "%.2f" % (hb / 1e9) if hb else "",
            "%.0f (%.0f %%)" % (hb / ms / 1e6, hb / ms / 1e6 / 80) if hb and ms else "",
            "%.3g (%.0f %%)" % (e["SQ_INSTS_VALU"], 100 * e["valu_fp64_frac"]) if e.get("valu_fp64_frac") is not None else "",
            "%.0f %%" % (100 * e["valu_issue_cycles_per_launch"] / 1024 / (ms * 1e-3 * 2.4e9)) if e.get("valu_issue_cycles_per_launch") and ms else "",
            "%.2f" % e["valu_active_per_wave_cycle"] if "valu_active_per_wave_cycle" in e else "",
            "%.2f" % e["l2_hit_rate"] if e.get("l2_hit_rate") is not None else ""))
    f.write("\nVALU issue %: 4 cycles per FP64 add/mul/fma, 16 per FP64 transcendental (v_rcp_f64), 8 per FP32 transcendental, and for every other VALU wave-instruction the mean issue cost "
            "of such instructions in the kernel's inner loops (profiles/isa_cost.py, rates measured by profiles/ubench/valu_rate.hip: 2 cycles for a few operations on vector registers alone, else 4), over 1,024 SIMDs at 2.4 GHz "
            "and the kernel's isolated time (bench.py HIP events with one batch in flight where the column is filled, else the rocprof average of the one-batch-in-flight run); the shader clock measured during the timed region is 2.37 GHz on average at 1.07 kW (profiles/r02d_clocks.json), 1.3 % under the 2.4 GHz used here.\n")
print(open("profiles/%s_summary.md" % tag).read())
