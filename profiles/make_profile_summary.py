#!/usr/bin/env python3
"""Turns a gpurun_out/profN directory (rocprofv3 --kernel-trace --stats of bench.py + two --pmc passes) into the
tracked summaries under profiles/:  python3 profiles/make_profile_summary.py gpurun_out/prof4 r01_final"""
import collections, csv, glob, json, shutil, sys

d, tag = sys.argv[1], sys.argv[2]
st = glob.glob(d + '/stats/runc/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(st)))
mine = [r for r in rows if r['Name'].startswith(('k_', 'void k_'))]
clean = lambda n: n.split('(')[0].replace('void ', '')
onetime = lambda nm: nm.startswith(('k_pack', 'k_tree', 'k_model', 'k_col'))
stage = [('k_viterbi', 'viterbi'), ('k_align', 'align_build'), ('k_encode', 'align_build'), ('k_tile', 'align_build'), ('k_merge', 'align_build'),
         ('k_seed_pdist', 'seed_pdist'), ('k_seed_topk', 'seed_topk'), ('k_estimate', 'estimate'), ('k_place', 'place')]
def stage_of(nm):
    for pre, s in stage:
        if nm.startswith(pre):
            return s
    return ''
tot = sum(int(r['TotalDurationNs']) for r in mine if not onetime(clean(r['Name'])))
b = json.load(open(d + '/bench.json'))
with open('profiles/%s_kernel_stats_engine.csv' % tag, 'w') as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); [w.writerow(r) for r in mine]
km = b['kernel_ms']; iso = b.get('kernel_ms_one_batch_in_flight', {})
with open('profiles/%s_summary.md' % tag, 'w') as f:
    f.write("# rocprofv3 --kernel-trace --stats, round 1 final kernels\n\n")
    f.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --cpu-sample 0` (defaults: %d warm-up + %d timed steps, "
            "batches in flight as in the bench line, then 3 steps of one batch alone)\n\n" % (b['warmup'], b['steps']))
    f.write("1 x MI355X, %s.  bench line of the same (profiled) run: %.0f reads/s, %.1f ms/step.\n\n" % (b['config']['workload'], b['value'], b['ms_per_step']))
    f.write("| kernel | calls | rocprof avg ms (all calls) | share of per-read engine GPU time | bench.py HIP events of its stage, timed region (batches overlapping) | bench.py HIP events, one batch alone |\n|---|---|---|---|---|---|\n")
    for r in sorted(mine, key=lambda r: -int(r['TotalDurationNs'])):
        nm = clean(r['Name']); sg = stage_of(nm)
        f.write("| %s | %s | %.3f | %s | %s | %s |\n" % (nm, r['Calls'], float(r['AverageNs']) / 1e6,
                "one-time (DB build/load)" if onetime(nm) else "%.1f%%" % (100 * int(r['TotalDurationNs']) / tot), km.get(sg, ''), iso.get(sg, '')))
    f.write("\nThe rocprof average mixes calls that overlap with the other batches' kernels (one stream per batch in flight) and calls of a batch alone; "
            "bench.py reports both regimes separately.\n`viterbi` in bench.py covers the fill and the traceback kernel; `align_build` covers k_align_rows + k_encode_rows + k_tile_lists.\n")
print(open('profiles/%s_summary.md' % tag).read())

def agg(path):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        a[clean(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in a.items()}
fp = glob.glob(d + '/fetch/runc/*counter_collection.csv')[0]; wp = glob.glob(d + '/write/runc/*counter_collection.csv')[0]
f_, w_ = agg(fp), agg(wp)
out = {}
for k in f_:
    fb = f_[k][0] * 1024 * 2; wb = w_.get(k, (0, 0))[0] * 1024
    out[k] = dict(fetch_size_kb=f_[k][0], write_size_kb=w_.get(k, (0, 0))[0], hbm_read_bytes=fb, hbm_write_bytes=wb, hbm_bytes_per_launch=fb + wb, launches=f_[k][1])
    print("%-44s read %8.2f GB  write %8.2f GB" % (k, fb / 1e9, wb / 1e9))
fb_ = json.load(open(d + '/fetch.json'))
json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (bench.py --steps 4 --warmup 0 --cpu-sample 0), full gg_97-scale DB, "
               "%s. FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; calibrated on k_pack_msgs: "
               "48.83 GB read, FETCH_SIZE reports 24.42 GB)." % fb_['config']['workload'], batch=8192, kernels=out),
          open('profiles/r01_pmc_traffic.json', 'w'), indent=1)
shutil.copy(fp, 'profiles/r01_pmc_fetch_size.csv'); shutil.copy(wp, 'profiles/r01_pmc_write_size.csv')
shutil.copy(st, 'profiles/%s_kernel_stats_full.csv' % tag); shutil.copy(d + '/bench.json', 'profiles/bench_%s_profiled.json' % tag)
