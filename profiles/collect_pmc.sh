#!/bin/bash
# Round-2 profile collection on the GPU box (gpurun):  bash profiles/collect_pmc.sh <tag> [bench.py workload flags]
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (+ the given workload flags): per-kernel durations
#   2. four --pmc passes (counters only, no trace domain) of the same workload with ONE batch in flight, so that every
#      dispatch runs alone and its counters are its own: FETCH_SIZE | WRITE_SIZE | SQ issue + L2 hit counters | VALU by type
# Outputs under gpurun_out/<tag>/; profiles/make_pmc_summary.py turns them into the tracked summaries.
set -o pipefail
tag=$1; shift
# one rank only: with --gpus N > 1 bench.py would start its launcher from a process the profiler's preloaded library has already
# initialised on the GPU — a launcher hop under rocprofv3, which this pool forbids (bench.py refuses it too)
prev=""
for a in "$@"; do
	if [ "$prev" = "--gpus" ] && [ "$a" != "1" ]; then echo "collect_pmc.sh profiles ONE rank: --gpus $a refused" >&2; exit 2; fi
	case "$a" in --gpus=1) ;; --gpus=*) echo "collect_pmc.sh profiles ONE rank: $a refused" >&2; exit 2;; esac
	prev="$a"
done
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
short="--steps 2 --warmup 0 --cpu-sample 0 --inflight 1 --e2e-reads 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/bench.py --cpu-sample 0 --e2e-reads 0 "$@" > $out/bench_profiled.json 2> $out/stats.err || exit 1
echo "stats pass done"
# the same with ONE batch in flight: every launch alone on the GPU -> the average duration bench.py's `roofline.ms` (isolated) must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o run -- python3 $root/bench.py --cpu-sample 0 --e2e-reads 0 --inflight 1 --steps 8 --warmup 2 "$@" > $out/bench_profiled_inflight1.json 2> $out/stats1.err || exit 1
echo "stats (one batch in flight) pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o run -- python3 $root/bench.py $short "$@" > $out/fetch.json 2> $out/fetch.err || exit 1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o run -- python3 $root/bench.py $short "$@" > $out/write.json 2> $out/write.err || exit 1
echo "write pass done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU TCC_HIT_sum TCC_MISS_sum \
	--output-format csv -d $out/sq -o run -- python3 $root/bench.py $short "$@" > $out/sq.json 2> $out/sq.err || exit 1
echo "sq pass done"
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 \
	--output-format csv -d $out/valu -o run -- python3 $root/bench.py $short "$@" > $out/valu.json 2> $out/valu.err || exit 1
echo "valu pass done"
# keep the merge small: the per-dispatch CSVs of the engine's kernels only
for d in fetch write sq valu; do
	f=$(ls $out/$d/*/*counter_collection.csv $out/$d/*counter_collection.csv 2>/dev/null | head -1)
	[ -n "$f" ] && (head -1 $f; grep -E '"(void )?k_' $f) > $out/${d}_engine.csv
done
f=$(ls $out/stats/*/*kernel_stats.csv $out/stats/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
f=$(ls $out/stats1/*/*kernel_stats.csv $out/stats1/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_inflight1.csv
rm -rf $out/stats $out/stats1 $out/fetch $out/write $out/sq $out/valu
ls -la $out
