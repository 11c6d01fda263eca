// HU_SEED_ORDER_LIBSTDCXX on the device: the first max_nseed places of std::sort(locs) — libstdc++'s introsort on dist ALONE over all
// ~n_nodes PTLocs in node order (src/HmmUFOtu_main.cpp:139, src/hmmufotu.cpp:646-647) — from the (d, N) pair row the scan left.
// The algorithm restated (hu_host.cpp has the sequential restatement and the note on its source):
//   introsort loop on [first, last) while it holds more than 16 elements: pivot = median of first + 1, mid, last - 1 swapped into *first;
//   unguarded Hoare partition of [first + 1, last) against it; recursion into [cut, last), loop on [first, cut); depth limit 2 lg n;
//   one final insertion sort.  Only ranges that reach into the first K places matter.
//
// One workgroup per read.  A Hoare partition is a deterministic pairing: with the LEFT STOPPERS (elements >= pivot) numbered from the
// left, i_1 < i_2 < ..., and the RIGHT STOPPERS (elements <= pivot) numbered from the right, j_1 > j_2 > ..., the sequential scans swap
// exactly the pairs (i_k, j_k) with i_k < j_k, k = 1 .. m, and return cut = min(i_(m+1), j_m) (the left scan of the last round stops at
// the next original left stopper or at the swapped-in element at j_m, whichever comes first; cut = i_1 when nothing is swapped).  Ranks
// are prefix counts, so a partition is two data-parallel passes: count the stoppers per 64-element subtile (ballots), scan the counts,
// locate cut and m, copy the right stoppers j_1 .. j_m out by rank, and write the left part [first, cut) of the next level with the left
// stoppers replaced by them.  Only the left part is written: the right part matters only when cut falls inside the first K places —
// then the (tiny) left part is set aside and the right part is written instead.  Ranges of at most HU_RS_SMALL elements are finished
// by one thread in LDS with the literal sequential algorithm.  Reads the device does not finish (a NaN distance, the heap-sort branch of
// introsort, keys that do not fit) are listed for the host path.
//
// No floating point anywhere: dist = d / N is compared through d_a * N_b < d_b * N_a in the streaming passes, and the sequential finisher works on
// the exact integer keys floor(d * 2^32 / N) of the few hundred elements it receives (rs_key).
#pragma once
#include <type_traits>
#include "hu_common.h"
#include "hu_kern_sep.h"

#define HU_RS_SMALL 512
#define HU_RS_THREADS 512
#ifndef HU_RS_U
#define HU_RS_U 4                 /* elements per thread and loop trip of the streaming passes: their loads are in flight together */
#endif
#ifndef HU_RS_WAVES_PER_EU
#define HU_RS_WAVES_PER_EU 6          /* 512 threads = 2 waves per SIMD and workgroup: three workgroups per CU need <= 85 VGPRs */
#endif
#define HU_RS_TRIP (HU_RS_THREADS * HU_RS_U)
#define HU_RS_FIN (HU_RS_SMALL + 96)        /* LDS array of the sequential finisher: the set-aside prefix (< 64 places) + the last range */

struct HuRsRange { int lo, hi, depth; };

#ifdef HU_RS_PROF     /* development only: cycles of thread 0 per phase, summed over the reads of every workgroup */
__device__ unsigned long long g_rs_prof[16];
#define RS_T(i) do { if(tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_rs_prof[i], now_ - t_prof); t_prof = now_; } } while(0)
#else
#define RS_T(i) do { } while(0)
#endif

__device__ inline uint64_t rs_lane_lt(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }
__device__ inline uint64_t rs_lane_ge(int lane) { return ~0ull << lane; }

/* An element is (d << 48 | N << 32 | node) in registers: the pair as the scan left it beside its position in node order.  dist = d / N as the
 * reference computes it is a correctly rounded double division of two integers below 2^16: equal fractions give equal doubles and different
 * fractions different ones (they differ by at least 2^-32 relative), so dist(a) < dist(b) <=> d_a * N_b < d_b * N_a — exact in 32-bit integers,
 * and no division in the streaming passes.  The sequential finisher works on keys (rs_key) of the at most HU_RS_FIN elements it receives.
 * In memory a level is two arrays — the pairs in the width the scan wrote them (2 or 4 bytes) and the node ids (4 bytes): the counting pass
 * reads the pairs only. */
__device__ inline bool rs_ltp(uint32_t pa, uint32_t pb) { return (pa >> 16) * (pb & 0xffffu) < (pb >> 16) * (pa & 0xffffu); }
__device__ inline bool rs_lt(uint64_t a, uint64_t b) { return rs_ltp((uint32_t)(a >> 32), (uint32_t)(b >> 32)); }
template<class PT> struct HuRsLevel {          /* one level in the workgroup's scratch: [cap] ids, [cap] pairs; positions are absolute */
	uint32_t* ids; PT* keys;
	__device__ inline uint64_t load(int p) const { return ((uint64_t) HuPair<PT>::canon(keys[p]) << 32) | (uint64_t) ids[p]; }
	__device__ inline void store(int p, uint64_t e) const { keys[p] = HuPair<PT>::pack((uint32_t)(e >> 32)); ids[p] = (uint32_t) e; }
};
/* The streaming passes load through rs_pair / rs_node: no branch between the loads of one trip (the level is a template argument, the two
 * patches of level 0 are selects), so that the loads of a thread are in flight together — behind a divergent branch each one is waited for
 * before the next is issued.  Level 0 is the row itself in node order without the root. */
template<bool L0, class PT>
__device__ inline uint32_t rs_pair(const HuRsLevel<PT>& src, const PT* __restrict__ row, int root, int p, int pA, uint64_t vA, int pB, uint64_t vB, bool& nan) {
	if(!L0) return HuPair<PT>::canon(src.keys[p]);
	uint32_t pr = HuPair<PT>::canon(row[p < root ? p : p + 1]);
	nan |= (pr & 0xffffu) == 0;
	pr = p == pA ? (uint32_t)(vA >> 32) : pr;
	pr = p == pB ? (uint32_t)(vB >> 32) : pr;
	return pr;
}
template<bool L0, class PT>
__device__ inline uint32_t rs_node(const HuRsLevel<PT>& src, int root, int p, int pA, uint64_t vA, int pB, uint64_t vB) {
	if(!L0) return src.ids[p];
	uint32_t id = (uint32_t)(p < root ? p : p + 1);
	id = p == pA ? (uint32_t) vA : id;
	id = p == pB ? (uint32_t) vB : id;
	return id;
}
/* key of an element for the finisher: floor(d * 2^32 / N) above the node id.  Exact: d <= N < 2^16, so the key has 33 bits, and two different
 * fractions differ by at least 1 / (N_a N_b) > 2^-32 — their keys differ; equal fractions have equal keys.  One 64-bit division per element
 * of the last range (<= HU_RS_FIN per read). */
#define HU_RS_IDBITS 31
__device__ __attribute__((noinline)) uint64_t rs_key(uint64_t e) {
	const uint32_t pr = (uint32_t)(e >> 32);
	return ((((uint64_t)(pr >> 16) << 32) / (uint64_t)(pr & 0xffffu)) << HU_RS_IDBITS) | (e & 0x7fffffffull);
}

/* scratch of one workgroup: two levels of cap positions (the m0 positions, then room for the right stoppers of a partition: at most half of
 * them are swapped), each a 4-byte id array and a pairBytes-wide pair array; in 8-byte words */
static inline size_t hu_refsort_cap(size_t m0) { return (((m0 + 63) & ~(size_t) 63) + m0 / 2 + 64 + 63) & ~(size_t) 63; }
static inline size_t hu_refsort_words(size_t m0, int pairBytes) { return 2 * hu_refsort_cap(m0) * (4 + (size_t) pairBytes) / 8; }
/* LDS bytes of k_seed_refsort for a tree of nNodes nodes */
static inline size_t hu_refsort_lds(int nNodes) { const size_t NT = ((size_t) nNodes - 1 + 63) / 64; return (HU_RS_FIN + 64) * 8 + 2 * (NT + 2) * 4; }

/* the literal sequential algorithm on an LDS array, for one thread: introsort loop restricted to ranges that start before place K */
__device__ __attribute__((noinline)) bool rs_seq_loop(uint64_t* a, int* stk /* LDS [72] */, int first, int last, int depth, int K) {
	constexpr int idBits = HU_RS_IDBITS;
	/* explicit stack of (first, last, depth): the recursion into [cut, last) happens only when cut < K */
	int* sf = stk; int* sl = stk + 24; int* sd = stk + 48; int sp = 0;
	sf[0] = first; sl[0] = last; sd[0] = depth; sp = 1;
	while(sp > 0) {
		--sp;
		int f = sf[sp], l = sl[sp], dp = sd[sp];
		while(l - f > 16) {
			if(dp == 0) return false;          /* heap-sort branch: left to the host */
			--dp;
			const int mid = f + (l - f) / 2;
			{ /* __move_median_to_first(f, f + 1, mid, l - 1) */
				const uint64_t x = a[f + 1] >> idBits, y = a[mid] >> idBits, z = a[l - 1] >> idBits;
				int w;
				if(x < y) { if(y < z) w = mid; else if(x < z) w = l - 1; else w = f + 1; }
				else if(x < z) w = f + 1;
				else if(y < z) w = l - 1;
				else w = mid;
				const uint64_t t = a[f]; a[f] = a[w]; a[w] = t;
			}
			const uint64_t pk = a[f] >> idBits;
			int i = f + 1, j = l;
			for(;;) {
				while((a[i] >> idBits) < pk) ++i;
				--j;
				while(pk < (a[j] >> idBits)) --j;
				if(!(i < j)) break;
				const uint64_t t = a[i]; a[i] = a[j]; a[j] = t;
				++i;
			}
			if(i < K) { if(sp >= 24) return false; sf[sp] = i; sl[sp] = l; sd[sp] = dp; ++sp; }
			l = i;
		}
	}
	return true;
}

template<class PT>
__global__ __launch_bounds__(HU_RS_THREADS, HU_RS_WAVES_PER_EU) void k_seed_refsort(HuDbDev db, const PT* __restrict__ pairs, int nReads,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, int K,
		unsigned long long* __restrict__ scratch, size_t cap, int rsOff,
		int32_t* __restrict__ seedCnt, int32_t* __restrict__ seedId, uint32_t* __restrict__ seedDN, uint32_t* __restrict__ parDN,
		int32_t* __restrict__ bail) {
	extern __shared__ unsigned char rs_smem[];
	const int m0 = db.nNodes - 1;
	const int NT = (m0 + 63) >> 6;
	uint64_t* fin = reinterpret_cast<uint64_t*>(rs_smem);                  /* [HU_RS_FIN]: the sequential finisher's array, by absolute position */
	uint64_t* lsb = fin + HU_RS_FIN;                                       /* [64]: left stoppers of a tiny left part */
	uint32_t* preL = reinterpret_cast<uint32_t*>(lsb + 64);                /* [NT + 2] */
	uint32_t* sufR = preL + (NT + 2);                                      /* [NT + 2] */
	__shared__ unsigned long long wtot[HU_RS_THREADS / 64];
	__shared__ uint64_t shE[4];              /* pivot element, patch values */
	__shared__ int shI[12];                  /* broadcast slots */
	__shared__ HuRsRange stash[12];
	__shared__ int seqStack[72];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef HU_RS_PROF
	unsigned long long t_prof = wall_clock64();
#endif
	/* reads are handed out through a counter (bail[1]): their cost differs with the pivots they meet, and the last ones do not wait for the
	 * slowest workgroup of a fixed schedule */
	for(;;) {
		__syncthreads();
		if(tid == 0) shI[6] = atomicAdd(&bail[1], 1);
		__syncthreads();
		const int read = shI[6];
		if(read >= nReads) break;
		RS_T(0);
		if(rend[read] < rstart[read] || m0 < 1) { if(tid == 0) seedCnt[read] = 0; continue; }
		const PT* __restrict__ row = pairs + (size_t) read * db.nNodesPad;
		HuRsLevel<PT> bufA, bufB;
		{
			unsigned char* base = reinterpret_cast<unsigned char*>(scratch) + (size_t) blockIdx.x * 2 * cap * (4 + sizeof(PT));
			bufA.ids = reinterpret_cast<uint32_t*>(base); bufA.keys = reinterpret_cast<PT*>(base + cap * 4);
			base += cap * (4 + sizeof(PT));
			bufB.ids = reinterpret_cast<uint32_t*>(base); bufB.keys = reinterpret_cast<PT*>(base + cap * 4);
		}
		HuRsLevel<PT> src = {nullptr, nullptr}, dst = bufA;       /* src.ids == nullptr: the implicit level-0 array */
		bool level0 = true;
		int lo = 0, hi = m0, depth = 0, nStash = 0;
		for(int x = m0; x > 1; x >>= 1) ++depth;
		depth *= 2;
		int pA = -1, pB = -1; uint64_t vA = 0, vB = 0;   /* level 0: the one swap of the pivot selection, kept as two patches */
		bool failed = false, nan = false;
		auto E = [&](int p) -> uint64_t {                /* any element, off the streaming passes */
			if(!level0) return src.load(p);
			return ((uint64_t) rs_pair<true>(src, row, db.root, p, pA, vA, pB, vB, nan) << 32) | (uint64_t) rs_node<true>(src, db.root, p, pA, vA, pB, vB);
		};
		while(hi - lo > HU_RS_SMALL) {
			if(depth == 0) { failed = true; break; }
			--depth;
			/* ---- pivot: median of lo + 1, mid, hi - 1 swapped into lo */
			if(tid == 0) {
				const int mid = lo + (hi - lo) / 2;
				const uint64_t ea = E(lo + 1), eb = E(mid), ec = E(hi - 1), ef = E(lo);
				int w; uint64_t ew;
				if(rs_lt(ea, eb)) { if(rs_lt(eb, ec)) { w = mid; ew = eb; } else if(rs_lt(ea, ec)) { w = hi - 1; ew = ec; } else { w = lo + 1; ew = ea; } }
				else if(rs_lt(ea, ec)) { w = lo + 1; ew = ea; }
				else if(rs_lt(eb, ec)) { w = hi - 1; ew = ec; }
				else { w = mid; ew = eb; }
				shE[0] = ew; shE[1] = ef; shI[0] = w;
				if(!level0) { src.store(lo, ew); src.store(w, ef); }
			}
			__threadfence_block();
			__syncthreads();
			RS_T(1);
			const uint64_t pivE = shE[0];
			if(level0) { pA = lo; vA = pivE; pB = shI[0]; vB = shE[1]; }
			const uint32_t pivP = (uint32_t)(pivE >> 32);
			const int M = hi - lo - 1, NTl = (M + 63) >> 6;
			/* ---- pass A: stoppers per subtile of 64 positions q = p - (lo + 1) */
			auto passA = [&](auto l0) {
				constexpr bool L0 = decltype(l0)::value;
				for(int base = 0; base < M; base += HU_RS_TRIP) {
					uint32_t kv[HU_RS_U];
#pragma unroll
					for(int u = 0; u < HU_RS_U; ++u) kv[u] = rs_pair<L0>(src, row, db.root, lo + 1 + min(base + u * HU_RS_THREADS + tid, M - 1), pA, vA, pB, vB, nan);
#pragma unroll
					for(int u = 0; u < HU_RS_U; ++u) {
						const int qb = base + u * HU_RS_THREADS, q = qb + tid;
						const bool valid = q < M;
						const unsigned long long mL = __ballot(valid && !rs_ltp(kv[u], pivP)), mR = __ballot(valid && !rs_ltp(pivP, kv[u]));
						if(lane == 0 && qb + wave * 64 < M) { preL[(qb >> 6) + wave] = (uint32_t) __popcll(mL); sufR[(qb >> 6) + wave] = (uint32_t) __popcll(mR); }
					}
				}
			};
			if(level0) passA(std::true_type{}); else passA(std::false_type{});
			if(__syncthreads_or(nan ? 1 : 0)) { failed = true; break; }
			RS_T(2);
			/* ---- scans: preL[t] = left stoppers before subtile t (exclusive), sufR[t] = right stoppers in subtiles >= t */
			{
				const int per = (NTl + HU_RS_THREADS - 1) / HU_RS_THREADS, a0 = tid * per, a1 = min(NTl, a0 + per);
				uint32_t sL = 0, sR = 0;
				for(int t = a0; t < a1; ++t) { sL += preL[t]; sR += sufR[t]; }
				/* both sums in one 64-bit word through one scan: inclusive within the wave on __shfl_up, the waves' totals through LDS */
				unsigned long long w = (unsigned long long) sL | ((unsigned long long) sR << 32);
#pragma unroll
				for(int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(w, o); if(lane >= o) w += t; }
				if(lane == 63) wtot[wave] = w;
				__syncthreads();
				unsigned long long before = 0, total = 0;
#pragma unroll
				for(int x = 0; x < HU_RS_THREADS / 64; ++x) { const unsigned long long t = wtot[x]; if(x < wave) before += t; total += t; }
				w += before;                                             /* inclusive prefix over the threads */
				uint32_t accL = (uint32_t) w - sL;                       /* left stoppers in the chunks before this thread's */
				uint32_t accR = (uint32_t)(total >> 32) - (uint32_t)(w >> 32);     /* right stoppers in the chunks after it */
				if(tid == 0) shI[1] = (int)(uint32_t) total;
				for(int t = a0; t < a1; ++t) { const uint32_t v = preL[t]; preL[t] = accL; accL += v; }
				for(int t = a1 - 1; t >= a0; --t) { accR += sufR[t]; sufR[t] = accR; }
				if(tid == 0) { preL[NTl] = (uint32_t) shI[1]; sufR[NTl] = 0; }
				__syncthreads();
			}
			RS_T(3);
			/* ---- cut and m (wave 0).  g(q) = L(q) - R(q + 1): left stoppers before q minus right stoppers after q, non-decreasing in q;
			 * the swapped pairs are the left stoppers with g < 0; c0 = the first position with g >= 0 */
			if(wave == 0) {
				/* first subtile t0 with preL[t0 + 1] >= sufR[t0 + 1] (positions of earlier subtiles all have g < 0) */
				int t0;
				{ int a = 0, b = NTl - 1; while(a < b) { const int md = (a + b) >> 1; if(preL[md + 1] >= sufR[md + 1]) b = md; else a = md + 1; } t0 = a; }
				auto masks = [&](int t, unsigned long long& mL, unsigned long long& mR) {
					const int q = t * 64 + lane; const bool valid = q < M;
					const uint32_t k = valid ? (uint32_t)(E(lo + 1 + q) >> 32) : 0;
					mL = __ballot(valid && !rs_ltp(k, pivP)); mR = __ballot(valid && !rs_ltp(pivP, k));
				};
				unsigned long long mL, mR;
				masks(t0, mL, mR);
				int c0;
				{
					const int Lq = (int) preL[t0] + __popcll(mL & rs_lane_lt(lane));
					const int Rq = (int) sufR[t0 + 1] + __popcll(mR & rs_lane_ge(lane) & ~(1ull << lane));
					const unsigned long long ok = __ballot(t0 * 64 + lane < M && Lq - Rq >= 0);
					c0 = ok ? t0 * 64 + (__ffsll((long long) ok) - 1) : (t0 + 1) * 64;
				}
				int tc = c0 >> 6, m, iNext = -1;                     /* iNext: i_(m+1), the first left stopper at or after c0 (-1: none) */
				if(tc < NTl) {
					if(tc != t0) masks(tc, mL, mR);
					m = (int) preL[tc] + __popcll(mL & rs_lane_lt(c0 & 63));
					unsigned long long cand = mL & rs_lane_ge(c0 & 63);
					int t = tc;
					while(!cand && ++t < NTl) { if(preL[t + 1] > preL[t]) { unsigned long long x, y; masks(t, x, y); cand = x; } }
					if(cand) iNext = t * 64 + (__ffsll((long long) cand) - 1);
				}
				else m = (int) preL[NTl];
				int jm = -1;                                         /* j_m: the m-th right stopper from the right */
				if(m >= 1) {
					int a = 0, b = NTl - 1;                           /* last subtile t with sufR[t] >= m */
					while(a < b) { const int md = (a + b + 1) >> 1; if((int) sufR[md] >= m) a = md; else b = md - 1; }
					unsigned long long x, y; masks(a, x, y);
					const int want = m - (int) sufR[a + 1];
					const unsigned long long hit = __ballot(((y >> lane) & 1ull) && __popcll(y & rs_lane_ge(lane)) == want);
					jm = a * 64 + (__ffsll((long long) hit) - 1);
				}
				const int cutq = (m >= 1 && (iNext < 0 || iNext > jm)) ? jm : iNext;
				if(lane == 0) { shI[2] = cutq; shI[3] = m; shI[4] = jm; }
			}
			__syncthreads();
			RS_T(4);
			const int cutq = shI[2], m = shI[3], jm = shI[4];
			if(cutq < 0) { failed = true; break; }                  /* no stopper where the sentinels guarantee one: not reached on consistent data */
			const int cutAbs = lo + 1 + cutq;
			const HuRsLevel<PT> RS = {dst.ids + rsOff, dst.keys + rsOff};        /* right stoppers by rank, beyond the positions */
			/* ---- pass B1: the right stoppers j_1 .. j_m (rank from the right <= m), from the subtile of j_m on */
			auto passB1 = [&](auto l0) {
				constexpr bool L0 = decltype(l0)::value;
				for(int base = (jm >> 6) * 64 / HU_RS_THREADS * HU_RS_THREADS; base < M; base += HU_RS_TRIP) {
					uint32_t kv[HU_RS_U], iv[HU_RS_U];
#pragma unroll
					for(int u = 0; u < HU_RS_U; ++u) {
						const int p = lo + 1 + min(base + u * HU_RS_THREADS + tid, M - 1);
						kv[u] = rs_pair<L0>(src, row, db.root, p, pA, vA, pB, vB, nan);
						iv[u] = rs_node<L0>(src, db.root, p, pA, vA, pB, vB);
					}
#pragma unroll
					for(int u = 0; u < HU_RS_U; ++u) {
						const int q = base + u * HU_RS_THREADS + tid, t = q >> 6;
						const bool rs = q < M && !rs_ltp(pivP, kv[u]);
						const unsigned long long mR = __ballot(rs);
						if(rs) { const int rk = (int) sufR[t + 1] + __popcll(mR & rs_lane_ge(lane)); if(rk <= m) { RS.keys[rk - 1] = HuPair<PT>::pack(kv[u]); RS.ids[rk - 1] = iv[u]; } }
					}
				}
			};
			if(m >= 1) { if(level0) passB1(std::true_type{}); else passB1(std::false_type{}); }
			__threadfence_block();
			__syncthreads();
			RS_T(5);
			const bool tiny = cutAbs < K;        /* the left part ends inside the first K places: the right part is needed too */
			if(!tiny) {
				/* ---- pass B2: the left part [lo, cut) of the next level */
				if(tid == 0) dst.store(lo, pivE);
				auto passB2 = [&](auto l0) {
					constexpr bool L0 = decltype(l0)::value;
					for(int base = 0; base < cutq; base += HU_RS_TRIP) {
						uint32_t kv[HU_RS_U], iv[HU_RS_U]; int rr[HU_RS_U];
#pragma unroll
						for(int u = 0; u < HU_RS_U; ++u) {
							const int p = lo + 1 + min(base + u * HU_RS_THREADS + tid, M - 1);
							kv[u] = rs_pair<L0>(src, row, db.root, p, pA, vA, pB, vB, nan);
							iv[u] = rs_node<L0>(src, db.root, p, pA, vA, pB, vB);
						}
#pragma unroll
						for(int u = 0; u < HU_RS_U; ++u) {       /* a left stopper takes the right stopper of its rank */
							const int q = base + u * HU_RS_THREADS + tid, t = q >> 6;
							const bool ls = q < cutq && !rs_ltp(kv[u], pivP);
							const unsigned long long mL = __ballot(ls);
							rr[u] = ls ? (int) preL[min(t, NTl)] + __popcll(mL & rs_lane_lt(lane)) : -1;
						}
#pragma unroll
						for(int u = 0; u < HU_RS_U; ++u) {       /* the others read RS[0] (one address) and keep their own: no branch around the loads */
							const int ri = max(rr[u], 0);
							const uint32_t rk = HuPair<PT>::canon(RS.keys[ri]), rid = RS.ids[ri];
							kv[u] = rr[u] >= 0 ? rk : kv[u]; iv[u] = rr[u] >= 0 ? rid : iv[u];
						}
#pragma unroll
						for(int u = 0; u < HU_RS_U; ++u) {
							const int q = base + u * HU_RS_THREADS + tid;
							if(q < cutq) { dst.keys[lo + 1 + q] = HuPair<PT>::pack(kv[u]); dst.ids[lo + 1 + q] = iv[u]; }
						}
					}
				};
				if(level0) passB2(std::true_type{}); else passB2(std::false_type{});
				hi = cutAbs;
			}
			else {
				/* the tiny left part goes to the finisher's array as it is after this partition; its left stoppers are kept for the right part */
				if(tid < 64) {
					const int q = tid; const bool valid = q < cutq;
					const uint64_t e = valid ? E(lo + 1 + q) : 0;
					const bool ls = valid && !rs_lt(e, pivE);
					const unsigned long long mL = __ballot(ls);
					const int k = __popcll(mL & rs_lane_lt(lane));
					if(ls) lsb[k] = e;
					if(valid) fin[lo + 1 + q] = rs_key(ls ? RS.load(k) : e);
					if(tid == 0) { fin[lo] = rs_key(pivE); if(nStash < 12) { stash[nStash].lo = lo; stash[nStash].hi = cutAbs; stash[nStash].depth = depth; } }
				}
				__syncthreads();
				if(nStash >= 12) { failed = true; break; }
				++nStash;
				/* the right part [cut, hi): a right stopper of rank k <= m receives the k-th left stopper (the element that sat at i_k) */
				for(int base = cutq / HU_RS_THREADS * HU_RS_THREADS; base < M; base += HU_RS_THREADS) {
					const int q = base + tid, t = q >> 6;
					const bool valid = q < M;
					const uint64_t e = valid ? E(lo + 1 + q) : 0;
					const bool rs = valid && !rs_lt(pivE, e);
					const unsigned long long mR = __ballot(rs);
					if(valid && q >= cutq) {
						const int rk = rs ? (int) sufR[t + 1] + __popcll(mR & rs_lane_ge(lane)) : 0;
						dst.store(lo + 1 + q, (rs && rk <= m) ? lsb[rk - 1] : e);
					}
				}
				lo = cutAbs;
			}
			__threadfence_block();
			__syncthreads();
			RS_T(6);
			src = dst; dst = (dst.ids == bufA.ids) ? bufB : bufA;
			level0 = false;
			pA = pB = -1;
		}
		if(__syncthreads_or((failed || nan) ? 1 : 0)) { /* the host path finishes this read */
			if(tid == 0) { seedCnt[read] = 0; const int at = atomicAdd(bail, 1); bail[2 + at] = read; }
			continue;
		}
		/* ---- the last range into LDS beside the set-aside prefix; one thread finishes with the literal algorithm */
		for(int p = lo + tid; p < hi; p += HU_RS_THREADS) { const uint64_t e = E(p); if(!nan) fin[p] = rs_key(e); }
		if(__syncthreads_or(nan ? 1 : 0)) { if(tid == 0) { seedCnt[read] = 0; const int at = atomicAdd(bail, 1); bail[2 + at] = read; } continue; }
		RS_T(7);
		if(tid == 0) {
			bool ok = true;
			for(int s = 0; s < nStash && ok; ++s) ok = rs_seq_loop(fin, seqStack, stash[s].lo, stash[s].hi, stash[s].depth, K);
			if(ok) ok = rs_seq_loop(fin, seqStack, lo, hi, depth, K);
			if(ok) { /* __final_insertion_sort over the blocks that hold the first K places */
				const int e = min(hi, K + 16);
				for(int i = 1; i < e; ++i) {
					const uint64_t v = fin[i]; int j = i;
					while(j > 0 && (v >> HU_RS_IDBITS) < (fin[j - 1] >> HU_RS_IDBITS)) { fin[j] = fin[j - 1]; --j; }
					fin[j] = v;
				}
			}
			shI[5] = ok ? 1 : 0;
		}
		__syncthreads();
		RS_T(8);
		if(!shI[5]) { if(tid == 0) { seedCnt[read] = 0; const int at = atomicAdd(bail, 1); bail[2 + at] = read; } continue; }
		const int keep = min(K, hi);
		if(tid == 0) seedCnt[read] = keep;
		if(tid < keep) {
			const int node = (int)(fin[tid] & ((1ull << HU_RS_IDBITS) - 1));
			seedId[(size_t) read * HU_MAX_SEEDS + tid] = node;
			seedDN[(size_t) read * HU_MAX_SEEDS + tid] = HuPair<PT>::canon(row[node]);
			parDN[(size_t) read * HU_MAX_SEEDS + tid] = HuPair<PT>::canon(row[db.parent[node]]);
		}
	}
}
