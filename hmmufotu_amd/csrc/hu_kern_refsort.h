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
// are prefix counts, so a partition is data-parallel: stopper masks per 64-element subtile (ballots), a scan of their counts, cut and m,
// the right stoppers j_1 .. j_m copied out by rank (pass B1), and the left part [first, cut) of the next level written with its left
// stoppers replaced by them (pass B2).  Only the left part is written: the right part matters only when cut falls inside the first K
// places — then the (tiny) left part is set aside and the right part is written instead.
//
// Second form (round 3, second half).  A level carries the KEYS ONLY (the (d, N) pair, 2 or 4 bytes): the first form moved a 4-byte node id
// beside every key through every level — two thirds of its bytes — although only the K elements that end in the first places ever need
// theirs.  Instead every level leaves its stopper masks and their scanned counts behind (24 bytes per 64 elements), and the K survivors are
// TRACED BACK through the levels afterwards: a position of the array after a partition came from the same position (not a stopper), or from
// the partner of its rank (select on the other mask through the scanned counts), then through the pivot swap.  And the count pass of a level
// is fused into the pass that writes it: the next pivot is the median of three elements of an array that is not written yet, but each of
// them is either an element of the current array or a right stopper of known rank — three lookups; pass B2 then classifies what it writes
// against that pivot.  Only level 0 (the pair row itself) and a level after a tiny left part run a counting pass of their own (pass A).
// Ranges of at most HU_RS_SMALL elements are finished by one thread in LDS with the literal sequential algorithm.  Reads the device does
// not finish (a NaN distance, the heap-sort branch of introsort, tables that overflow) are listed for the host path.
//
// No floating point anywhere: dist = d / N is compared through d_a * N_b < d_b * N_a in the streaming passes, and the sequential finisher works on
// the exact integer keys floor(d * 2^32 / N) of the few hundred elements it receives (rs_key).
#pragma once
#include <type_traits>
#include "hu_common.h"
#include "hu_kern_sep.h"

#define HU_RS_SMALL 128
/* Geometry (round 4): workgroups of 256 threads, SIX per CU (<= 85 VGPRs by the launch bounds, ~26 KB of LDS each).  What this kernel costs the other batches in
 * flight is the LDS / register share of its resident workgroups over their life, and a read's life is mostly dependent memory round trips (pivot look-ups, the cut's
 * searches, level after level) that more threads do not shorten: half the threads per read and twice the reads per CU gave 365 k -> 377 k reads/s at cfg3 against
 * three workgroups of 512 threads with 45 KB (the streaming levels' rank tables had to leave the LDS for that, hu_refsort_big).  Measured beside it (same run,
 * `gpurun_out/b_r4l_*`, `b_r4m_*`): 512 x 3 with the tables in global memory 357 k, 512 x 2 with a 40 KB region 354 k, 256 x 5 with 20 KB 373 k, 256 x 8 with 10 KB
 * (64 VGPRs) 367 k, three vector loads in flight per thread instead of two 366 k (spills). */
#ifndef HU_RS_THREADS
#define HU_RS_THREADS 256
#endif
#ifndef HU_RS_WAVES_PER_EU
#define HU_RS_WAVES_PER_EU 6          /* 256 threads = one wave per SIMD and workgroup: six workgroups per CU need <= 85 VGPRs */
#endif
#ifndef HU_RS_WGS_PER_CU
#define HU_RS_WGS_PER_CU (HU_RS_WAVES_PER_EU * 256 / HU_RS_THREADS)      /* resident workgroups per CU the launch bounds allow (the LDS must hold them too) */
#endif
#define HU_RS_FIN (HU_RS_SMALL + 96)        /* LDS array of the sequential finisher: the set-aside prefix (< 64 places) + the last range */
#define HU_RS_MAXLEV 64                     /* partitions of one read on the device; introsort's own limit is 2 lg n (<= 48 for n < 2^24) */

struct HuRsRange { int lo, hi, depth; };

#ifdef HU_RS_PROF     /* development only: cycles of thread 0 per phase, summed over the reads of every workgroup */
__device__ unsigned long long g_rs_prof[16 + 160];   /* [16 ..): the same per level (0 .. 8, 9 = after the levels) */
#define RS_T(i) do { if(tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_rs_prof[i], now_ - t_prof); atomicAdd(&g_rs_prof[16 + 10 * rs_lv + (i)], now_ - t_prof); t_prof = now_; } } while(0)
#else
#define RS_T(i) do { } while(0)
#endif

__device__ inline uint64_t rs_lane_lt(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }
__device__ inline uint64_t rs_lane_ge(int lane) { return ~0ull << lane; }
/* position of the n-th (1-based) set bit of m from bit 0; m holds at least n bits */
__device__ inline int rs_nth_low(uint64_t m, int n) {
	int pos = 0;
#pragma unroll
	for(int w = 32; w >= 1; w >>= 1) {
		const int c = __popcll(m & (((1ull << w) - 1ull) << pos));
		if(c < n) { n -= c; pos += w; }
	}
	return pos;
}
__device__ inline int rs_nth_high(uint64_t m, int n) { return 63 - rs_nth_low(__brevll(m), n); }

/* An element is its (d << 16 | N) pair as the scan left it.  dist = d / N as the reference computes it is a correctly rounded double division
 * of two integers below 2^16: equal fractions give equal doubles and different fractions different ones (they differ by at least 2^-32
 * relative), so dist(a) < dist(b) <=> d_a * N_b < d_b * N_a — exact in 32-bit integers, and no division in the streaming passes.  The
 * sequential finisher works on keys (rs_key) of the at most HU_RS_FIN elements it receives. */
__device__ inline bool rs_ltp(uint32_t pa, uint32_t pb) { return (pa >> 16) * (pb & 0xffffu) < (pb >> 16) * (pa & 0xffffu); }
/* key of an element for the finisher: floor(d * 2^32 / N) above a 31-bit tag (the level whose array the element sits in << 24 | its position
 * there; the trace-back turns it into a node).  Exact: d <= N < 2^16, so the key has 33 bits, and two different fractions differ by at least
 * 1 / (N_a N_b) > 2^-32 — their keys differ; equal fractions have equal keys.  One 64-bit division per element of the last range. */
#define HU_RS_IDBITS 31
__device__ __attribute__((noinline)) uint64_t rs_key(uint32_t pr, uint32_t tag) {
	return ((((uint64_t)(pr >> 16) << 32) / (uint64_t)(pr & 0xffffu)) << HU_RS_IDBITS) | (uint64_t) tag;
}

/* scratch of one workgroup.  Two key buffers of cap positions (the m0 positions, then room for the right stoppers of a partition: at most half
 * of them are swapped); the tables of all levels: per 64-element subtile the scanned counts (2 x 4 bytes) and the two stopper masks (2 x 8
 * bytes), tabCap subtiles in all (a level takes its subtiles + 1 rounded up to 64; levels shrink geometrically on anything but adversarial
 * rows, and a read whose tables do not fit goes to the host path) */
static inline size_t hu_refsort_cap(size_t m0) { return (((m0 + 63) & ~(size_t) 63) + m0 / 2 + 128 + 63) & ~(size_t) 63; }
static inline size_t hu_refsort_tabcap(size_t m0) { const size_t nt = ((m0 + 63) / 64 + 64) & ~(size_t) 63; return 8 * nt + 2048; }
static inline size_t hu_refsort_words(size_t m0, int pairBytes) { return (((2 * hu_refsort_cap(m0) * (size_t) pairBytes + 7) / 8 + 3 * hu_refsort_tabcap(m0) + (hu_refsort_tabcap(m0) + 3) / 4) + 1) & ~(size_t) 1; }      /* + the stopper counts of a streaming level, 2 bytes per subtile */
/* LDS bytes of k_seed_refsort for a tree of nNodes nodes */
/* Round 4: the rank tables of a STREAMING level (scanned counts, 10 bytes per subtile: 31 KB for a gg_97-scale row) live in the workgroup's global scratch,
 * where the trace-back wanted its copy anyway; LDS keeps every 64th entry (cPre / cSuf) for the searches.  The region below holds only a range that fits,
 * with its tables — so a workgroup takes ~30 KB of LDS instead of 45, and its size no longer grows with the tree. */
#define HU_RS_LTAB 260                 /* subtiles of a range held in LDS, + 1 */
#define HU_RS_LHEAD (HU_RS_LTAB * (4 + 4 + 8 + 8 + 2) + 24)     /* = 6784: scanned counts, masks, counts of such a range; the keys follow */
#ifndef HU_RS_BIG
#define HU_RS_BIG 16384                /* bytes of that region: 4,608 places of 16-bit keys, 2,240 of 32-bit pairs (12 KB measured the same, 24 KB costs the sixth workgroup) */
#endif
__host__ __device__ static inline size_t hu_refsort_big(size_t) { return HU_RS_BIG; }
/* places a range may hold to live in LDS (the kernel's LCAP); a tree with more places than this starts with a streaming level on the pair row */
template<class PT> __host__ __device__ static inline int hu_refsort_lcap(size_t m0) {
	const int c = ((((int) hu_refsort_big(m0) - HU_RS_LHEAD) / (int) sizeof(PT)) - 128) & ~63;
	return c < 16384 - 192 ? c : 16384 - 192;
}
static inline size_t hu_refsort_lds(int nNodes) {
	const size_t m0 = (size_t) nNodes - 1;
	return HU_RS_FIN * 8 + 64 * 4 + 2 * (hu_refsort_tabcap(m0) / 64) * 4 + hu_refsort_big(m0);
}

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

/* which of three elements __move_median_to_first picks: 0 = the one at first + 1, 1 = mid, 2 = last - 1 */
__device__ inline int rs_median3(uint32_t a, uint32_t b, uint32_t c) {
	if(rs_ltp(a, b)) { if(rs_ltp(b, c)) return 1; else if(rs_ltp(a, c)) return 2; else return 0; }
	else if(rs_ltp(a, c)) return 0;
	else if(rs_ltp(b, c)) return 2;
	return 1;
}

/* ---- the streaming passes work on GROUPS: 16 bytes of keys per lane and load (8 keys of 2 bytes, 4 of 4), so that a trip of the workgroup
 * has 16 KB in flight instead of 2 (a pass of one key per lane was bound by its round trips: 2 GB/s per workgroup).  Subtiles are aligned to
 * ABSOLUTE positions (p >> 6), so every level — the pair row included — is read and written in aligned vectors; positions of a subtile outside
 * (lo, hi) simply have no mask bit.  A subtile is 64 / EPL lanes; ranks inside it come from a scan over those lanes. */
template<class PT> struct HuRsGeom { static constexpr int EPL = 16 / (int) sizeof(PT); static constexpr int LPS = 64 / EPL; };
#ifndef HU_RS_VU
#define HU_RS_VU 2                /* vector loads in flight per thread */
#endif
#ifndef HU_RS_VU_B1
#define HU_RS_VU_B1 HU_RS_VU
#endif
#ifndef HU_RS_VU_B2
#define HU_RS_VU_B2 HU_RS_VU
#endif
struct __attribute__((packed, aligned(2))) HuRsU4h { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) HuRsU4w { uint32_t x, y, z, w; };

/* The streaming passes work on the keys AS STORED ("raw": the 16-bit form d << 8 | N, or the 32-bit pair itself) — what a pass moves it moves unchanged, and a
 * comparison takes d and N straight out of the stored form.  Round 3 widened every key to the canonical 32-bit pair on load and narrowed it again on store: 7 of
 * the ~30 vector instructions a key cost per visit.  The whole path is bound by vector ISSUE (six batches in flight: this kernel's time hides behind the other
 * batches' kernels, its instructions do not — resident workgroups 768 -> 256 double its time and leave the step rate where it was), so instructions are what counts. */
template<class PT> struct HuRsRaw;
template<> struct HuRsRaw<uint16_t> {
	__device__ static inline uint32_t d(uint32_t x) { return x >> 8; }
	__device__ static inline uint32_t N(uint32_t x) { return x & 0xffu; }
	__device__ static inline void unpack(const uint4& v, uint32_t* o) {
		o[0] = v.x & 0xffffu; o[1] = v.x >> 16; o[2] = v.y & 0xffffu; o[3] = v.y >> 16; o[4] = v.z & 0xffffu; o[5] = v.z >> 16; o[6] = v.w & 0xffffu; o[7] = v.w >> 16;
	}
};
template<> struct HuRsRaw<uint32_t> {
	__device__ static inline uint32_t d(uint32_t x) { return x >> 16; }
	__device__ static inline uint32_t N(uint32_t x) { return x & 0xffffu; }
	__device__ static inline void unpack(const uint4& v, uint32_t* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};
template<bool L0, class PT>
__device__ inline void rs_load(const PT* __restrict__ src, const PT* __restrict__ row, int root, int rowLast, int p0, uint32_t (&k)[HuRsGeom<PT>::EPL]) {
	constexpr int EPL = HuRsGeom<PT>::EPL;
	if(!L0) { const uint4 v = *reinterpret_cast<const uint4*>(src + p0); HuRsRaw<PT>::unpack(v, k); return; }
	/* level 0: the row in node order without the root — from the root on, the element of place p is row[p + 1] */
	uint32_t o[EPL + 1];
	const uint4 v = *reinterpret_cast<const uint4*>(row + p0);
	HuRsRaw<PT>::unpack(v, o);
	o[EPL] = (uint32_t) row[min(p0 + EPL, rowLast)];
	if(p0 >= root) {
#pragma unroll
		for(int e = 0; e < EPL; ++e) k[e] = o[e + 1];
	}
	else {
#pragma unroll
		for(int e = 0; e < EPL; ++e) k[e] = (p0 + e < root) ? o[e] : o[e + 1];
	}
}
/* level 0 only: the two patches of the pivot's swap (raw values); a compared-site count of zero among the valid places is a NaN distance */
template<class PT, int EPL>
__device__ inline void rs_fix_l0(uint32_t (&k)[EPL], int p0, int pA, uint32_t vA, int pB, uint32_t vB, uint32_t vb, bool& nan) {
	if((unsigned)(pA - p0) < (unsigned) EPL || (unsigned)(pB - p0) < (unsigned) EPL) {
#pragma unroll
		for(int e = 0; e < EPL; ++e) { k[e] = p0 + e == pA ? vA : k[e]; k[e] = p0 + e == pB ? vB : k[e]; }
	}
	if(vb) {
		uint32_t z = 0;
#pragma unroll
		for(int e = 0; e < EPL; ++e) z |= (uint32_t)(HuRsRaw<PT>::N(k[e]) == 0) << e;
		nan |= (z & vb) != 0;
	}
}
template<int EPL>
__device__ inline uint32_t rs_valid(int q0, int qlo, int qhi) {
	const int a = min(max(qlo - q0, 0), EPL), b = min(max(qhi - q0, 0), EPL);
	return ((1u << b) - 1u) & ~((1u << a) - 1u);
}
/* stopper bits of a group of raw keys against a pivot (canonical pair): left stopper = !(k < piv), right stopper = !(piv < k), both through the cross products */
template<class PT, int EPL>
__device__ inline void rs_classify(const uint32_t (&k)[EPL], uint32_t piv, uint32_t vb, uint32_t& mLb, uint32_t& mRb) {
	const uint32_t dp = piv >> 16, Np = piv & 0xffffu;
	uint32_t l = 0, r = 0;
#pragma unroll
	for(int e = 0; e < EPL; ++e) { const uint32_t a = HuRsRaw<PT>::d(k[e]) * Np, b = dp * HuRsRaw<PT>::N(k[e]); l |= (uint32_t)(a >= b) << e; r |= (uint32_t)(b >= a) << e; }
	mLb = l & vb; mRb = r & vb;
}
/* a group's stopper bits as the pass that classified the level left them in its mask arrays (rs_emit, or the words the scan's masks became) */
template<int EPL>
__device__ inline uint32_t rs_mask_bits(const unsigned char* __restrict__ bits, int g) {
	return EPL == 8 ? (uint32_t) bits[g] : (((uint32_t) bits[g >> 1] >> ((g & 1) * 4)) & 15u);
}
/* a group's bits into the level's mask arrays (as bytes: the arrays are bit arrays over the positions) and its subtile's counts */
template<int EPL>
__device__ inline void rs_emit(uint32_t mLb, uint32_t mRb, int g, bool store, int lane, unsigned char* bL, unsigned char* bR, uint16_t* cnt16) {
	constexpr int LPS = 64 / EPL;
	if(EPL == 8) { if(store) { bL[g] = (unsigned char) mLb; bR[g] = (unsigned char) mRb; } }
	else { const uint32_t o = __shfl_down(mLb | (mRb << 4), 1); if(store && !(lane & 1)) { bL[g >> 1] = (unsigned char)(mLb | ((o & 15u) << 4)); bR[g >> 1] = (unsigned char)(mRb | ((o >> 4) << 4)); } }
	uint32_t c = __popc(mLb) | (__popc(mRb) << 16);
#pragma unroll
	for(int o = 1; o < LPS; o <<= 1) c += __shfl_xor(c, o);
	if(store && (lane & (LPS - 1)) == 0) cnt16[g / LPS] = (uint16_t)((c & 0xffu) | ((c >> 16) << 8));
}

/* Level 0 of a read whose pair row is a streaming level, prepared BEFORE the scan: the pivot of introsort's first partition is the median of the
 * elements at places 1, mid and last - 1 (__move_median_to_first), swapped with the element at place 0 — four FIXED nodes, whatever the read.  Their
 * pairs straight from the bit-planes (pair_exact: what the scan will write for them), one thread per read:
 *     piv[read * 4] = {pivot pair, the pair that stood at place 0, the place the pivot came from, flags (bit 0: a compared-site count of zero, set by the scan)}
 * With these the scan classifies every pair it writes (k_seed_pdist2<PT, true>) and k_seed_refsort starts from the masks. */
__global__ __launch_bounds__(64) void k_ref_pivots(HuDbDev db, HuReadPlanes R, int n, uint32_t* __restrict__ piv) {
	const int r = blockIdx.x * 64 + threadIdx.x;
	if(r >= n) return;
	const int m0 = db.nNodes - 1, lo = 0, hi = m0, mid = lo + (hi - lo) / 2;
	auto nodeOf = [&](int p) { return p < db.root ? p : p + 1; };
	const uint32_t ea = pair_exact(db, R, r, nodeOf(lo + 1)), eb = pair_exact(db, R, r, nodeOf(mid)), ec = pair_exact(db, R, r, nodeOf(hi - 1)), ef = pair_exact(db, R, r, nodeOf(lo));
	const int c = rs_median3(ea, eb, ec);
	piv[(size_t) r * 4 + 0] = c == 0 ? ea : (c == 1 ? eb : ec);
	piv[(size_t) r * 4 + 1] = ef;
	piv[(size_t) r * 4 + 2] = (uint32_t)(c == 0 ? lo + 1 : (c == 1 ? mid : hi - 1));
	piv[(size_t) r * 4 + 3] = ((ea & 0xffffu) == 0 || (eb & 0xffffu) == 0 || (ec & 0xffffu) == 0 || (ef & 0xffffu) == 0) ? 1u : 0u;
}

template<class PT>
__global__ __launch_bounds__(HU_RS_THREADS, HU_RS_WAVES_PER_EU) void k_seed_refsort(HuDbDev db, const PT* __restrict__ pairs, int nReads,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, int K,
		unsigned long long* __restrict__ scratch, size_t wgWords, size_t cap, int rsOff, int tabCap,
		int32_t* __restrict__ seedCnt, int32_t* __restrict__ seedId, uint32_t* __restrict__ seedDN, uint32_t* __restrict__ parDN,
		int32_t* __restrict__ bail, const uint32_t* __restrict__ l0piv = nullptr, const unsigned long long* __restrict__ l0m = nullptr) {
	constexpr int EPL = HuRsGeom<PT>::EPL, LPS = HuRsGeom<PT>::LPS, VU = HU_RS_VU, VU1 = HU_RS_VU_B1, VU2 = HU_RS_VU_B2;
	extern __shared__ unsigned char rs_smem[];
	const int m0 = db.nNodes - 1;
	const int NT = ((m0 + 63) >> 6) + 1;
	const int NC = tabCap >> 6;
	uint64_t* fin = reinterpret_cast<uint64_t*>(rs_smem);                  /* [HU_RS_FIN]: the sequential finisher's array, by absolute position */
	uint32_t* lsb = reinterpret_cast<uint32_t*>(fin + HU_RS_FIN);          /* [64]: left stoppers of a tiny left part */
	uint32_t* cPre = lsb + 64;                                             /* [NC]: every 64th entry of every level's scanned counts (trace-back) */
	uint32_t* cSuf = cPre + NC;                                            /* [NC] */
	unsigned char* big = reinterpret_cast<unsigned char*>(cSuf + NC);      /* hu_refsort_big bytes, 16-byte aligned */
	/* streaming levels keep their tables in the global scratch (gPre / gSuf / gCnt below) */
	/* a range that fits (at most LCAP places): its tables, its masks and the range itself, partitioned in place */
	uint32_t* const preLs = reinterpret_cast<uint32_t*>(big); uint32_t* const sufRs = preLs + HU_RS_LTAB;
	unsigned long long* const mLs = reinterpret_cast<unsigned long long*>(sufRs + HU_RS_LTAB); unsigned long long* const mRs = mLs + HU_RS_LTAB;
	uint16_t* const cnt16s = reinterpret_cast<uint16_t*>(mRs + HU_RS_LTAB);
	PT* const lk = reinterpret_cast<PT*>(big + HU_RS_LHEAD);               /* place p at lk[p - lbase + 64] */
	const int LCAP = hu_refsort_lcap<PT>((size_t) m0);
	__shared__ unsigned long long wtot[HU_RS_THREADS / 64];
	__shared__ uint32_t shP[4];              /* pivot pair, the pair it displaced, next pivot pair */
	__shared__ int shI[12];                  /* broadcast slots */
	__shared__ HuRsRange stash[12];
	__shared__ int seqStack[72 * (HU_RS_THREADS / 64)];      /* one stack per wave: the sequential finishers of a read's stretches run side by side */
	__shared__ int hLo[HU_RS_MAXLEV], hQB[HU_RS_MAXLEV], hW[HU_RS_MAXLEV], hCut[HU_RS_MAXLEV], hM[HU_RS_MAXLEV], hOff[HU_RS_MAXLEV], hNC[HU_RS_MAXLEV];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int rowLast = db.nNodesPad - 1;
	/* this workgroup's scratch */
	PT* bufA; PT* bufB; uint32_t* gPre; uint32_t* gSuf; unsigned long long* gML; unsigned long long* gMR; uint16_t* gCnt;
	{
		unsigned long long* base = scratch + (size_t) blockIdx.x * wgWords;
		gML = base; gMR = base + tabCap;
		gPre = reinterpret_cast<uint32_t*>(base + 2 * (size_t) tabCap); gSuf = gPre + tabCap;
		gCnt = reinterpret_cast<uint16_t*>(base + 3 * (size_t) tabCap);      /* [tabCap]: stoppers per subtile of the streaming level being counted, left | right << 8 */
		bufA = reinterpret_cast<PT*>(base + 3 * (size_t) tabCap + ((size_t) tabCap + 3) / 4); bufB = bufA + cap;
	}
#ifdef HU_RS_PROF
	unsigned long long t_prof = wall_clock64(); int rs_lv = 9;
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
		const PT* src = nullptr; PT* dst = bufA;              /* src == nullptr: the implicit level-0 array */
		bool level0 = true;
		int lo = 0, hi = m0, depth = 0, nStash = 0, level = 0, tabNext = 0;
		for(int x = m0; x > 1; x >>= 1) ++depth;
		depth *= 2;
		int pA = -1, pB = -1; uint32_t vA = 0, vB = 0;   /* level 0: the one swap of the pivot selection, kept as two patches */
		bool failed = false, nan = false; int why = 0;        /* why a read is left to the host (bits 26.. of its entry in the list) */
		bool counted = false;                            /* this level's pivot, swap, masks and counts were made by the pass that wrote it */
		uint32_t pivP = 0; int wAbs = -1, off = 0;
		bool inLds = false; int lbase = 0;               /* the range lives in LDS from now on */
		uint32_t* preL = gPre; uint32_t* sufR = gSuf;      /* streaming levels: set to the level's stretch of the global tables before its scan */
#ifndef HU_RS_CNT_GLOBAL
		uint16_t* cnt16 = (size_t) NT * 2 <= (size_t) HU_RS_BIG ? reinterpret_cast<uint16_t*>(big) : gCnt;   /* the stopper counts of a streaming level live in the (still unused) LDS region of the
		                                                                                                        * later in-LDS ranges when they fit: the scan below reads each of them three times, one
		                                                                                                        * dependent round trip each from global memory (40 us of a gg_97-scale read's level 0) */
#else
		uint16_t* cnt16 = gCnt;
#endif
		auto E = [&](int p) -> uint32_t {                /* any one element, off the streaming passes */
			if(inLds) return HuPair<PT>::canon(lk[p - lbase + 64]);
			if(!level0) return HuPair<PT>::canon(src[p]);
			uint32_t pr = HuPair<PT>::canon(row[p < db.root ? p : p + 1]);
			nan |= (pr & 0xffffu) == 0;
			pr = p == pA ? vA : pr;
			return p == pB ? vB : pr;
		};
		for(;;) {
#ifdef HU_RS_PROF
			rs_lv = level < 9 ? level : 8;
#endif
			if(!inLds && hi - lo <= LCAP) {
				/* ---- the range fits: into LDS, where the remaining partitions run in place without a round trip to memory */
				lbase = (lo + 1) & ~63;
				for(int p = lo + tid; p < hi; p += HU_RS_THREADS) lk[p - lbase + 64] = HuPair<PT>::pack(E(p));
				if(__syncthreads_or(nan ? 1 : 0)) { failed = true; why = 3; break; }
				inLds = true; level0 = false; pA = pB = -1;
				preL = preLs; sufR = sufRs; cnt16 = cnt16s;
			}
			if(inLds && hi - lo <= HU_RS_SMALL) break;
			if(depth == 0 || level >= HU_RS_MAXLEV) { failed = true; why = 1; break; }
			--depth;
			/* valid places lo + 1 .. hi - 1; q = p - qBase on the grid of absolute subtiles, valid for o0 <= q < qEnd */
			const int M = hi - lo - 1, qBase = (lo + 1) & ~63, o0 = (lo + 1) & 63, qEnd = o0 + M, NTl = (qEnd + 63) >> 6;
			const int R = (NTl + 1 + 63) & ~63;             /* table entries of this level: 0 .. NTl, padded */
			const int gEnd = NTl * LPS;
			if(!counted) {
				off = tabNext;
				if(off + R > tabCap) { failed = true; why = 2; break; }
				tabNext = off + R;
				const bool fusedL0 = l0m != nullptr && level0 && level == 0 && !inLds;
				if(fusedL0) {
					/* ---- level 0 from the scan: pivot, swap and both stopper masks are there (k_ref_pivots, k_seed_pdist2<PT, true>).  The masks are in NODE order:
					 * place p is node p before the root and node p + 1 from it on, so from the root's subtile on a word of the place order is a funnel shift of
					 * two words of the node order.  Place lo holds the pivot (not part of the partition), place w the element that stood at lo. */
					pivP = l0piv[(size_t) read * 4]; const uint32_t ef = l0piv[(size_t) read * 4 + 1];
					wAbs = (int) l0piv[(size_t) read * 4 + 2]; nan |= (l0piv[(size_t) read * 4 + 3] & 1u) != 0;
					pA = lo; vA = pivP; pB = wAbs; vB = ef;
					RS_T(1);
					const int npw = db.nNodesPad >> 6, tr = db.root >> 6;
					const unsigned long long* nm = l0m + (size_t) read * npw * 2;
					const unsigned long long below = rs_lane_lt(db.root & 63);
					const bool efL = !rs_ltp(ef, pivP), efR = !rs_ltp(pivP, ef);
					for(int tb = 0; tb < NTl; tb += 4 * HU_RS_THREADS) { /* four subtiles per thread and trip: eight 16-byte loads in flight (one at a time, the six trips of a
					                                                      * gg_97-scale row cost a memory round trip each: 39 us of the read's 340) */
						ulonglong2 a4[4], c4[4];
#pragma unroll
						for(int u = 0; u < 4; ++u) {
							const int t = min(tb + u * HU_RS_THREADS + tid, NTl - 1);
							a4[u] = *reinterpret_cast<const ulonglong2*>(nm + 2 * (size_t) t);
							c4[u] = *reinterpret_cast<const ulonglong2*>(nm + 2 * (size_t) min(t + 1, npw - 1));
						}
#pragma unroll
						for(int u = 0; u < 4; ++u) {
							const int t = tb + u * HU_RS_THREADS + tid;
							if(t >= NTl) continue;
							const ulonglong2 a = a4[u];
							unsigned long long mL = a.x, mR = a.y;
							if(t >= tr) {
								ulonglong2 c = make_ulonglong2(0ull, 0ull);
								if(t + 1 < npw) c = c4[u];
								const unsigned long long sL = (a.x >> 1) | (c.x << 63), sR = (a.y >> 1) | (c.y << 63);
								mL = t > tr ? sL : ((a.x & below) | (sL & ~below));
								mR = t > tr ? sR : ((a.y & below) | (sR & ~below));
							}
							/* valid places of this subtile: o0 <= 64 t + i < qEnd */
							const int v0 = max(o0 - 64 * t, 0), v1 = min(qEnd - 64 * t, 64);
							const unsigned long long vm = (v1 >= 64 ? ~0ull : rs_lane_lt(max(v1, 0))) & ~rs_lane_lt(min(v0, 63)) & (v0 >= 64 ? 0ull : ~0ull);
							mL &= vm; mR &= vm;
							if((wAbs >> 6) == t && wAbs > lo) { const unsigned long long bit = 1ull << (wAbs & 63); mL = efL ? (mL | bit) : (mL & ~bit); mR = efR ? (mR | bit) : (mR & ~bit); }
							gML[off + t] = mL; gMR[off + t] = mR;
							cnt16[t] = (uint16_t)(__popcll(mL) | (__popcll(mR) << 8));
						}
					}
				}
				else {
				/* ---- pivot: median of lo + 1, mid, hi - 1 swapped into lo */
					if(tid == 0) {
						const int mid = lo + (hi - lo) / 2;
						const uint32_t ea = E(lo + 1), eb = E(mid), ec = E(hi - 1), ef = E(lo);
						const int c = rs_median3(ea, eb, ec);
						const int w = c == 0 ? lo + 1 : (c == 1 ? mid : hi - 1);
						const uint32_t ew = c == 0 ? ea : (c == 1 ? eb : ec);
						shP[0] = ew; shP[1] = ef; shI[0] = w;
						if(inLds) { lk[lo - lbase + 64] = HuPair<PT>::pack(ew); lk[w - lbase + 64] = HuPair<PT>::pack(ef); }
						else if(!level0) { PT* s = const_cast<PT*>(src); s[lo] = HuPair<PT>::pack(ew); s[w] = HuPair<PT>::pack(ef); }
					}
					__threadfence_block();
					__syncthreads();
					RS_T(1);
					pivP = shP[0]; wAbs = shI[0];
					if(level0) { pA = lo; vA = pivP; pB = wAbs; vB = shP[1]; }
					/* ---- pass A: stopper masks and counts per subtile */
					auto passA = [&](auto l0) {
						constexpr bool L0 = decltype(l0)::value;
						unsigned char* bL = reinterpret_cast<unsigned char*>(gML + off); unsigned char* bR = reinterpret_cast<unsigned char*>(gMR + off);
						for(int gb = 0; gb < gEnd; gb += HU_RS_THREADS * VU) {
							uint32_t k[VU][EPL];
#pragma unroll
							for(int u = 0; u < VU; ++u) rs_load<L0, PT>(src, row, db.root, rowLast, qBase + min(gb + u * HU_RS_THREADS + tid, gEnd - 1) * EPL, k[u]);
#pragma unroll
							for(int u = 0; u < VU; ++u) {
								const int g = gb + u * HU_RS_THREADS + tid; const bool in = g < gEnd;
								const uint32_t vb = in ? rs_valid<EPL>(g * EPL, o0, qEnd) : 0u;
								if(L0) rs_fix_l0<PT, EPL>(k[u], qBase + g * EPL, pA, HuPair<PT>::pack(vA), pB, HuPair<PT>::pack(vB), vb, nan);
								uint32_t mLb, mRb;
								rs_classify<PT, EPL>(k[u], pivP, vb, mLb, mRb);
								rs_emit<EPL>(mLb, mRb, g, in, lane, bL, bR, cnt16);
							}
						}
					};
					if(inLds) {
						for(int qb = wave * 64; qb < NTl * 64; qb += HU_RS_THREADS) {
							const int q = qb + lane, t = qb >> 6; const bool valid = q >= o0 && q < qEnd;
							const uint32_t e = valid ? HuPair<PT>::canon(lk[qBase + q - lbase + 64]) : 0u;
							const unsigned long long mL = __ballot(valid && !rs_ltp(e, pivP)), mR = __ballot(valid && !rs_ltp(pivP, e));
							if(lane == 0) { cnt16[t] = (uint16_t)(__popcll(mL) | (__popcll(mR) << 8)); mLs[t] = mL; mRs[t] = mR; gML[off + t] = mL; gMR[off + t] = mR; }
						}
					}
					else if(level0) passA(std::true_type{}); else passA(std::false_type{});
				}
				if(__syncthreads_or(nan ? 1 : 0)) { failed = true; why = 3; break; }
				RS_T(2);
			}
			counted = false;
			if(!inLds) { preL = gPre + off; sufR = gSuf + off; }
			/* ---- scans: preL[t] = left stoppers before subtile t (exclusive), sufR[t] = right stoppers in subtiles >= t */
			if(!inLds) {
				/* a streaming level: every wave takes a quarter of the subtiles in rows of 64 — counts in, tables out in whole lines (the chunk-per-thread form below
				 * wrote them 4 bytes at a stride of 52 and read them back for the coarse entries: 40 us of a gg_97-scale read's level 0) */
				constexpr int NW = HU_RS_THREADS / 64;
				const int Q = (((NTl + NW - 1) / NW) + 63) & ~63, q0 = wave * Q, q1 = min(NTl, q0 + Q);
				unsigned long long w = 0;                         /* left | right << 32 */
				for(int t = q0 + lane; t < q1; t += 64) { const uint32_t c = cnt16[t]; w += (unsigned long long)(c & 0xffu) | ((unsigned long long)(c >> 8) << 32); }
#pragma unroll
				for(int o = 32; o >= 1; o >>= 1) w += __shfl_xor(w, o);
				if(lane == 0) wtot[wave] = w;
				__syncthreads();
				unsigned long long base = 0, total = 0;
#pragma unroll
				for(int x = 0; x < NW; ++x) { const unsigned long long t = wtot[x]; if(x < wave) base += t; total += t; }
				const uint32_t totR = (uint32_t)(total >> 32);
				for(int tb = q0; tb < q1; tb += 64) {
					const int t = tb + lane; const bool in = t < q1;
					const uint32_t c = in ? cnt16[t] : 0u;
					const unsigned long long mine = (unsigned long long)(c & 0xffu) | ((unsigned long long)(c >> 8) << 32);
					unsigned long long inc = mine;
#pragma unroll
					for(int o = 1; o < 64; o <<= 1) { const unsigned long long v = __shfl_up(inc, o); if(lane >= o) inc += v; }
					const unsigned long long ex = base + inc - mine;
					const uint32_t vL = (uint32_t) ex, vR = totR - (uint32_t)(ex >> 32);
					if(in) { preL[t] = vL; sufR[t] = vR; if(lane == 0) { cPre[(off + t) >> 6] = vL; cSuf[(off + t) >> 6] = vR; } }
					base += __shfl(inc, 63);
				}
				/* entries NTl (the totals) .. R - 1, padded so that a block of 64 can be counted blindly */
				for(int t = NTl + tid; t < R; t += HU_RS_THREADS) {
					const uint32_t vL = t == NTl ? (uint32_t) total : 0xffffffffu, vR = 0u;
					preL[t] = vL; sufR[t] = vR;
					if((t & 63) == 0) { cPre[(off + t) >> 6] = vL; cSuf[(off + t) >> 6] = vR; }
				}
				__threadfence_block();
				__syncthreads();      /* the cut's searches start from the coarse entries */
			}
			else {
				const int per = (NTl + HU_RS_THREADS - 1) / HU_RS_THREADS, a0 = tid * per, a1 = min(NTl, a0 + per);
				uint32_t sL = 0, sR = 0;
				for(int t = a0; t < a1; ++t) { const uint32_t c = cnt16[t]; sL += c & 0xffu; sR += c >> 8; }
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
				for(int t = a0; t < a1; ++t) { const uint32_t v = cnt16[t] & 0xffu; preL[t] = accL; accL += v; }
				for(int t = a1 - 1; t >= a0; --t) { accR += cnt16[t] >> 8; sufR[t] = accR; }
				if(tid == 0) { preL[NTl] = (uint32_t) total; sufR[NTl] = 0; }
				__threadfence_block();
				__syncthreads();
				/* the level's tables for the trace-back: entries 0 .. NTl, padded so that a block of 64 can be counted blindly */
				for(int t = tid; t < R; t += HU_RS_THREADS) {
					const uint32_t vL = t <= NTl ? preL[t] : 0xffffffffu, vR = t <= NTl ? sufR[t] : 0u;
					gPre[off + t] = vL; gSuf[off + t] = vR;
					if((t & 63) == 0) { cPre[(off + t) >> 6] = vL; cSuf[(off + t) >> 6] = vR; }
				}
			}
			RS_T(3);
			/* ---- cut and m (wave 0).  g(q) = L(q) - R(q + 1): left stoppers before q minus right stoppers after q, non-decreasing in q;
			 * the swapped pairs are the left stoppers with g < 0; c0 = the first position with g >= 0 */
			auto ML = [&](int t) -> unsigned long long { return inLds ? mLs[t] : gML[off + t]; };
			auto MR = [&](int t) -> unsigned long long { return inLds ? mRs[t] : gMR[off + t]; };
			if(wave == 0) {
				/* first subtile t0 with preL[t0 + 1] >= sufR[t0 + 1] (positions of earlier subtiles all have g < 0) */
				int t0;
				if(inLds) { int a = 0, b = NTl - 1; while(a < b) { const int md = (a + b) >> 1; if(preL[md + 1] >= sufR[md + 1]) b = md; else a = md + 1; } t0 = a; }
				else { /* the tables are in global memory: every 64th entry from LDS (one compare per lane), then ONE block of 64 entries (one per lane) */
					const int nc = R >> 6, cb = off >> 6;
					int blk;                                                              /* the first true entry lies in (64 blk, 64 blk + 64] */
					{ int a = 1, b = nc; while(a < b) { const int md = (a + b) >> 1; if(cPre[cb + md] >= cSuf[cb + md]) b = md; else a = md + 1; } blk = a - 1; }   /* first coarse entry x in [1, nc) that holds (nc: none), in LDS; the predicate is monotone */
					const int i = 64 * blk + 1 + lane;                                    /* table index = subtile + 1 */
					const unsigned long long fm = __ballot(i <= NTl - 1 && preL[i] >= sufR[i]);
					t0 = fm ? 64 * blk + (__ffsll((long long) fm) - 1) : min(64 * blk + 64, NTl - 1);
					if(!fm && 64 * blk + 64 < NTl - 1) t0 = NTl - 1;                      /* (not reached: the coarse entry said the block holds it) */
				}
				unsigned long long mL = ML(t0), mR = MR(t0);
				int c0;
				{
					const int q = t0 * 64 + lane;
					const int Lq = (int) preL[t0] + __popcll(mL & rs_lane_lt(lane));
					const int Rq = (int) sufR[t0 + 1] + __popcll(mR & rs_lane_ge(lane) & ~(1ull << lane));
					const unsigned long long ok = __ballot(q >= o0 && q < qEnd && Lq - Rq >= 0);
					c0 = ok ? t0 * 64 + (__ffsll((long long) ok) - 1) : min((t0 + 1) * 64, qEnd);
				}
				int tc = c0 >> 6, m, iNext = -1;                     /* iNext: i_(m+1), the first left stopper at or after c0 (-1: none) */
				if(tc < NTl) {
					if(tc != t0) mL = ML(tc);
					m = (int) preL[tc] + __popcll(mL & rs_lane_lt(c0 & 63));
					unsigned long long cand = mL & rs_lane_ge(c0 & 63);
					int t = tc;
					while(!cand && ++t < NTl) { if(preL[t + 1] > preL[t]) cand = ML(t); }
					if(cand) iNext = t * 64 + (__ffsll((long long) cand) - 1);
				}
				else m = (int) preL[NTl];
				int jm = -1;                                         /* j_m: the m-th right stopper from the right */
				if(m >= 1) {
					int a;                                             /* last subtile t with sufR[t] >= m (sufR does not increase with t; sufR[0] >= m) */
					if(inLds) { int lo_ = 0, b = NTl - 1; while(lo_ < b) { const int md = (lo_ + b + 1) >> 1; if((int) sufR[md] >= m) lo_ = md; else b = md - 1; } a = lo_; }
					else {
						const int nc = R >> 6, cb = off >> 6;
						int blk;                                                           /* the last coarse entry still >= m: the answer lies in [64 blk, 64 blk + 63] */
						{ int a_ = 0, b = min(nc - 1, (NTl - 1) >> 6); while(a_ < b) { const int md = (a_ + b + 1) >> 1; if((int) cSuf[cb + md] >= m) a_ = md; else b = md - 1; } blk = a_; }
						const int i = 64 * blk + lane;
						const unsigned long long fm = __ballot(i <= NTl - 1 && (int) sufR[i] >= m);
						a = fm ? 64 * blk + (63 - __clzll((long long) fm)) : 64 * blk;
					}
					jm = a * 64 + rs_nth_high(MR(a), m - (int) sufR[a + 1]);
				}
				const int cutq = (m >= 1 && (iNext < 0 || iNext > jm)) ? jm : iNext;
				if(lane == 0) { shI[2] = cutq; shI[3] = m; shI[4] = jm; }
			}
			__syncthreads();
			RS_T(4);
			const int cutq = shI[2], m = shI[3], jm = shI[4];
#ifdef HU_RS_PROF
			if(tid == 0 && level < 3) { atomicAdd(&g_rs_prof[10], (unsigned long long)(level == 0 ? cutq : 0)); atomicAdd(&g_rs_prof[11], (unsigned long long)(level == 0 ? max(jm, 0) : 0)); atomicAdd(&g_rs_prof[12], (unsigned long long)(level == 0 ? m : 0)); atomicAdd(&g_rs_prof[13 + level], (unsigned long long)(hi - lo)); }
#endif
			if(cutq < 0) { failed = true; why = 4; break; }         /* no stopper where the sentinels guarantee one: not reached on consistent data */
			const int cutAbs = qBase + cutq;
			PT* RS = dst + rsOff;                                   /* right stoppers by rank, beyond the positions */
			if(tid == 0) { hLo[level] = lo; hQB[level] = qBase; hW[level] = wAbs; hCut[level] = cutAbs; hM[level] = m; hOff[level] = off; hNC[level] = R >> 6; }
			if(inLds) {
				/* ---- the m swaps in place: the left stopper of rank k <= m (all of them lie before the cut) finds j_k by a search on the scanned counts */
				for(int q = o0 + tid; q < cutq; q += HU_RS_THREADS) {
					const int t = q >> 6, bit = q & 63;
					const unsigned long long mL = mLs[t];
					if((mL >> bit) & 1ull) {
						const int k = (int) preL[t] + __popcll(mL & rs_lane_lt(bit)) + 1;
						if(k <= m) {
							int a = 0, b = NTl - 1;
							while(a < b) { const int md = (a + b + 1) >> 1; if((int) sufR[md] >= k) a = md; else b = md - 1; }
							const int j = a * 64 + rs_nth_high(mRs[a], k - (int) sufR[a + 1]);
							const PT x = lk[qBase + q - lbase + 64], y = lk[qBase + j - lbase + 64];
							lk[qBase + q - lbase + 64] = y; lk[qBase + j - lbase + 64] = x;
						}
					}
				}
				__syncthreads();
				RS_T(5);
				if(cutAbs < K) {       /* the left part ends inside the first K places: set aside as it stands, the right part goes on */
					for(int p = lo + tid; p < cutAbs; p += HU_RS_THREADS) fin[p] = rs_key(HuPair<PT>::canon(lk[p - lbase + 64]), ((uint32_t)(level + 1) << 24) | (uint32_t) p);
					if(tid == 0 && nStash < 12) { stash[nStash].lo = lo; stash[nStash].hi = cutAbs; stash[nStash].depth = depth; }
					if(nStash >= 12) { failed = true; why = 5; break; }
					++nStash;
					lo = cutAbs;
				}
				else hi = cutAbs;
				++level;
				__syncthreads();
				RS_T(6);
				continue;
			}
			/* ---- pass B1: the right stoppers j_1 .. j_m (rank from the right <= m), from the subtile of j_m on */
			auto passB1 = [&](auto l0) {
				constexpr bool L0 = decltype(l0)::value;
				const unsigned char* __restrict__ bRc = reinterpret_cast<const unsigned char*>(gMR + off);      /* this level's right-stopper bits: classified once, by the pass that wrote or counted it */
				for(int gb = (jm >> 6) * LPS; gb < gEnd; gb += HU_RS_THREADS * VU1) {
					uint32_t k[VU1][EPL], mR[VU1], sufNext[VU1];
#pragma unroll
					for(int u = 0; u < VU1; ++u) {
						const int g = min(gb + u * HU_RS_THREADS + tid, gEnd - 1);
						rs_load<L0, PT>(src, row, db.root, rowLast, qBase + g * EPL, k[u]);
						mR[u] = rs_mask_bits<EPL>(bRc, g);
						sufNext[u] = sufR[min(g / LPS, NTl - 1) + 1];      /* (global: requested with the keys) */
					}
#pragma unroll
					for(int u = 0; u < VU1; ++u) {
						const int g = gb + u * HU_RS_THREADS + tid; const bool in = g < gEnd;
						if(L0) { bool dummy = false; rs_fix_l0<PT, EPL>(k[u], qBase + g * EPL, pA, HuPair<PT>::pack(vA), pB, HuPair<PT>::pack(vB), 0u, dummy); }
						const uint32_t mRb = in ? mR[u] : 0u;
						const int c = __popc(mRb);
						int s = c;                                     /* right stoppers of this lane and the higher lanes of its subtile */
#pragma unroll
						for(int o = 1; o < LPS; o <<= 1) { const int v = __shfl_down(s, o); if((lane & (LPS - 1)) + o < LPS) s += v; }
						const int above = (int) sufNext[u] + s - c;
#pragma unroll
						for(int e = 0; e < EPL; ++e) if((mRb >> e) & 1u) { const int rk = above + __popc(mRb >> e); if(rk <= m) RS[rk - 1] = (PT) k[u][e]; }
					}
				}
			};
			if(m >= 1) { if(level0) passB1(std::true_type{}); else passB1(std::false_type{}); }
			__threadfence_block();
			__syncthreads();
			RS_T(5);
			const bool tiny = cutAbs < K;        /* the left part ends inside the first K places: the right part is needed too */
			if(!tiny) {
				/* ---- the next level is [lo, cut).  While it is still a streaming level its pivot is chosen NOW — the median of the elements that
				 * WILL stand at lo + 1, mid and cut - 1: each is the element there, or the right stopper of its rank if that one is a left stopper —
				 * and pass B2 classifies what it writes against it */
				const int nNTl = (cutq + 63) >> 6, nR = (nNTl + 1 + 63) & ~63, gEnd2 = nNTl * LPS;
				const bool fuse = cutAbs - lo > LCAP && depth > 0 && level + 1 < HU_RS_MAXLEV && tabNext + nR <= tabCap;
				if(fuse && wave == 0) {
					const int mid = lo + (cutAbs - lo) / 2;
					const int P = lane == 0 ? lo + 1 : (lane == 1 ? mid : cutAbs - 1);
					uint32_t pr = 0;
					if(lane < 3) {
						pr = E(P);
						if(!rs_ltp(pr, pivP)) { const int q = P - qBase; pr = HuPair<PT>::canon(RS[(int) preL[q >> 6] + __popcll(gML[off + (q >> 6)] & rs_lane_lt(q & 63))]); }
					}
					const uint32_t ea = __shfl(pr, 0), eb = __shfl(pr, 1), ec = __shfl(pr, 2);
					if(lane == 0) {
						const int c = rs_median3(ea, eb, ec);
						shP[2] = c == 0 ? ea : (c == 1 ? eb : ec);
						shI[7] = c == 0 ? lo + 1 : (c == 1 ? mid : cutAbs - 1);
					}
				}
				__syncthreads();
				const uint32_t nPiv = fuse ? shP[2] : 0u; const int nW = fuse ? shI[7] : -1, nOff = tabNext;
				const uint32_t loVal = fuse ? nPiv : pivP;      /* what stands at lo in the array written: this level's pivot, or the next one after its swap */
				/* ---- pass B2: the left part [lo, cut) of the next level — after the next pivot's swap when that is known */
				if(tid == 0) dst[lo] = HuPair<PT>::pack(loVal);
				auto passB2 = [&](auto l0) {
					constexpr bool L0 = decltype(l0)::value;
					unsigned char* bL = reinterpret_cast<unsigned char*>(gML + nOff); unsigned char* bR = reinterpret_cast<unsigned char*>(gMR + nOff);
					const unsigned char* __restrict__ bLc = reinterpret_cast<const unsigned char*>(gML + off);      /* this level's left-stopper bits */
					const uint32_t loRaw = HuPair<PT>::pack(loVal), pivRaw = HuPair<PT>::pack(pivP);
					for(int gb = 0; gb < gEnd2; gb += HU_RS_THREADS * VU2) {
						uint32_t k[VU2][EPL], mLt[VU2], preHere[VU2];
#pragma unroll
						for(int u = 0; u < VU2; ++u) {
							const int g = min(gb + u * HU_RS_THREADS + tid, gEnd2 - 1);
							rs_load<L0, PT>(src, row, db.root, rowLast, qBase + g * EPL, k[u]);
							mLt[u] = rs_mask_bits<EPL>(bLc, g);
							preHere[u] = preL[min(g / LPS, NTl)];
						}
#pragma unroll
						for(int u = 0; u < VU2; ++u) {
							const int g = gb + u * HU_RS_THREADS + tid, p0 = qBase + g * EPL; const bool in = g < gEnd2;
							const uint32_t vb = in ? rs_valid<EPL>(g * EPL, o0, cutq) : 0u;
							if(L0) { bool dummy = false; rs_fix_l0<PT, EPL>(k[u], p0, pA, HuPair<PT>::pack(vA), pB, HuPair<PT>::pack(vB), 0u, dummy); }
							const uint32_t mLb = mLt[u] & vb;
							/* the left stoppers of a group have consecutive ranks: one (unaligned) vector of right stoppers, dealt out in order */
							const int c = __popc(mLb);
							int s = c;
#pragma unroll
							for(int o = 1; o < LPS; o <<= 1) { const int v = __shfl_up(s, o); if((lane & (LPS - 1)) >= o) s += v; }
							const int r0 = min((int) preHere[u] + s - c, max(m - 1, 0));
							uint32_t rx, ry, rz, rw;
							if(sizeof(PT) == 2) { const HuRsU4h v = *reinterpret_cast<const HuRsU4h*>(RS + r0); rx = v.x; ry = v.y; rz = v.z; rw = v.w; }
							else { const HuRsU4w v = *reinterpret_cast<const HuRsU4w*>(RS + r0); rx = v.x; ry = v.y; rz = v.z; rw = v.w; }
#pragma unroll
							for(int e = 0; e < EPL; ++e) {
								const bool bit = (mLb >> e) & 1u;
								if(sizeof(PT) == 2) { /* the window of right stoppers moves on by one key (16 bits) where one was dealt: four funnel shifts by 0 or 16 */
									k[u][e] = bit ? (rx & 0xffffu) : k[u][e];
									const uint32_t sh = bit ? 16u : 0u;
									rx = __builtin_amdgcn_alignbit(ry, rx, sh); ry = __builtin_amdgcn_alignbit(rz, ry, sh); rz = __builtin_amdgcn_alignbit(rw, rz, sh); rw >>= sh;
								}
								else {
									k[u][e] = bit ? rx : k[u][e];
									rx = bit ? ry : rx; ry = bit ? rz : ry; rz = bit ? rw : rz;
								}
							}
							if((unsigned)(lo - p0) < (unsigned) EPL || (unsigned)(nW - p0) < (unsigned) EPL) {
#pragma unroll
								for(int e = 0; e < EPL; ++e) { k[u][e] = p0 + e == lo ? loRaw : k[u][e]; k[u][e] = p0 + e == nW ? pivRaw : k[u][e]; }   /* the next pivot's swap: this level's pivot goes where that one stood */
							}
							if(in) {
								uint4 v;
								if(sizeof(PT) == 2) {
									v.x = k[u][0] | (k[u][1] << 16); v.y = k[u][2] | (k[u][3] << 16);
									v.z = k[u][4 % EPL] | (k[u][5 % EPL] << 16); v.w = k[u][6 % EPL] | (k[u][7 % EPL] << 16);
								}
								else { v.x = k[u][0]; v.y = k[u][1]; v.z = k[u][2]; v.w = k[u][3]; }
								*reinterpret_cast<uint4*>(dst + p0) = v;
							}
							if(fuse) {
								uint32_t nLb, nRb;
								rs_classify<PT, EPL>(k[u], nPiv, vb, nLb, nRb);
								rs_emit<EPL>(nLb, nRb, g, in, lane, bL, bR, cnt16);
							}
						}
					}
				};
				if(level0) passB2(std::true_type{}); else passB2(std::false_type{});
				hi = cutAbs;
				if(fuse) { counted = true; pivP = nPiv; wAbs = nW; off = nOff; tabNext = nOff + nR; }
			}
			else {
				/* the tiny left part goes to the finisher's array as it is after this partition (positions of the array of level + 1); its left
				 * stoppers are kept for the right part */
				if(tid < 64) {
					const int i = tid; const bool valid = lo + 1 + i < cutAbs;
					const uint32_t e = valid ? E(lo + 1 + i) : 0;
					const bool ls = valid && !rs_ltp(e, pivP);
					const unsigned long long mL = __ballot(ls);
					const int k = __popcll(mL & rs_lane_lt(lane));
					if(ls) lsb[k] = e;
					const uint32_t tag = ((uint32_t)(level + 1) << 24);
					if(valid) fin[lo + 1 + i] = rs_key(ls ? HuPair<PT>::canon(RS[k]) : e, tag | (uint32_t)(lo + 1 + i));
					if(tid == 0) { fin[lo] = rs_key(pivP, tag | (uint32_t) lo); if(nStash < 12) { stash[nStash].lo = lo; stash[nStash].hi = cutAbs; stash[nStash].depth = depth; } }
				}
				__syncthreads();
				if(nStash >= 12) { failed = true; why = 5; break; }
				++nStash;
				/* the right part [cut, hi): a right stopper of rank k <= m receives the k-th left stopper (the element that sat at i_k) */
				for(int base = cutq / HU_RS_THREADS * HU_RS_THREADS; base < qEnd; base += HU_RS_THREADS) {
					const int q = base + tid, t = q >> 6;
					const bool valid = q >= o0 && q < qEnd;
					const uint32_t e = valid ? E(qBase + q) : 0;
					const bool rs = valid && !rs_ltp(pivP, e);
					const unsigned long long mR = __ballot(rs);
					if(valid && q >= cutq) {
						const int rk = rs ? (int) sufR[t + 1] + __popcll(mR & rs_lane_ge(lane)) : 0;
						dst[qBase + q] = HuPair<PT>::pack((rs && rk <= m) ? lsb[rk - 1] : e);
					}
				}
				lo = cutAbs;
			}
			__threadfence_block();
			__syncthreads();
			RS_T(6);
			src = dst; dst = (dst == bufA) ? bufB : bufA;
			level0 = false;
			pA = pB = -1;
			++level;
		}
#ifdef HU_RS_PROF
		rs_lv = 9;
#endif
		if(__syncthreads_or((failed || nan) ? 1 : 0)) { /* the host path finishes this read */
			if(tid == 0) { seedCnt[read] = 0; const int at = atomicAdd(bail, 1); bail[2 + at] = read | ((why ? why : 3) << 26); }
			continue;
		}
		/* ---- the last range into LDS beside the set-aside prefix; finished with the literal sequential algorithm */
		for(int p = lo + tid; p < hi; p += HU_RS_THREADS) { const uint32_t e = E(p); if(!nan) fin[p] = rs_key(e, ((uint32_t) level << 24) | (uint32_t) p); }
		if(__syncthreads_or(nan ? 1 : 0)) { if(tid == 0) { seedCnt[read] = 0; const int at = atomicAdd(bail, 1); bail[2 + at] = read | (3 << 26); } continue; }
		RS_T(7);
		/* the set-aside parts and the last range are disjoint stretches of fin: one THREAD each (lane 0 of a wave: the sequential loops of up to eight
		 * stretches run side by side on different SIMDs instead of one after the other), each with a stack of its own */
		if(tid == 0) shI[5] = 1;
		__syncthreads();
		if(lane == 0) {
			for(int s = wave; s <= nStash; s += HU_RS_THREADS / 64) {
				const bool ok = s < nStash ? rs_seq_loop(fin, seqStack + 72 * wave, stash[s].lo, stash[s].hi, stash[s].depth, K) : rs_seq_loop(fin, seqStack + 72 * wave, lo, hi, depth, K);
				if(!ok) shI[5] = 0;
			}
		}
		__syncthreads();
		if(shI[5]) { /* __final_insertion_sort over the blocks that hold the first K places: an insertion sort is a STABLE sort, and a stable sort of e elements is a
		              * rank per element — keys smaller, or equal and earlier — counted by e threads at once */
			const int e = min(hi, K + 16);
			uint64_t v = 0; int rank = 0;
			if(tid < e) {
				v = fin[tid]; const uint64_t kv = v >> HU_RS_IDBITS;
				for(int j = 0; j < e; ++j) { const uint64_t kj = fin[j] >> HU_RS_IDBITS; rank += (kj < kv) || (kj == kv && j < tid); }
			}
			__syncthreads();
			if(tid < e) fin[rank] = v;
		}
		__syncthreads();
		RS_T(8);
		if(!shI[5]) { if(tid == 0) { seedCnt[read] = 0; const int at = atomicAdd(bail, 1); bail[2 + at] = read | (6 << 26); } continue; }
		const int keep = min(K, hi);
		if(tid == 0) seedCnt[read] = keep;
		if(tid < keep) {
			/* ---- trace-back: from a position of the array of level `lv` to the position in node order.  Per level: undo the partition (a place
			 * left of the cut that held a left stopper received the right stopper of the same rank, and the other way round right of it), then
			 * undo the pivot's swap */
			const uint32_t tag = (uint32_t)(fin[tid] & ((1ull << HU_RS_IDBITS) - 1));
			int P = (int)(tag & 0xffffffu);
			for(int l = (int)(tag >> 24) - 1; l >= 0; --l) {
				const int llo = hLo[l], qb = hQB[l], o = hOff[l], mm = hM[l];
				if(P != llo) {
					const int q = P - qb, t = q >> 6, bit = q & 63;
					if(P < hCut[l]) {
						const unsigned long long mL = gML[o + t];
						if((mL >> bit) & 1ull) {
							const int k = (int) gPre[o + t] + __popcll(mL & rs_lane_lt(bit)) + 1;
							if(k <= mm) {   /* j_k: in the last subtile a with sufR[a] >= k */
								int c = 0;
								for(int x = 1; x < hNC[l]; ++x) c = (int) cSuf[(o >> 6) + x] >= k ? x : c;
								const uint4* blk = reinterpret_cast<const uint4*>(gSuf + o + c * 64);
								int n = 0;
#pragma unroll 4
								for(int x = 0; x < 16; ++x) { const uint4 v = blk[x]; n += ((int) v.x >= k) + ((int) v.y >= k) + ((int) v.z >= k) + ((int) v.w >= k); }
								const int a = c * 64 + n - 1;
								P = qb + a * 64 + rs_nth_high(gMR[o + a], k - (int) gSuf[o + a + 1]);
							}
						}
					}
					else {
						const unsigned long long mR = gMR[o + t];
						if((mR >> bit) & 1ull) {
							const int k = (int) gSuf[o + t + 1] + __popcll(mR & rs_lane_ge(bit));
							if(k <= mm) {   /* i_k: in the last subtile a with preL[a] < k */
								int c = 0;
								for(int x = 1; x < hNC[l]; ++x) c = cPre[(o >> 6) + x] < (uint32_t) k ? x : c;
								const uint4* blk = reinterpret_cast<const uint4*>(gPre + o + c * 64);
								int n = 0;
#pragma unroll 4
								for(int x = 0; x < 16; ++x) { const uint4 v = blk[x]; n += (v.x < (uint32_t) k) + (v.y < (uint32_t) k) + (v.z < (uint32_t) k) + (v.w < (uint32_t) k); }
								const int a = c * 64 + n - 1;
								P = qb + a * 64 + rs_nth_low(gML[o + a], k - (int) gPre[o + a]);
							}
						}
					}
				}
				P = P == llo ? hW[l] : (P == hW[l] ? llo : P);
			}
			const int node = P < db.root ? P : P + 1;
			seedId[(size_t) read * HU_MAX_SEEDS + tid] = node;
			seedDN[(size_t) read * HU_MAX_SEEDS + tid] = HuPair<PT>::canon(row[node]);
			parDN[(size_t) read * HU_MAX_SEEDS + tid] = HuPair<PT>::canon(row[db.parent[node]]);
		}
		RS_T(9);
	}
}
