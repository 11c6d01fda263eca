// The <= 64-record stages of the per-read task on the device: filterPlacements (src/HmmUFOtu_main.cpp:162-173), the assembly of
// PTPlacement after placeSeq (src/PhyloTreeUnrooted.cpp:936-952), calcQValues (src/HmmUFOtu_main.cpp:182-216) and the final sort +
// bestPlace (src/hmmufotu.cpp:725-733).  Rounds 1-2 ran them on the host with literal std::sort calls, because all final keys tie (SURVEY.md
// F4) and the pick IS the tie permutation of libstdc++'s introsort; with that algorithm restated for the seed stage (hu_kern_refsort.h,
// pinned against the literal std::sort by tests/test_seed_order.py) the same restatement, here complete with the heap-sort branch, runs
// one thread per read on <= 64 records and the stages need no copy of seeds / estimates to the host and no candidate list back.
#pragma once
#include "hu_common.h"
#include "hu_kern_sep.h"

/* a candidate's PTPlacement (src/PhyloTreeUnrooted.h:410-510) with node ids; shared by host and device */
struct HuPlaceRec {
	int32_t seedIdx, cNode, pNode, aNode, iters, pad_;
	double wuv, ratio, wnr, loglik, height, qPlace, qTaxon, estLoglik, rootLoglik;
	__host__ __device__ double annoDist() const { return aNode == cNode ? wuv * ratio + wnr : (1 - ratio) * wuv + wnr; }
};

/* ---- std::sort (libstdc++ <bits/stl_algo.h>, restated from its published algorithm; see hu_host.cpp) on n <= 64 doubles compared with <,
 * each carrying a byte: introsort loop down to 16 elements (median of first + 1, mid, last - 1 to first; unguarded Hoare partition; depth
 * limit 2 floor(lg n), then heap sort: __make_heap + __sort_heap over __adjust_heap / __push_heap), one final insertion sort. */
struct HuSort64 {
	double* v; uint8_t* ix;
	__device__ inline void swp(int a, int b) { const double t = v[a]; v[a] = v[b]; v[b] = t; const uint8_t u = ix[a]; ix[a] = ix[b]; ix[b] = u; }
	__device__ inline void push_heap(int first, int hole, int top, double val, uint8_t vi) {
		int parent = (hole - 1) / 2;
		while(hole > top && v[first + parent] < val) { v[first + hole] = v[first + parent]; ix[first + hole] = ix[first + parent]; hole = parent; parent = (hole - 1) / 2; }
		v[first + hole] = val; ix[first + hole] = vi;
	}
	__device__ inline void adjust_heap(int first, int hole, int len, double val, uint8_t vi) {
		const int top = hole;
		int child = hole;
		while(child < (len - 1) / 2) {
			child = 2 * (child + 1);
			if(v[first + child] < v[first + child - 1]) child--;
			v[first + hole] = v[first + child]; ix[first + hole] = ix[first + child];
			hole = child;
		}
		if((len & 1) == 0 && child == (len - 2) / 2) {
			child = 2 * (child + 1);
			v[first + hole] = v[first + child - 1]; ix[first + hole] = ix[first + child - 1];
			hole = child - 1;
		}
		push_heap(first, hole, top, val, vi);
	}
	__device__ inline void heap_sort(int first, int last) { /* __partial_sort(first, last, last) */
		const int len = last - first;
		if(len >= 2) for(int parent = (len - 2) / 2; ; --parent) { adjust_heap(first, parent, len, v[first + parent], ix[first + parent]); if(parent == 0) break; }
		for(int l = last; l - first > 1; ) { /* __sort_heap: __pop_heap(first, l - 1, l - 1) */
			--l;
			const double val = v[l]; const uint8_t vi = ix[l];
			v[l] = v[first]; ix[l] = ix[first];
			adjust_heap(first, 0, l - first, val, vi);
		}
	}
	__device__ inline void loop(int first, int last, int depth) {
		/* the recursion into [cut, last) as an explicit stack: at most lg 64 + a few entries are ever pending */
		int sf[16], sl[16], sd[16], sp = 0;
		sf[0] = first; sl[0] = last; sd[0] = depth; sp = 1;
		while(sp > 0) {
			--sp;
			int f = sf[sp], l = sl[sp], dp = sd[sp];
			while(l - f > 16) {
				if(dp == 0) { heap_sort(f, l); break; }
				--dp;
				const int mid = f + (l - f) / 2, a = f + 1, c = l - 1;
				if(v[a] < v[mid]) { if(v[mid] < v[c]) swp(f, mid); else if(v[a] < v[c]) swp(f, c); else swp(f, a); }
				else if(v[a] < v[c]) swp(f, a);
				else if(v[mid] < v[c]) swp(f, c);
				else swp(f, mid);
				const double pv = v[f];
				int i = f + 1, j = l;
				for(;;) {
					while(i < l && v[i] < pv) ++i;            /* (the bounds only matter for NaN keys, on which std::sort is undefined) */
					--j;
					while(j > f && pv < v[j]) --j;
					if(!(i < j)) break;
					swp(i, j);
					++i;
				}
				if(sp < 16) { sf[sp] = i; sl[sp] = l; sd[sp] = dp; ++sp; }
				l = i;
			}
		}
	}
	__device__ inline void sort(int n) {
		if(n < 2) return;
		int lg = 0; for(int m = n; m > 1; m >>= 1) ++lg;
		loop(0, n, 2 * lg);
		for(int i = 1; i < n; ++i) { /* __final_insertion_sort */
			const double val = v[i]; const uint8_t vi = ix[i]; int j = i;
			while(j > 0 && val < v[j - 1]) { v[j] = v[j - 1]; ix[j] = ix[j - 1]; --j; }
			v[j] = val; ix[j] = vi;
		}
	}
};
/* order[p] = index of the element that std::sort(rbegin, rend, less-by-key) leaves at place p (descending; ties as libstdc++ leaves them) */
__device__ inline void hu_sort_desc64(const double* key, int n, uint8_t* order) {
	double v[64]; uint8_t ix[64];
	for(int i = 0; i < n; ++i) { v[i] = key[n - 1 - i]; ix[i] = (uint8_t)(n - 1 - i); }       /* the reverse iterators' view of the sequence */
	HuSort64 s{v, ix};
	s.sort(n);
	for(int p = 0; p < n; ++p) order[p] = ix[n - 1 - p];
}

__global__ __launch_bounds__(64) void k_sort_desc_test(int rows, int n, const double* __restrict__ keys, int32_t* __restrict__ order) {
	const int r = blockIdx.x * 64 + threadIdx.x;
	if(r >= rows) return;
	double k[64]; uint8_t o[64];
	for(int i = 0; i < n; ++i) k[i] = keys[(size_t) r * n + i];
	hu_sort_desc64(k, n, o);
	for(int i = 0; i < n; ++i) order[(size_t) r * n + i] = o[i];
}

/* filterPlacements: per read the seed slots in descending order of the estimated loglik, cut at maxError below the best.
 * One WAVE per read.  While a read's keys are pairwise different (and none is a NaN) the sorted order is a fact of the keys, not of the sort:
 * the place of a seed is the number of greater keys — sixty-four broadcasts and compares — and the kept seeds are those within maxError of the
 * best (what the reference's loop over the sorted list keeps: the gap to the best only grows along it).  A read with two equal keys (the two
 * writings of an attachment at a node, or -inf) or a NaN gets libstdc++'s own order from one lane running the restated std::sort.
 * (One thread per read sorting in private memory took 0.55 ms per 8,192 reads.) */
__device__ __attribute__((noinline)) void hu_filter_seq(int r, int cnt, const HuEstOut* __restrict__ est, double maxError, int32_t* __restrict__ candCnt, uint8_t* __restrict__ filtSlot) {
	double k[64]; uint8_t o[64];
	for(int s = 0; s < cnt; ++s) k[s] = est[(size_t) r * HU_MAX_SEEDS + s].loglik;
	hu_sort_desc64(k, cnt, o);
	int g = 0;
	if(cnt > 0) {
		const double best = k[o[0]];
		for(; g < cnt; ++g) { if(best - k[o[g]] > maxError) break; filtSlot[(size_t) r * HU_MAX_SEEDS + g] = o[g]; }
	}
	candCnt[r] = g;
}
__global__ __launch_bounds__(256) void k_filter(int n, const int32_t* __restrict__ seedCnt, const HuEstOut* __restrict__ est, double maxError,
		int32_t* __restrict__ candCnt, uint8_t* __restrict__ filtSlot, int forceSeq) {
	const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if(r >= n) return;
	const int cnt = min(seedCnt[r], (int) HU_MAX_SEEDS);
	if(cnt <= 0) { if(lane == 0) candCnt[r] = 0; return; }
	const double key = lane < cnt ? est[(size_t) r * HU_MAX_SEEDS + lane].loglik : 0.0;
	int gt = 0, eq = 0;
	for(int j = 0; j < cnt; ++j) {
		const double kj = __shfl(key, j);
		gt += kj > key ? 1 : 0; eq += kj == key ? 1 : 0;
	}
	const bool odd = lane < cnt && eq != 1;          /* a tie (eq > 1) or a NaN (eq == 0) */
	if(__ballot(odd) || forceSeq) { if(lane == 0) hu_filter_seq(r, cnt, est, maxError, candCnt, filtSlot); return; }
	const unsigned long long top = __ballot(lane < cnt && gt == 0);
	const double best = __shfl(key, __ffsll((long long) top) - 1);
	const bool keep = lane < cnt && !(best - key > maxError);
	const unsigned long long kept = __ballot(keep);
	if(keep) filtSlot[(size_t) r * HU_MAX_SEEDS + gt] = (uint8_t) lane;
	if(lane == 0) candCnt[r] = __popcll(kept);
}

/* candidate offsets (exclusive scan of the counts, total at [n]) and what the placement launch needs to know of the batch: the number of
 * candidates and the largest gap / base site counts of a read with a region of at most rMain columns (0: any; k_site_count) — one workgroup */
__global__ __launch_bounds__(1024) void k_cand_scan(int n, const int32_t* __restrict__ candCnt, int32_t* __restrict__ candOff, const int32_t* __restrict__ permCnt,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, int32_t* __restrict__ meta, int rMain) {
	__shared__ int part[1024], mg[1024], mb[1024];
	const int tid = threadIdx.x, per = (n + 1023) / 1024, a0 = tid * per, a1 = min(n, a0 + per);
	int s = 0, g = 0, bs = 0;
	for(int r = a0; r < a1; ++r) { s += candCnt[r]; if(permCnt && rend[r] >= rstart[r] && (!rMain || rend[r] - rstart[r] + 1 <= rMain)) { g = max(g, permCnt[2 * r]); bs = max(bs, permCnt[2 * r + 1]); } }   /* the maxima of the main launch's reads */
	part[tid] = s; mg[tid] = g; mb[tid] = bs;
	__syncthreads();
	if(tid == 0) { int acc = 0, G = 0, B = 0; for(int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = acc; acc += v; G = max(G, mg[i]); B = max(B, mb[i]); } candOff[n] = acc; meta[0] = acc; meta[1] = G; meta[2] = B; }
	__syncthreads();
	int acc = part[tid];
	for(int r = a0; r < a1; ++r) { candOff[r] = acc; acc += candCnt[r]; }
}

/* the candidates of every read in filter order: what the placement kernels read (HuCand) and the start of each PTPlacement */
__global__ void k_build_cands(HuDbDev db, int n, const int32_t* __restrict__ candCnt, const int32_t* __restrict__ candOff, const uint8_t* __restrict__ filtSlot,
		const int32_t* __restrict__ seedId, const HuEstOut* __restrict__ est, HuCand* __restrict__ cands, HuPlaceRec* __restrict__ places) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if(i >= n * HU_MAX_SEEDS) return;
	const int r = i / HU_MAX_SEEDS, k = i % HU_MAX_SEEDS;
	if(k >= candCnt[r]) return;
	const int slot = filtSlot[i], at = candOff[r] + k;
	const int node = seedId[(size_t) r * HU_MAX_SEEDS + slot];
	const HuEstOut e = est[(size_t) r * HU_MAX_SEEDS + slot];
	HuCand c; c.read = r; c.node = node; c.ratio0 = e.ratio; c.wnr0 = e.wnr;
	cands[at] = c;
	HuPlaceRec p;
	p.seedIdx = slot; p.cNode = node; p.pNode = db.parent[node]; p.aNode = e.ratio <= 0.5 ? node : p.pNode; p.iters = 0; p.pad_ = 0;
	p.wuv = db.blen[node]; p.ratio = e.ratio; p.wnr = e.wnr; p.loglik = e.loglik; p.height = 0; p.qPlace = p.qTaxon = 0; p.estLoglik = e.loglik; p.rootLoglik = NAN;
	places[at] = p;
}

/* what std::sort(rbegin, rend) leaves in the first place when all n keys are equal: g_alleq_first[n], filled once per process by k_alleq_init */
__device__ uint8_t g_alleq_first[HU_MAX_SEEDS + 1];
__global__ __launch_bounds__(128) void k_alleq_init() {
	const int n = threadIdx.x;
	if(n > (int) HU_MAX_SEEDS) return;
	double k[64]; uint8_t o[64];
	for(int i = 0; i < n; ++i) k[i] = 0.0;
	o[0] = 0;
	hu_sort_desc64(k, n, o);
	g_alleq_first[n] = n ? o[0] : 0;
}

__device__ inline double hu_add_scaled(double a, double c) { const double s = fmax(a, c); return log(exp(a - s) + exp(c - s)) + s; }   /* src/math/Stats.h:233-239 */
__device__ inline double hu_p2q(double p) { return -10 * log(p) / log(10.0); }                                                       /* :240-241 */

/* PTPlacement after placeSeq, calcQValues, the final sort and bestPlace — one thread per read.
 * llTab[k] = the F4 constant of a region of k columns (k sequential additions of log(sum_i pi_i e), as treeLoglik adds them) */
__global__ __launch_bounds__(64) void k_finish(HuDbDev db, int n, const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend,
		const int32_t* __restrict__ candOff, HuPlaceRec* __restrict__ places, const HuPlaceOut* __restrict__ placeOut, const double* __restrict__ rootLL,
		const double* __restrict__ llTab, const int32_t* __restrict__ annoId, double maxHeight, int onlyML, int prior, int given, hu_place_rec* __restrict__ best, int forceSort) {
	const int r = blockIdx.x * 64 + threadIdx.x;
	if(r >= n) return;
	hu_place_rec br;
	br.c_node = br.p_node = br.a_node = -1; br.n_cand = 0;
	br.wuv = br.ratio = br.wnr = br.loglik = br.height = br.q_place = br.q_taxon = br.anno_dist = br.est_loglik = br.root_loglik = NAN;
	const int lo = candOff[r], hi = candOff[r + 1], cnt = min(hi - lo, (int) HU_MAX_SEEDS);
	if(cnt <= 0) { best[r] = br; return; }
	const int nsite = rend[r] - rstart[r] + 1;
	const double ll = llTab[nsite > 0 ? nsite : 0];
	double key[64]; uint8_t ord[64];
	for(int c = 0; c < cnt; ++c) { /* PTUnrooted::placeSeq const (src/PhyloTreeUnrooted.cpp:936-952) */
		HuPlaceRec p = places[lo + c];
		if(!given) {
			const HuPlaceOut po = placeOut[lo + c];
			p.rootLoglik = rootLL ? rootLL[lo + c] : NAN;
			p.loglik = rootLL ? p.rootLoglik : ll; p.wnr = po.wnr; p.ratio = po.wur / p.wuv; p.height = db.height[p.cNode] + po.wur; p.iters = po.iters | (po.pad << 8);
			p.aNode = (p.ratio <= 0.5 || db.height[p.pNode] > maxHeight) ? p.cNode : p.pNode;
		}
		p.qPlace = p.qTaxon = NAN;      /* --ML computes none */
		places[lo + c] = p;
		key[c] = p.loglik;
	}
	if(!onlyML) { /* calcQValues (src/HmmUFOtu_main.cpp:182-216) */
		int taxKey[64]; double taxVal[64]; int nTax = 0;
		double pp[64];
		double norm = -INFINITY;
		for(int c = 0; c < cnt; ++c) {
			const HuPlaceRec p = places[lo + c];
			const double logPrior = prior == HU_PRIOR_UNIFORM ? -0.0 : -(p.annoDist() - p.wnr + p.height);
			const double v = p.loglik + logPrior;
			pp[c] = v;
			const int tk = annoId[p.aNode];
			int f = -1;
			for(int t = 0; t < nTax; ++t) if(taxKey[t] == tk) { f = t; break; }
			if(f >= 0) taxVal[f] = hu_add_scaled(taxVal[f], v); else { taxKey[nTax] = tk; taxVal[nTax] = v; ++nTax; }
			norm = hu_add_scaled(norm, v);
		}
		double mx = pp[0];
		for(int c = 0; c < cnt; ++c) mx = fmax(mx, pp[c]);
		double sum = 0;
		for(int c = 0; c < cnt; ++c) { pp[c] = exp(pp[c] - mx); sum += pp[c]; }
		for(int c = 0; c < cnt; ++c) {
			HuPlaceRec p = places[lo + c];
			double q = hu_p2q(1 - pp[c] / sum);
			p.qPlace = q > 250 ? 250 : q;
			const int tk = annoId[p.aNode];
			double tp = 0;
			for(int t = 0; t < nTax; ++t) if(taxKey[t] == tk) tp = taxVal[t];
			q = hu_p2q(1 - exp(tp - norm));
			p.qTaxon = q > 250 ? 250 : q;
			places[lo + c] = p;
			key[c] = p.qPlace;
		}
	}
	/* std::sort(places.rbegin(), places.rend(), compareByQPlace | compareByLoglik) (src/hmmufotu.cpp:726, 730), of which bestPlace takes the first
	 * element.  A unique maximum is first whatever the sort does with the rest; when ALL keys tie (the rule: SURVEY.md F4 gives every candidate
	 * the same loglik) the first place is a function of the count alone (g_alleq_first, from the same restated std::sort); only a maximum shared
	 * by some candidates needs the sort itself */
	int first;
	{
		double mx = key[0]; int am = 0, nmx = 1; bool nanKey = key[0] != key[0];
		for(int c = 1; c < cnt; ++c) { nanKey |= key[c] != key[c]; if(key[c] > mx) { mx = key[c]; am = c; nmx = 1; } else if(key[c] == mx) ++nmx; }
		if(!nanKey && nmx == 1 && !forceSort) first = am;
		else if(!nanKey && nmx == cnt && !forceSort) first = g_alleq_first[cnt];
		else { hu_sort_desc64(key, cnt, ord); first = ord[0]; }
	}
	const HuPlaceRec p = places[lo + first];
	br.c_node = p.cNode; br.p_node = p.pNode; br.a_node = p.aNode; br.n_cand = hi - lo;
	br.wuv = p.wuv; br.ratio = p.ratio; br.wnr = p.wnr; br.loglik = p.loglik; br.height = p.height;
	br.q_place = p.qPlace; br.q_taxon = p.qTaxon; br.anno_dist = p.annoDist(); br.est_loglik = p.estLoglik; br.root_loglik = p.rootLoglik;
	best[r] = br;
}

/* ------------------------------------------------------------------------------------------------
 * Launch orders (the scan's tiles by region start, the estimate workgroups by seed node, the placement workgroups by candidate node): the order of
 * (key, value) pairs by key, ties by value, for keys in [0, bound] (a larger key counts as bound: invalid slots, last).  The keys are node ids or
 * CS columns — a few hundred thousand distinct values at most — so this is a COUNTING sort on the key itself: one histogram, one scan of the
 * counters, one scatter, and a pass that puts every key's short run of values in ascending order (which makes the order independent of the
 * scatter's atomics).  Rounds 1-3 called hipcub::DeviceRadixSort here, four passes of a general 32-bit radix sort for a key space this small.
 * Only speed depends on these orders (which workgroups share an L2), never a result. */
/* (the overflow bucket — the empty (read, seed) slots, a fifth of all — is counted and filled once per WAVE: 10^5 atomics on one address serialise) */
__global__ __launch_bounds__(256) void k_cs_hist(int n, const uint32_t* __restrict__ key, uint32_t bound, uint32_t* __restrict__ cnt) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	const uint32_t k = i < n ? key[i] : 0u;
	const unsigned long long over = __ballot(i < n && k >= bound);
	if(i < n && k < bound) atomicAdd(&cnt[k], 1u);
	if(over && (threadIdx.x & 63) == 0) atomicAdd(&cnt[bound], (uint32_t) __popcll(over));
}
/* start[b] = elements with a key below b (b = 0 .. m), cursor = start: tiles of 1,024 counters — (1) the tiles' sums, (2) their scan by one workgroup,
 * (3) every tile scanned behind its offset.  All loads coalesced; a one-workgroup scan of 2 x 10^5 counters walked them at one memory round trip per
 * element and thread: 3 ms. */
__device__ inline uint32_t cs_block_scan(uint32_t v, uint32_t* wsum /* LDS [16] */, uint32_t& total) { /* inclusive scan over 1,024 threads */
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for(int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(v, o); if(lane >= o) v += t; }
	if(lane == 63) wsum[wave] = v;
	__syncthreads();
	uint32_t before = 0, tot = 0;
#pragma unroll
	for(int w = 0; w < 16; ++w) { const uint32_t t = wsum[w]; if(w < wave) before += t; tot += t; }
	total = tot;
	__syncthreads();
	return v + before;
}
__global__ __launch_bounds__(1024) void k_cs_tile_sums(int m, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ tileSum) {
	__shared__ uint32_t wsum[16];
	const int i = blockIdx.x * 1024 + threadIdx.x;
	uint32_t tot;
	(void) cs_block_scan(i < m ? cnt[i] : 0u, wsum, tot);
	if(threadIdx.x == 0) tileSum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_cs_scan_tiles(int nTiles, uint32_t* __restrict__ tileSum) { /* exclusive, in place; nTiles <= 2^24 / 1024 */
	__shared__ uint32_t wsum[16];
	uint32_t carry = 0;
	for(int base = 0; base < nTiles; base += 1024) {
		const int i = base + threadIdx.x;
		const uint32_t v = i < nTiles ? tileSum[i] : 0u;
		uint32_t tot;
		const uint32_t inc = cs_block_scan(v, wsum, tot);
		if(i < nTiles) tileSum[i] = carry + inc - v;
		carry += tot;
	}
}
__global__ __launch_bounds__(1024) void k_cs_scan(int m, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ tileOff, uint32_t* __restrict__ start, uint32_t* __restrict__ cursor) {
	__shared__ uint32_t wsum[16];
	const int i = blockIdx.x * 1024 + threadIdx.x;
	const uint32_t v = i < m ? cnt[i] : 0u;
	uint32_t tot;
	const uint32_t inc = cs_block_scan(v, wsum, tot);
	const uint32_t ex = tileOff[blockIdx.x] + inc - v;
	if(i < m) { start[i] = ex; cursor[i] = ex; }
	if(i == m - 1) start[m] = ex + v;
}
__global__ __launch_bounds__(256) void k_cs_scatter(int n, const uint32_t* __restrict__ key, const uint32_t* __restrict__ val, uint32_t bound, uint32_t* __restrict__ cursor, uint32_t* __restrict__ out) {
	const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
	const uint32_t k = i < n ? key[i] : 0u;
	const bool ov = i < n && k >= bound;
	const unsigned long long over = __ballot(ov);
	if(i < n && k < bound) out[atomicAdd(&cursor[k], 1u)] = val[i];
	if(over) {
		uint32_t base = 0;
		if(lane == 0) base = atomicAdd(&cursor[bound], (uint32_t) __popcll(over));
		base = __shfl(base, 0);
		if(ov) out[base + (uint32_t) __popcll(over & ((1ull << lane) - 1ull))] = val[i];
	}
}
/* the values of one key in ascending order: one wave per key, runs of up to 1,024 values staged in LDS and placed by rank (the values of a run are
 * distinct: slot or read numbers).  A key shared by more values than that keeps the scatter's order — harmless, see above. */
__global__ __launch_bounds__(256) void k_cs_runs(int m, const uint32_t* __restrict__ start, uint32_t* __restrict__ out) {
	__shared__ uint32_t stage[4][1024];
	const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if(b >= m) return;
	const uint32_t s = start[b], len = start[b + 1] - s;
	if(len < 2 || len > 1024) return;
	uint32_t* st = stage[threadIdx.x >> 6];
	for(uint32_t i = lane; i < len; i += 64) st[i] = out[s + i];
	__builtin_amdgcn_wave_barrier();
	__threadfence_block();
	for(uint32_t i = lane; i < len; i += 64) {
		const uint32_t v = st[i];
		uint32_t r = 0;
		for(uint32_t j = 0; j < len; ++j) r += st[j] < v;
		out[s + r] = v;
	}
}
