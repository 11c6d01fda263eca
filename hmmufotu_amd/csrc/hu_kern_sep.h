// HIP kernels of the Seed-Estimate-Place stage (gfx950, wave64).  Included by hu_engine.hip.
//
//   k_seed_pdist  getSeed's node scan: SeqUtils::pDist of the read against every node sequence
//                 (src/HmmUFOtu_main.cpp:131-137, src/SeqUtils.cpp:37-54)
//   k_seed_topk   std::sort + truncation to max_nseed (src/HmmUFOtu_main.cpp:139,
//                 src/hmmufotu.cpp:646-647) as an exact (dist, node id) selection
//   k_estimate    PTUnrooted::estimateSeq (src/PhyloTreeUnrooted.cpp:849-877)
//   k_place       PTUnrooted::placeSeq + joint optimizeBranchLength
//                 (src/PhyloTreeUnrooted.cpp:879-954, 800-847, 749-798)
#pragma once
#include "hu_common.h"

#define HU_TOPK_BINS 4097   /* bins 0..4096 = floor(4096 d/N); bin 4097 = N == 0 */
#define HU_TOPK_CAP 4096

/* ------------------------------------------------------------------------------------------
 * Node sequences live in HBM as three bit-planes per 32 sites (b0, b1 = the 2-bit base code,
 * v = "is a base"), 4 words (128 sites) per uint4, node index fastest:
 *     planes[(q*3 + p) * nNodesPad + node]          q = site / 128
 * so that one wave reads 64 consecutive nodes x 16 B = 1 KiB per plane per quad, coalesced.
 * A lane owns one node and HU_READ_TILE reads; the reads' planes are wave-uniform and come in
 * through the scalar cache:   rp[((tile*WQ + q) * T + t) * 16 + p*4 + w].
 * Per (node, read, 32 sites): 2 xor + or + 2 and + 2 popcount-accumulate.
 * Grid: x = read tile (fastest: consecutive workgroups re-use one node block from L2),
 *       y = node block of 256. */
/* The (d, N) pair of a (read, node): 32 bits (d << 16 | N) or, when no read of the batch has more than 255 bases in its region
 * (N <= 255: every single-end read of up to 255 bp), 16 bits (d << 8 | N) — half the bytes the scan writes and the top-k reads
 * (6.5 + 7.7 GB per 8,192 reads at gg_97 scale with 32-bit pairs).  Everything downstream works on the canonical 32-bit form. */
template<class PT> struct HuPair;
template<> struct HuPair<uint32_t> {
	static constexpr int PPV = 4;      /* pairs per 16-byte vector */
	__device__ static inline uint32_t canon(uint32_t x) { return x; }
	__device__ static inline uint32_t pack(uint32_t c) { return c; }
	__device__ static inline void unpack(const uint4& v, uint32_t* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};
template<> struct HuPair<uint16_t> {
	static constexpr int PPV = 8;
	__device__ static inline uint32_t canon(uint32_t x) { return ((x >> 8) << 16) | (x & 0xffu); }
	__device__ static inline uint16_t pack(uint32_t c) { return (uint16_t)(((c >> 16) << 8) | (c & 0xffu)); }
	__device__ static inline void unpack(const uint4& v, uint32_t* o) {
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for(int e = 0; e < 8; ++e) o[e] = canon((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
	}
};
/* one pair, width chosen at run time (estimate kernels, given-seed lookup: one load per workgroup) */
__device__ inline uint32_t hu_pair_load(const void* pairs, size_t idx, int p16) {
	return p16 ? HuPair<uint16_t>::canon(((const uint16_t*) pairs)[idx]) : ((const uint32_t*) pairs)[idx];
}

/* The scan with the inserts of a whole tile as ONE list (k_tile_lists): entries (read t << 24 | scan position << 2 |
 * base) read from the TRANSPOSED planes of the non-profile positions (HuDbDev::colPlanes: three wave-uniform
 * 8-byte words per insert instead of three 1-KB gathers that fetch a whole quad's lines for one bit per node) and added
 * into an LDS accumulator [read][lane] (`ds_add_u32`, so the read index may be a run-time value) in the packed
 * (d << 16 | N) form of the output.  In k_seed_pdist the per-read insert loop (scalar count -> scalar entry -> three
 * dependent gathers, sixteen times per workgroup) took 36 % of the kernel for ~1 insert per read.
 * (Staging the reads' planes in LDS instead of the scalar cache was measured: no gain.) */
/* one-time: colPlanes from planes (see HuDbDev); grid (nNodesPad / 64, WQ - QM), one wave */
__global__ __launch_bounds__(64) void k_col_planes(HuDbDev db, unsigned long long* __restrict__ col) {
	const int nb = blockIdx.x, q = db.QM + blockIdx.y, lane = threadIdx.x;
	const size_t np = (size_t) db.nNodesPad, npw = np / 64;
	const int node = nb * 64 + lane;
	for(int p = 0; p < 3; ++p) {
		const uint4 v = db.planes[((size_t) q * 3 + p) * np + node];
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		for(int i = 0; i < 128; ++i) {
			const unsigned long long m = __ballot((w[i >> 5] >> (i & 31)) & 1u);
			if(lane == 0) col[(((size_t) blockIdx.y * 128 + i) * 3 + p) * npw + nb] = m;
		}
	}
}

/* one-time: HuDbDev::nodeCover from the validity plane */
__global__ __launch_bounds__(256) void k_node_cover(HuDbDev db, uint2* __restrict__ cover) {
	const int node = blockIdx.x * 256 + threadIdx.x;
	if(node >= db.nNodesPad) return;
	uint32_t f1 = 0xffffu, l1 = 0, f2 = 0xffffu, l2 = 0; bool any1 = false, any2 = false;
	for(int q = 0; q < db.WQ; ++q) {
		const uint4 v = db.planes[((size_t) q * 3 + 2) * db.nNodesPad + node];
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		for(int k = 0; k < 4; ++k) {
			if(!w[k]) continue;
			const uint32_t lo = (uint32_t) q * 128 + k * 32 + (__ffs(w[k]) - 1), hi = (uint32_t) q * 128 + k * 32 + 31 - __clz(w[k]);
			if(q < db.QM) { if(!any1) { f1 = lo; any1 = true; } l1 = hi; }
			else { if(!any2) { f2 = lo; any2 = true; } l2 = hi; }
		}
	}
	cover[node] = make_uint2((any1 ? f1 : 0xffffu) | ((any1 ? l1 : 0u) << 16), (any2 ? f2 : 0xffffu) | ((any2 ? l2 : 0u) << 16));
}
/* do two (first | last << 16) intervals share a position? (first > last: empty) */
__device__ inline bool hu_span_meets(uint32_t a, uint32_t b) {
	return (a & 0xffffu) <= (b >> 16) && (b & 0xffffu) <= (a >> 16) && (a & 0xffffu) <= (a >> 16) && (b & 0xffffu) <= (b >> 16);
}
__device__ inline bool hu_cover_meets(uint2 node, uint2 read) { return hu_span_meets(node.x, read.x) || hu_span_meets(node.y, read.y); }

/* Measured and not kept (round 2): requesting the reads' planes one or two reads ahead with explicit s_load_dwordx8/x4 asm and
 * two / four SGPR sets in rotation (left to the compiler every request is followed at once by s_waitcnt lgkmcnt(0)): 4.89 ms against
 * 3.68 ms.  Scalar loads return out of order, so a wait is always "all of them" and a request is covered by one vector block only;
 * the four register sets push the kernel to its 102 SGPRs (v_writelane spills) and the fake dependencies that keep the scheduler from
 * hoisting all sixteen request / wait pairs in front of the vector work cost more than the latency four waves per SIMD already hide. */
template<int CTRL>
__device__ inline uint32_t dpp_min_full(uint32_t v) {
	const uint32_t o = (uint32_t) __builtin_amdgcn_mov_dpp((int) v, CTRL, 0xf, 0xf, true);
	return o < v ? o : v;
}
/* L0 (the reference's seed order on a tree whose pair row is a streaming level of k_seed_refsort): the scan also CLASSIFIES every pair it writes
 * against the pivot of introsort's first partition of that read — the median of the elements at places 1, mid and last - 1 of the row, three
 * FIXED nodes whose pairs k_ref_pivots computes before the scan (hu_kern_refsort.h) — and leaves the two stopper masks of level 0 in node order,
 * one bit per node: bit (node & 63) of l0m[(read * np / 64 + node / 64) * 2 + {0: !(e < pivot), 1: !(pivot < e)}].  The sort kernel then starts
 * from 16 bytes per 64 nodes instead of a counting pass over the 128 (256) bytes of their pairs: that pass was a quarter of its time.
 * A compared-site count of zero (dist = 0 / 0, std::sort undefined) is flagged in piv[read * 4 + 3]. */
template<class PT, bool L0 = false>
__global__ __launch_bounds__(256) void k_seed_pdist2(HuDbDev db, const uint32_t* __restrict__ rp,
		const int32_t* __restrict__ tileQ, const int32_t* __restrict__ tileIns, PT* __restrict__ pairs, const int32_t* __restrict__ slotRead,
		uint32_t* __restrict__ piv = nullptr, unsigned long long* __restrict__ l0m = nullptr) {
	constexpr int T = HU_READ_TILE;
	__shared__ uint32_t acc[T][256];
	const int tile = blockIdx.x, tid = threadIdx.x;
	const int node = blockIdx.y * 256 + tid;
	const int32_t* __restrict__ ql = tileQ + (size_t) tile * (db.WQ + 1);
	const int32_t* __restrict__ til = tileIns + (size_t) tile * (T * HU_MAX_INS + 1);
	const int nq = ql[0], ne = til[0];
	uint32_t d[T], N[T];
#pragma unroll
	for(int t = 0; t < T; ++t) { d[t] = 0; N[t] = 0; }
	if(ne) {
#pragma unroll
		for(int t = 0; t < T; ++t) acc[t][tid] = 0;
	}
	const size_t np = (size_t) db.nNodesPad;
	for(int qi = 0; qi < nq; ++qi) {
		const int q = ql[1 + qi];
		const uint4 n0 = db.planes[((size_t) q * 3 + 0) * np + node], n1 = db.planes[((size_t) q * 3 + 1) * np + node], nv = db.planes[((size_t) q * 3 + 2) * np + node];
		const uint32_t* __restrict__ r = rp + (((size_t) tile * db.WQ + q) * T) * 16;
#pragma unroll
		for(int t = 0; t < T; ++t) {
			const uint32_t* rt = r + t * 16;
			uint32_t k, x;
			k = nv.x & rt[8];  x = ((n0.x ^ rt[0]) | (n1.x ^ rt[4])) & k; d[t] += __popc(x); N[t] += __popc(k);
			k = nv.y & rt[9];  x = ((n0.y ^ rt[1]) | (n1.y ^ rt[5])) & k; d[t] += __popc(x); N[t] += __popc(k);
			k = nv.z & rt[10]; x = ((n0.z ^ rt[2]) | (n1.z ^ rt[6])) & k; d[t] += __popc(x); N[t] += __popc(k);
			k = nv.w & rt[11]; x = ((n0.w ^ rt[3]) | (n1.w ^ rt[7])) & k; d[t] += __popc(x); N[t] += __popc(k);
		}
	}
	{ /* inserts: three wave-uniform 8-byte words per position (bit = node within the wave's 64) */
		const size_t npw = np / 64;
		const int nb = __builtin_amdgcn_readfirstlane(node >> 6), lane = tid & 63;
		for(int e = 0; e < ne; ++e) {
			const int ent = til[1 + e];
			const int t = ent >> 24, pos = (ent >> 2) & 0x3fffff, code = ent & 3;
			const unsigned long long* cp = db.colPlanes + ((size_t)(pos - db.QM * 128) * 3) * npw + nb;
			const unsigned long long W0 = cp[0], W1 = cp[npw], Wv = cp[2 * npw];
			const uint32_t valid = (uint32_t)(Wv >> lane) & 1u, nc = ((uint32_t)(W0 >> lane) & 1u) | (((uint32_t)(W1 >> lane) & 1u) << 1);
			const uint32_t add = valid | ((valid & (nc != (uint32_t) code ? 1u : 0u)) << 16);
			atomicAdd(&acc[t][tid], add);     /* own slot: no contention, a plain ds_add_u32 */
		}
	}
	const bool counts = L0 && node < db.nNodes && node != db.root;      /* a place of the sort: every node but the root */
	bool zero = false;
#pragma unroll
	for(int t = 0; t < T; ++t) {
		const int read = slotRead[tile * T + t];
		if(read >= 0) {
			const uint32_t v = ((d[t] << 16) | N[t]) + (ne ? acc[t][tid] : 0u);
			pairs[(size_t) read * np + node] = HuPair<PT>::pack(v);
			if(L0) { /* the pivot is wave-uniform (scalar); dist(e) < dist(pivot) <=> d_e N_p < d_p N_e, exact in 32 bits */
				const uint32_t pp = piv[(size_t) read * 4];
				const uint32_t a = (v >> 16) * (pp & 0xffffu), bb = (pp >> 16) * (v & 0xffffu);
				const unsigned long long mL = __ballot(counts && a >= bb), mR = __ballot(counts && bb >= a);
				if((tid & 63) == 0) *reinterpret_cast<ulonglong2*>(l0m + ((size_t) read * (np / 64) + (size_t)(node >> 6)) * 2) = make_ulonglong2(mL, mR);
				zero |= v == 0u;      /* d <= N: a compared-site count of zero is an all-zero pair */
			}
		}
	}
	if(L0 && __ballot(zero && counts)) { /* a node that shares no column with some read of the tile (dist = 0 / 0: std::sort is undefined, the read takes the fallback
	                                      * rule): rare on complete sequences, so which read it was is looked up only here */
#pragma unroll
		for(int t = 0; t < T; ++t) {
			const int read = slotRead[tile * T + t];
			if(read < 0) continue;
			const uint32_t v = ((d[t] << 16) | N[t]) + (ne ? acc[t][tid] : 0u);
			if(__ballot(counts && v == 0u) && (tid & 63) == 0) atomicOr(&piv[(size_t) read * 4 + 3], 1u);
		}
	}
}

/* ------------------------------------------------------------------------------------------
 * The distance-only scan (large trees).  Measured on gfx950 (profiles/ubench/valu_rate.hip): a vector instruction with a scalar
 * operand issues in 4 cycles, v_xor / v_and / v_bitop3 on vector registers alone in 2, v_bcnt_u32_b32 in 4.  The read's planes are
 * wave-uniform (scalar), so a (node, read, 32 sites) step of k_seed_pdist2 costs and(s) + xor(s) + xor(s) + bitop3 + 2 x bcnt = 22
 * cycles, 6 of them for N.  Here N is not computed at all: three operations with one scalar operand each (the floor for six
 * inputs) and one population count, 16 cycles,
 *     a = n0 ^ r0;   b = a | (n1 ^ r1);   x = b & nv & rv;   d += popc(x)
 * and the matrix holds d alone, saturated to DT — over the quads of the tile only: the mismatches at a read's LISTED inserts
 * (planes_of_codes) are left out.  Any part of the mismatches over the read's bases bounds the distance from below,
 *     d_scan <= d,   N <= L = bases of the read   =>   d / N >= d_scan / L,
 * which is all the selection needs (k_seed_topk_d); the exact (d, N) of the few candidates are recomputed there from the planes.
 * `bminD`: per (read, block of 256 nodes) the minimum d_scan over the block's nodes other than the root.
 *
 * k_seed_dscan4 is the same with FOUR nodes per lane and the reads' planes broadcast from LDS into vector registers: every
 * operation then runs on vector registers alone (2 cycles) and a step costs 10 cycles instead of 16; the LDS serves 48 broadcast
 * reads of 16 B per wave and quad against 2,560 cycles of arithmetic. */
/* Measured and not kept: two passes of eight reads per tile (32 counters per lane instead of 64: 128 VGPRs, four workgroups per CU, node planes
 * loaded twice): 2.52 ms against 2.20 — the second pass over the quads costs more than the fourth workgroup hides. */
template<class DT>
__global__ __launch_bounds__(256, 3) void k_seed_dscan4(HuDbDev db, const uint32_t* __restrict__ rp,
		const int32_t* __restrict__ tileQ, DT* __restrict__ dm, const int32_t* __restrict__ slotRead, uint32_t* __restrict__ bminD, const uint2* __restrict__ tileSpan) {
	constexpr int T = HU_READ_TILE, M = 4;
	constexpr uint32_t DMAX = (uint32_t)(DT) ~(DT) 0;
	static_assert(T == 16, "sixteen reads x sixteen dwords per quad = one dword per thread");
	__shared__ __attribute__((aligned(16))) uint32_t rpl2[2][T * 16];          /* the tile's planes of one quad, two quads in rotation: one barrier per quad */
	__shared__ __attribute__((aligned(16))) uint32_t mb[4][T][64];             /* per wave: minimum over a lane's four nodes, read by read */
	const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int node0 = blockIdx.y * (256 * M) + tid * M;                         /* four consecutive nodes per lane: a wave = one block of 256 */
	const bool live = node0 < db.nNodesPad;
	const int nl = live ? node0 : 0;
	const int32_t* __restrict__ ql = tileQ + (size_t) tile * (db.WQ + 1);
	const int nq = ql[0];
	uint32_t d[T][M];
#pragma unroll
	for(int t = 0; t < T; ++t)
#pragma unroll
		for(int m = 0; m < M; ++m) d[t][m] = 0;
	const size_t np = (size_t) db.nNodesPad;
	for(int qi = 0; qi < nq; ++qi) {
		const int q = ql[1 + qi];
		const uint32_t mine = rp[((size_t) tile * db.WQ + q) * (T * 16) + tid];
		uint4 n0[M], n1[M], nv[M];
#pragma unroll
		for(int m = 0; m < M; ++m) {
			n0[m] = db.planes[((size_t) q * 3 + 0) * np + nl + m]; n1[m] = db.planes[((size_t) q * 3 + 1) * np + nl + m]; nv[m] = db.planes[((size_t) q * 3 + 2) * np + nl + m];
		}
		uint32_t* rpl = rpl2[qi & 1];          /* written while slower waves may still read the other one; whoever passes the barrier below has finished with both older quads */
		rpl[tid] = mine;
		__syncthreads();
		/* a read's sixteen (node, word) steps are independent until the four population counts of a node meet in its counter: issued
		 * stage by stage (left alone the compiler chains all 64 instructions through two registers, and a dependent instruction
		 * waits out the pipeline with only three waves per SIMD to fill it); the next read's planes are requested before the
		 * arithmetic of this one */
		uint4 r0 = *reinterpret_cast<const uint4*>(&rpl[0]), r1 = *reinterpret_cast<const uint4*>(&rpl[4]), rv = *reinterpret_cast<const uint4*>(&rpl[8]);
#pragma unroll
		for(int t = 0; t < T; ++t) {
			const int tn = t + 1 < T ? t + 1 : t;
			const uint4 p0 = *reinterpret_cast<const uint4*>(&rpl[tn * 16]), p1 = *reinterpret_cast<const uint4*>(&rpl[tn * 16 + 4]), pv = *reinterpret_cast<const uint4*>(&rpl[tn * 16 + 8]);
			uint32_t a[M][4];
#pragma unroll
			for(int m = 0; m < M; ++m) { a[m][0] = n0[m].x ^ r0.x; a[m][1] = n0[m].y ^ r0.y; a[m][2] = n0[m].z ^ r0.z; a[m][3] = n0[m].w ^ r0.w; }
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for(int m = 0; m < M; ++m) {
				a[m][0] = __builtin_amdgcn_bitop3_b32(a[m][0], n1[m].x, r1.x, 0xf6); a[m][1] = __builtin_amdgcn_bitop3_b32(a[m][1], n1[m].y, r1.y, 0xf6);
				a[m][2] = __builtin_amdgcn_bitop3_b32(a[m][2], n1[m].z, r1.z, 0xf6); a[m][3] = __builtin_amdgcn_bitop3_b32(a[m][3], n1[m].w, r1.w, 0xf6);
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for(int m = 0; m < M; ++m) {
				a[m][0] = __builtin_amdgcn_bitop3_b32(a[m][0], nv[m].x, rv.x, 0x80); a[m][1] = __builtin_amdgcn_bitop3_b32(a[m][1], nv[m].y, rv.y, 0x80);
				a[m][2] = __builtin_amdgcn_bitop3_b32(a[m][2], nv[m].z, rv.z, 0x80); a[m][3] = __builtin_amdgcn_bitop3_b32(a[m][3], nv[m].w, rv.w, 0x80);
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for(int w = 0; w < 4; ++w) {
#pragma unroll
				for(int m = 0; m < M; ++m) asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d[t][m]) : "v"(a[m][w]));    /* count and accumulate in one (the compiler splits it into count + add3) */
			}
			__builtin_amdgcn_sched_barrier(0);
			r0 = p0; r1 = p1; rv = pv;
		}
	}
	/* a node that shares no position with ANY read of the tile (N = 0 for all of them: d_scan = 0 without being near) stays out of the minima */
	uint32_t skip[M];
	const uint2 tsp = tileSpan[tile];
#pragma unroll
	for(int m = 0; m < M; ++m) skip[m] = live && node0 + m < db.nNodes && node0 + m != db.root && hu_cover_meets(db.nodeCover[nl + m], tsp) ? 0u : 0xffffffffu;
#pragma unroll
	for(int t = 0; t < T; ++t) {
		const int read = slotRead[tile * T + t];
		uint32_t v[M];
#pragma unroll
		for(int m = 0; m < M; ++m) v[m] = min(d[t][m], DMAX);
		if(read >= 0 && live) {
			DT* dst = dm + (size_t) read * np + node0;
			if(sizeof(DT) == 1) *reinterpret_cast<uint32_t*>(dst) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
			else *reinterpret_cast<uint2*>(dst) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
		}
		mb[wave][t][lane] = min(min(v[0] | skip[0], v[1] | skip[1]), min(v[2] | skip[2], v[3] | skip[3]));
	}
	__syncthreads();
	{ /* the wave's block: read t = lane / 4, four lanes x sixteen values, then two steps inside the quad of lanes */
		const int t = lane >> 2, sg = lane & 3;
		const uint4* row = reinterpret_cast<const uint4*>(&mb[wave][t][sg * 16]);
		const uint4 a = row[0], b = row[1], c = row[2], e = row[3];
		uint32_t m = min(min(min(a.x, a.y), min(a.z, a.w)), min(min(b.x, b.y), min(b.z, b.w)));
		m = min(m, min(min(min(c.x, c.y), min(c.z, c.w)), min(min(e.x, e.y), min(e.z, e.w))));
		m = dpp_min_full<0xB1>(m); m = dpp_min_full<0x4E>(m);
		const int read = slotRead[tile * T + t], blk = blockIdx.y * M + wave;
		if(sg == 0 && read >= 0 && blk < db.nNodesPad / 256) bminD[(size_t) read * (db.nNodesPad / 256) + blk] = m;
	}
}

/* Measured and not kept (round 3): the PAIR scan in the form of k_seed_dscan4 — two nodes per lane (64 + 64 counters), the reads' planes broadcast
 * from LDS, every operation on vector registers: xor, bitop3, and, and + two population counts = 16 cycles per (node, read, 32 sites) step
 * against the 22 of k_seed_pdist2.  134 VGPRs, bit-identical pairs, 3.93 ms against 3.52: with two nodes per lane the 48 broadcast reads per
 * wave and quad stand against 2,048 cycles of arithmetic instead of 2,560 for four, and the LDS becomes the bound; four nodes per lane with
 * both counters do not fit the register file — and taking the tile's sixteen reads in two passes of eight to make them fit (167 VGPRs, the node
 * planes loaded twice) measured 4.36 ms.  One node per lane with scalar read planes stays: 99 % of its issue slots. */
/* exact order-preserving integer image of dist = d/N for d <= N < 2^16: two different
 * fractions differ by more than 2^-32, so floor(d * 2^39 / N) separates them; N == 0 (the
 * reference's 0/0 = NaN) sorts last.  Ties are broken by node id in the low 24 bits. */
__device__ inline unsigned long long seed_key(uint32_t d, uint32_t N, uint32_t node) {
	unsigned long long q = N ? (((unsigned long long) d) << 39) / N : ((1ull << 40) - 1);
	return (q << 24) | node;
}

/* bin of a (d, N) pair: floor(4096 d / N), or HU_TOPK_BINS for N == 0; pairs at or beyond `limit`
 * (in units of 1/4096) collapse into bin `limit` without a division */
__device__ inline uint32_t seed_bin(uint32_t d, uint32_t N, uint32_t limit) {
	if(N == 0) return limit < HU_TOPK_BINS ? limit : HU_TOPK_BINS;
	if(limit < HU_TOPK_BINS && (unsigned long long) d * 4096ull >= (unsigned long long) limit * N) return limit;
	return (d << 12) / N;
}

template<class PT>
__global__ __launch_bounds__(256) void k_seed_topk(HuDbDev db, const PT* __restrict__ pairs, double maxHeight,
		int maxNSeed, int32_t* __restrict__ seedCnt, int32_t* __restrict__ seedId, uint32_t* __restrict__ seedDN, int fastMinNodes) {
	__shared__ unsigned long long keys[HU_TOPK_CAP];
	static_assert(sizeof(unsigned long long) * HU_TOPK_CAP >= sizeof(uint32_t) * (HU_TOPK_BINS + 1), "histogram must fit the key buffer");
	uint32_t* hist = reinterpret_cast<uint32_t*>(keys);   /* the histogram is dead once the threshold bin is known: the keys take its
	                                                       * place (33 KB instead of 50 KB of LDS: four workgroups per CU) */
	__shared__ uint32_t chunk[256];
	__shared__ uint32_t sh[5];
	const int read = blockIdx.x, tid = threadIdx.x;
	constexpr int PPV = HuPair<PT>::PPV, NPI = 256 * HuPair<PT>::PPV;      /* pairs per 16-byte vector, nodes per workgroup pass */
	const PT* __restrict__ pr = pairs + (size_t) read * db.nNodesPad;
	const int per = (HU_TOPK_BINS + 1 + 255) / 256;
	const bool useHeight = !(maxHeight == INFINITY);
	int32_t* outId = seedId + (size_t) read * HU_MAX_SEEDS;
	uint32_t* outDN = seedDN + (size_t) read * HU_MAX_SEEDS;
	/* Fast path (large trees, no height filter): the threshold bin is ESTIMATED from a histogram of one eighth
	 * of the pairs (every eighth 128-byte line: 32 consecutive nodes, the phase advancing line by line), aiming at ~200
	 * survivors; the exact pass then collects every pair at or below that bin.  The result is exact whenever
	 * the survivors number at least max_nseed (they then contain the max_nseed smallest keys, all ties of the
	 * last bin included); otherwise, or when they overflow the key buffer, the two-pass path below runs.
	 * The pair matrix is read 1.125 times instead of twice, four 16-byte groups in flight per thread. */
	if(!useHeight && db.nNodes >= fastMinNodes && db.nNodes - 1 >= maxNSeed) {
		const uint32_t limitS = 1024;
		for(int i = tid; i <= (int) limitS; i += 256) hist[i] = 0;
		if(tid == 0) { sh[3] = 0; sh[4] = 0; }
		__syncthreads();
		const int nIt = (db.nNodes + NPI - 1) / NPI;
		for(int it = (8 - ((tid >> 3) & 7)) & 7; it < nIt; it += 8) { /* whole 128-byte lines (8 lanes x 16 B = 32 or 64 nodes): iterations with (it + tid / 8) % 8 == 0 */
			const int base = it * NPI + tid * PPV;
			if(base >= db.nNodes) break;
			const uint4 v4 = *reinterpret_cast<const uint4*>(pr + base);
			uint32_t vv[PPV]; HuPair<PT>::unpack(v4, vv);
#pragma unroll
			for(int e = 0; e < PPV; ++e) {
				const int node = base + e;
				if(node >= db.nNodes || node == db.root) continue;
				const uint32_t bin = seed_bin(vv[e] >> 16, vv[e] & 0xffffu, limitS);
				if(bin < limitS) atomicAdd(&hist[bin], 1u);
			}
		}
		__syncthreads();
		{
			uint32_t sm = 0;
			for(int i = tid * 4; i < tid * 4 + 4; ++i) sm += hist[i];     /* 256 x 4 = the 1024 bins below the limit */
			chunk[tid] = sm;
		}
		__syncthreads();
		if(tid == 0) {
			const uint32_t target = 24;   /* sampled pairs at or below the threshold: ~192 expected in the full matrix */
			uint32_t cum = 0; int c = 0;
			while(c < 256 && cum + chunk[c] < target) { cum += chunk[c]; ++c; }
			if(c >= 256) sh[4] = 1;       /* too few near pairs in the sample: exact path */
			else { int b = c * 4; while(cum + hist[b] < target) { cum += hist[b]; ++b; } sh[1] = (uint32_t) b; }
		}
		__syncthreads();
		if(!sh[4]) {
			const uint32_t thr = sh[1];
			__syncthreads();
			for(int it0 = 0; it0 < nIt; it0 += 4) {
				uint4 v4[4];
#pragma unroll
				for(int k = 0; k < 4; ++k) {
					const int base = (it0 + k) * NPI + tid * PPV;
					v4[k] = base < db.nNodes ? *reinterpret_cast<const uint4*>(pr + base) : make_uint4(0, 0, 0, 0);
				}
#pragma unroll
				for(int k = 0; k < 4; ++k) {
					const int base = (it0 + k) * NPI + tid * PPV;
					uint32_t vv[PPV]; HuPair<PT>::unpack(v4[k], vv);
#pragma unroll
					for(int e = 0; e < PPV; ++e) {
						const int node = base + e;
						if(node >= db.nNodes || node == db.root) continue;
						const uint32_t d = vv[e] >> 16, N = vv[e] & 0xffffu;
						if(N != 0 && (unsigned long long) d * 4096ull < (unsigned long long)(thr + 1) * N) { /* floor(4096 d / N) <= thr */
							const uint32_t slot = atomicAdd(&sh[3], 1u);
							if(slot < HU_TOPK_CAP) keys[slot] = seed_key(d, N, (uint32_t) node);
						}
					}
				}
			}
			__syncthreads();
			const uint32_t got = sh[3];
			const uint32_t need = (uint32_t) maxNSeed;            /* nNodes - 1 >= max_nseed here */
			if(got >= need && got <= HU_TOPK_CAP) {
				uint32_t n2 = 1;
				while(n2 < got) n2 <<= 1;
				for(uint32_t i = got + tid; i < n2; i += 256) keys[i] = ~0ull;
				__syncthreads();
				for(uint32_t k = 2; k <= n2; k <<= 1)
					for(uint32_t j = k >> 1; j > 0; j >>= 1) {
						for(uint32_t i = tid; i < n2; i += 256) {
							uint32_t l = i ^ j;
							if(l > i) {
								unsigned long long a = keys[i], b = keys[l];
								bool up = (i & k) == 0;
								if((a > b) == up) { keys[i] = b; keys[l] = a; }
							}
						}
						__syncthreads();
					}
				for(uint32_t i = tid; i < need; i += 256) {
					uint32_t node = (uint32_t)(keys[i] & 0xffffffu);
					outId[i] = (int32_t) node; outDN[i] = HuPair<PT>::canon(pr[node]);
				}
				if(tid == 0) seedCnt[read] = (int32_t) need;
				return;
			}
		}
		__syncthreads();
	}
	/* the wanted seeds are the nearest nodes: histogram only distances < 1/4 first (the bulk of the
	 * tree is farther and would serialise on a few hot LDS counters); fall back to all bins if short */
	uint32_t limit = 1024;
	for(int attempt = 0; attempt < 2; ++attempt) {
		for(int i = tid; i <= HU_TOPK_BINS; i += 256) hist[i] = 0;
		__syncthreads();
		uint32_t over = 0;
		for(int base = tid * PPV; base < db.nNodes; base += NPI) {
			const uint4 v4 = *reinterpret_cast<const uint4*>(pr + base);
			uint32_t vv[PPV]; HuPair<PT>::unpack(v4, vv);
#pragma unroll
			for(int e = 0; e < PPV; ++e) {
				const int node = base + e;
				if(node >= db.nNodes || node == db.root || (useHeight && !(db.height[node] <= maxHeight))) continue;
				const uint32_t bin = seed_bin(vv[e] >> 16, vv[e] & 0xffffu, limit);
				if(bin == limit && limit < HU_TOPK_BINS) over++;
				else atomicAdd(&hist[bin], 1u);
			}
		}
		for(int m = 32; m > 0; m >>= 1) over += __shfl_xor(over, m);
		if((tid & 63) == 0 && over) atomicAdd(&hist[limit], over);
		__syncthreads();
		{
			uint32_t sm = 0;
			for(int i = tid * per; i < (tid + 1) * per && i <= HU_TOPK_BINS; ++i) sm += hist[i];
			chunk[tid] = sm;
		}
		__syncthreads();
		if(tid == 0) {
			uint32_t total = 0;
			for(int i = 0; i < 256; ++i) total += chunk[i];
			uint32_t need = total < (uint32_t) maxNSeed ? total : (uint32_t) maxNSeed;
			uint32_t cum = 0; int c = 0;
			while(c < 255 && cum + chunk[c] < need) { cum += chunk[c]; ++c; }
			int b = c * per;
			while(b < HU_TOPK_BINS && cum + hist[b] < need) { cum += hist[b]; ++b; }
			sh[0] = need; sh[1] = (uint32_t) b; sh[2] = cum + hist[b]; sh[3] = 0;
			sh[4] = (limit < HU_TOPK_BINS && (uint32_t) b >= limit && need > 0) ? 1u : 0u; /* threshold fell into the overflow bin */
		}
		__syncthreads();
		if(!sh[4]) break;
		limit = HU_TOPK_BINS;
		__syncthreads();
	}
	const uint32_t need = sh[0], thr = sh[1], cntLE = sh[2];
	if(need == 0) { if(tid == 0) seedCnt[read] = 0; return; }
	if(cntLE <= HU_TOPK_CAP) {
		for(int base = tid * PPV; base < db.nNodes; base += NPI) {
			const uint4 v4 = *reinterpret_cast<const uint4*>(pr + base);
			uint32_t vv[PPV]; HuPair<PT>::unpack(v4, vv);
#pragma unroll
			for(int e = 0; e < PPV; ++e) {
				const int node = base + e;
				if(node >= db.nNodes || node == db.root || (useHeight && !(db.height[node] <= maxHeight))) continue;
				const uint32_t d = vv[e] >> 16, N = vv[e] & 0xffffu;
				if(seed_bin(d, N, thr + 1) <= thr) { uint32_t slot = atomicAdd(&sh[3], 1u); keys[slot] = seed_key(d, N, (uint32_t) node); }
			}
		}
		__syncthreads();
		uint32_t n2 = 1;
		while(n2 < cntLE) n2 <<= 1;
		for(uint32_t i = cntLE + tid; i < n2; i += 256) keys[i] = ~0ull;
		__syncthreads();
		for(uint32_t k = 2; k <= n2; k <<= 1)
			for(uint32_t j = k >> 1; j > 0; j >>= 1) {
				for(uint32_t i = tid; i < n2; i += 256) {
					uint32_t l = i ^ j;
					if(l > i) {
						unsigned long long a = keys[i], b = keys[l];
						bool up = (i & k) == 0;
						if((a > b) == up) { keys[i] = b; keys[l] = a; }
					}
				}
				__syncthreads();
			}
		for(uint32_t i = tid; i < need; i += 256) {
			uint32_t node = (uint32_t)(keys[i] & 0xffffffu);
			outId[i] = (int32_t) node; outDN[i] = HuPair<PT>::canon(pr[node]);
		}
	}
	else { /* degenerate tie mass: one exact minimum per pass, bounded by max_nseed passes */
		unsigned long long last = 0; bool first = true;
		for(uint32_t s = 0; s < need; ++s) {
			unsigned long long best = ~0ull;
			for(int node = tid; node < db.nNodes; node += 256) {
				if(node == db.root || (useHeight && !(db.height[node] <= maxHeight))) continue;
				const uint32_t v = HuPair<PT>::canon(pr[node]), d = v >> 16, N = v & 0xffffu;
				if(seed_bin(d, N, thr + 1) > thr) continue;
				unsigned long long k = seed_key(d, N, (uint32_t) node);
				if((first || k > last) && k < best) best = k;
			}
			for(int m = 32; m > 0; m >>= 1) { unsigned long long o = __shfl_xor(best, m); best = o < best ? o : best; }
			if((tid & 63) == 0) keys[tid >> 6] = best;
			__syncthreads();
			best = keys[0];
			for(int wv = 1; wv < 4; ++wv) best = keys[wv] < best ? keys[wv] : best;
			__syncthreads();
			last = best; first = false;
			if(tid == 0) { uint32_t node = (uint32_t)(best & 0xffffffu); outId[s] = (int32_t) node; outDN[s] = HuPair<PT>::canon(pr[node]); }
		}
	}
	if(tid == 0) seedCnt[read] = (int32_t) need;
}

/* ------------------------------------------------------------------------------------------
 * The exact (d, N) of one (read, node) from the bit-planes: what k_seed_pdist2 writes for every pair, here on demand for the few
 * pairs the distance-only path needs (candidates of the top-k, the seeds' parents, given seeds).  The read's planes are found
 * through its slot in the scan's tiling, its quads through its quad bitmap, its listed inserts through `ins`. */
struct HuReadPlanes { const uint32_t* rp; const uint32_t* rq; const int32_t* ins; const int32_t* readSlot; const uint2* rspan; /* per read: its two position intervals, as HuDbDev::nodeCover */ };
__device__ inline uint32_t pair_exact(const HuDbDev& db, const HuReadPlanes& R, int r, int node, bool listed = true) {
	const int slot = R.readSlot[r], tile = slot / HU_READ_TILE, t = slot % HU_READ_TILE;
	const int nw32 = (db.WQ + 31) / 32;
	const size_t np = (size_t) db.nNodesPad;
	uint32_t d = 0, N = 0;
	for(int w = 0; w < nw32; ++w) {
		uint32_t m = R.rq[(size_t) r * nw32 + w];
		while(m) {
			const int q = w * 32 + __ffs(m) - 1; m &= m - 1;
			const uint32_t* __restrict__ rt = R.rp + (((size_t) tile * db.WQ + q) * HU_READ_TILE + t) * 16;
			const uint4 n0 = db.planes[((size_t) q * 3 + 0) * np + node], n1 = db.planes[((size_t) q * 3 + 1) * np + node], nv = db.planes[((size_t) q * 3 + 2) * np + node];
			uint32_t k, x;
			k = nv.x & rt[8];  x = ((n0.x ^ rt[0]) | (n1.x ^ rt[4])) & k; d += __popc(x); N += __popc(k);
			k = nv.y & rt[9];  x = ((n0.y ^ rt[1]) | (n1.y ^ rt[5])) & k; d += __popc(x); N += __popc(k);
			k = nv.z & rt[10]; x = ((n0.z ^ rt[2]) | (n1.z ^ rt[6])) & k; d += __popc(x); N += __popc(k);
			k = nv.w & rt[11]; x = ((n0.w ^ rt[3]) | (n1.w ^ rt[7])) & k; d += __popc(x); N += __popc(k);
		}
	}
	const int32_t* __restrict__ il = R.ins + (size_t) r * (HU_MAX_INS + 1);
	const int cnt = listed ? il[0] : 0;
	for(int e = 0; e < cnt; ++e) {
		const int ent = il[1 + e], pos = ent >> 2, code = ent & 3;
		const int q = pos >> 7, w = (pos >> 5) & 3, bit = pos & 31;
		const uint32_t* pw = reinterpret_cast<const uint32_t*>(db.planes + ((size_t) q * 3) * np + node) + w;
		const uint32_t w0 = pw[0], w1 = pw[np * 4], wv = pw[np * 8];
		const uint32_t valid = (wv >> bit) & 1u, nc = ((w0 >> bit) & 1u) | (((w1 >> bit) & 1u) << 1);
		N += valid; d += valid & (nc != (uint32_t) code ? 1u : 0u);
	}
	return (d << 16) | N;
}
/* bases of the read inside its region: an upper bound of every N of the read */
__device__ inline uint32_t read_bases(const HuDbDev& db, const HuReadPlanes& R, int r) {
	const int slot = R.readSlot[r], tile = slot / HU_READ_TILE, t = slot % HU_READ_TILE;
	const int nw32 = (db.WQ + 31) / 32;
	uint32_t L = (uint32_t) R.ins[(size_t) r * (HU_MAX_INS + 1)];
	for(int w = 0; w < nw32; ++w) {
		uint32_t m = R.rq[(size_t) r * nw32 + w];
		while(m) {
			const int q = w * 32 + __ffs(m) - 1; m &= m - 1;
			const uint32_t* __restrict__ rt = R.rp + (((size_t) tile * db.WQ + q) * HU_READ_TILE + t) * 16;
			L += __popc(rt[8]) + __popc(rt[9]) + __popc(rt[10]) + __popc(rt[11]);
		}
	}
	return L;
}

/* every (d, N) of one read (hu_batch_get_pdist after the distance-only scan) */
__global__ __launch_bounds__(256) void k_pairs_of_read(HuDbDev db, HuReadPlanes R, int read, uint32_t* __restrict__ out, uint32_t* __restrict__ outScan) {
	const int node = blockIdx.x * 256 + threadIdx.x;
	if(node < db.nNodes) { out[node] = pair_exact(db, R, read, node); if(outScan) outScan[node] = pair_exact(db, R, read, node, false); }
}
/* (d, N) of the seeds' parents from the pair matrix (the paths that keep one): estimateSeq's pDist(v.seq) */
__global__ void k_parent_pairs(HuDbDev db, int n, const void* __restrict__ pairs, int p16, const int32_t* __restrict__ seedCnt, const int32_t* __restrict__ seedId,
		uint32_t* __restrict__ parDN) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if(i >= n * HU_MAX_SEEDS) return;
	const int r = i / HU_MAX_SEEDS, sl = i % HU_MAX_SEEDS;
	if(sl < seedCnt[r]) parDN[i] = hu_pair_load(pairs, (size_t) r * db.nNodesPad + db.parent[seedId[i]], p16);
}

/* The exact selection over ALL nodes with the pairs recomputed from the planes — for the reads the block path of k_seed_topk_d cannot
 * serve (ties beyond its buffers, a saturated distance).  Histogram of floor(4096 d / N), keys of the bins up to the one that
 * completes max_nseed, sorted; with more ties than the buffer holds, one exact minimum per pass (as in k_seed_topk). */
__device__ inline void topk_exact_recompute(const HuDbDev& db, const HuReadPlanes& R, int read, int maxNSeed, unsigned long long* keys, uint32_t* chunk,
		uint32_t* sh, int32_t* outId, uint32_t* outDN, uint32_t* outPar, int32_t* seedCnt) {
	const int tid = threadIdx.x;
	uint32_t* hist = reinterpret_cast<uint32_t*>(keys);
	const int per = (HU_TOPK_BINS + 1 + 255) / 256;
	__syncthreads();
	for(int i = tid; i <= HU_TOPK_BINS; i += 256) hist[i] = 0;
	__syncthreads();
	for(int node = tid; node < db.nNodes; node += 256) {
		if(node == db.root) continue;
		const uint32_t c = pair_exact(db, R, read, node);
		atomicAdd(&hist[seed_bin(c >> 16, c & 0xffffu, HU_TOPK_BINS)], 1u);
	}
	__syncthreads();
	{
		uint32_t sm = 0;
		for(int i = tid * per; i < (tid + 1) * per && i <= HU_TOPK_BINS; ++i) sm += hist[i];
		chunk[tid] = sm;
	}
	__syncthreads();
	if(tid == 0) {
		uint32_t total = 0;
		for(int i = 0; i < 256; ++i) total += chunk[i];
		const uint32_t need = total < (uint32_t) maxNSeed ? total : (uint32_t) maxNSeed;
		uint32_t cum = 0; int c = 0;
		while(c < 255 && cum + chunk[c] < need) { cum += chunk[c]; ++c; }
		int b = c * per;
		while(b < HU_TOPK_BINS && cum + hist[b] < need) { cum += hist[b]; ++b; }
		sh[0] = need; sh[1] = (uint32_t) b; sh[2] = cum + hist[b]; sh[3] = 0;
	}
	__syncthreads();
	const uint32_t need = sh[0], thr = sh[1], cntLE = sh[2];
	if(need == 0) { if(tid == 0) seedCnt[read] = 0; return; }
	__syncthreads();
	if(cntLE <= HU_TOPK_CAP) {
		for(int node = tid; node < db.nNodes; node += 256) {
			if(node == db.root) continue;
			const uint32_t c = pair_exact(db, R, read, node), d = c >> 16, N = c & 0xffffu;
			if(seed_bin(d, N, thr + 1) <= thr) { const uint32_t slot = atomicAdd(&sh[3], 1u); keys[slot] = seed_key(d, N, (uint32_t) node); }
		}
		__syncthreads();
		uint32_t n2 = 1;
		while(n2 < cntLE) n2 <<= 1;
		for(uint32_t i = cntLE + tid; i < n2; i += 256) keys[i] = ~0ull;
		__syncthreads();
		for(uint32_t k = 2; k <= n2; k <<= 1)
			for(uint32_t j = k >> 1; j > 0; j >>= 1) {
				for(uint32_t i = tid; i < n2; i += 256) {
					const uint32_t l = i ^ j;
					if(l > i) {
						const unsigned long long a = keys[i], b = keys[l];
						const bool up = (i & k) == 0;
						if((a > b) == up) { keys[i] = b; keys[l] = a; }
					}
				}
				__syncthreads();
			}
		for(uint32_t i = tid; i < need; i += 256) {
			const int node = (int)(keys[i] & 0xffffffu);
			outId[i] = node; outDN[i] = pair_exact(db, R, read, node); outPar[i] = pair_exact(db, R, read, db.parent[node]);
		}
	}
	else {
		unsigned long long last = 0; bool first = true;
		for(uint32_t s = 0; s < need; ++s) {
			unsigned long long best = ~0ull;
			for(int node = tid; node < db.nNodes; node += 256) {
				if(node == db.root) continue;
				const uint32_t c = pair_exact(db, R, read, node), d = c >> 16, N = c & 0xffffu;
				if(seed_bin(d, N, thr + 1) > thr) continue;
				const unsigned long long k = seed_key(d, N, (uint32_t) node);
				if((first || k > last) && k < best) best = k;
			}
			for(int m = 32; m > 0; m >>= 1) { const unsigned long long o = __shfl_xor(best, m); best = o < best ? o : best; }
			if((tid & 63) == 0) keys[tid >> 6] = best;
			__syncthreads();
			best = keys[0];
			for(int wv = 1; wv < 4; ++wv) best = keys[wv] < best ? keys[wv] : best;
			__syncthreads();
			last = best; first = false;
			if(tid == 0) { const int node = (int)(best & 0xffffffu); outId[s] = node; outDN[s] = pair_exact(db, R, read, node); outPar[s] = pair_exact(db, R, read, db.parent[node]); }
		}
	}
	if(tid == 0) seedCnt[read] = (int32_t) need;
}

/* std::sort + truncation to max_nseed after the distance-only scan: one workgroup per read.
 *   d_scan <= d (the scan leaves the listed inserts out), L = bases of the read >= every N: a node at distance x has d_scan <= x L.
 *   1. Dsel = the max_nseed-th smallest block minimum of d_scan (bitwise selection on one wave): at least max_nseed blocks hold a
 *      node with d_scan <= Dsel; no other block does.
 *   2. histogram of d_scan over those blocks -> Dk, the max_nseed-th smallest d_scan overall; C1 = {d_scan <= Dk} (>= max_nseed nodes).
 *   3. exact (d, N) of C1 from the planes; (d1, N1) = its max_nseed-th smallest key, an upper bound of the wanted distance.
 *   4. every wanted node has d_scan <= D1 = floor(d1 L / N1); if D1 > Dk the nodes with Dk < d_scan <= D1 join.
 *   5. (d, N) of the seeds and of their parents written in (dist, node id) order.
 * Candidates are absorbed into a kept set of the max_nseed best by counting ranks on the exact keys — all at once when the
 * histogram says they fit the buffers (the usual case: ~80 candidates), else three blocks at a time.  Comparisons on the saturated
 * matrix are valid below the saturation value; a read that reaches it, or has fewer than max_nseed nodes with N > 0, takes
 * topk_exact_recompute; a read without bases gets the first nodes by id with (0, 0) like k_seed_topk.
 * stat (optional): reads served here, blocks, candidates, reads passed on. */
/* The straight path of the top-k after the distance-only scan, as its own small kernel (with the rare paths in the same kernel the
 * common path's code ran 40 % slower, 0.47 against 0.33 ms per 8,192 reads: a matter of code generation that was not pinned down): Dsel by bitwise selection, one histogram pass, C1, the bound D1, a second absorb when needed.  Anything else — fewer
 * than max_nseed nodes within the chosen blocks, a candidate with N = 0 among the kept, buffers too small, saturation — and the read
 * is appended to `retry` for k_seed_topk_d<DT, true>, the general launch below, which also documents the algorithm. */
template<class DT>
__global__ __launch_bounds__(256, 4) void k_seed_topk_straight(HuDbDev db, const DT* __restrict__ dm, const uint32_t* __restrict__ bminD, HuReadPlanes R,
		int maxNSeed, int32_t* __restrict__ seedCnt, int32_t* __restrict__ seedId, uint32_t* __restrict__ seedDN, uint32_t* __restrict__ parDN,
		uint32_t* __restrict__ stat, int32_t* __restrict__ retry, int leaveAll) {
	constexpr uint32_t DMAX = (uint32_t)(DT) ~(DT) 0;
	constexpr int NBITS = 8 * (int) sizeof(DT);
	constexpr uint32_t CAP = 1024, HB = 1024, NONE = 0xffffffffu;
	static_assert(3 * 256 + HU_MAX_SEEDS <= CAP, "three blocks and the kept set fit the candidate buffers");
	__shared__ unsigned long long keys[HU_TOPK_CAP];        /* 32 KB, carved up below */
	__shared__ uint32_t chunk[256];
	__shared__ uint32_t sh[8];
	uint32_t* bm = reinterpret_cast<uint32_t*>(keys);                                   /* [2048] block minima          0 ..  8 KB */
	unsigned short* sel = reinterpret_cast<unsigned short*>(keys + 1024);               /* [2048] chosen blocks         8 .. 12 KB */
	uint32_t* hist = reinterpret_cast<uint32_t*>(keys + 1536);                          /* [HB] histogram of d_scan    12 .. 16 KB */
	unsigned long long* ck = keys + 2048;                                               /* [CAP] candidate keys        16 .. 24 KB */
	uint32_t* cp = reinterpret_cast<uint32_t*>(keys + 3072);                            /* [CAP] candidate (d, N)      24 .. 28 KB */
	uint32_t* cn = reinterpret_cast<uint32_t*>(keys + 3584);                            /* [CAP] candidate nodes       28 .. 32 KB */
	const int read = blockIdx.x, tid = threadIdx.x;
	if(leaveAll) { if(tid == 0) retry[1 + atomicAdd(&retry[0], 1)] = read; return; }      /* test knob: every read through the general launch */
	const size_t np = (size_t) db.nNodesPad;
	const DT* __restrict__ dr = dm + (size_t) read * np;
	const int nBlk = db.nNodesPad / 256;
	int32_t* outId = seedId + (size_t) read * HU_MAX_SEEDS;
	uint32_t* outDN = seedDN + (size_t) read * HU_MAX_SEEDS;
	uint32_t* outPar = parDN + (size_t) read * HU_MAX_SEEDS;
	const uint32_t need = (uint32_t) maxNSeed;
	const int per = (nBlk + 63) / 64;                                                   /* block minima per lane of wave 0: <= 32 */
	for(int i = tid; i < per * 64; i += 256) bm[i] = i < nBlk ? bminD[(size_t) read * nBlk + i] : NONE;
	if(tid == 0) { sh[0] = NONE; sh[2] = 0; sh[3] = 0; sh[5] = read_bases(db, R, read); }
	__syncthreads();
	const uint32_t L = sh[5];
	const uint2 rsp = R.rspan[read];
	if(L == 0) { /* no base in the region (a read that was not aligned): every N is 0, the order is the node ids' */
		if(tid == 0) {
			int k = 0;
			for(int node = 0; node < db.nNodes && k < maxNSeed; ++node) if(node != db.root) { outId[k] = node; outDN[k] = 0; outPar[k] = 0; ++k; }
			seedCnt[read] = k;
		}
		return;
	}
	if(tid < 64) { /* the max_nseed-th smallest, bit by bit from the top: the largest r with #{x < r} < max_nseed */
		uint32_t x[32];
#pragma unroll
		for(int k = 0; k < 32; ++k) x[k] = k < per ? bm[k * 64 + tid] : NONE;
		uint32_t res = 0;
		if(per <= 16) {
			for(int bit = NBITS; bit >= 0; --bit) {
				const uint32_t trial = res | (1u << bit);
				int cnt = 0;
#pragma unroll
				for(int k = 0; k < 16; ++k) cnt += __popcll(__ballot(x[k] < trial));
				if(cnt < maxNSeed) res = trial;
			}
		} else {
			for(int bit = NBITS; bit >= 0; --bit) {
				const uint32_t trial = res | (1u << bit);
				int cnt = 0;
#pragma unroll
				for(int k = 0; k < 32; ++k) cnt += __popcll(__ballot(x[k] < trial));
				if(cnt < maxNSeed) res = trial;
			}
		}
		if(tid == 0) sh[0] = res;
	}
	__syncthreads();
	const uint32_t Dsel = sh[0];
	bool served = false;
	if(Dsel < DMAX && Dsel < HB) {
		/* blocks with minimum <= lim -> sel; returns their number */
		auto choose = [&](uint32_t lim) __attribute__((always_inline)) {
			if(tid == 0) sh[2] = 0;
			__syncthreads();
			for(int i = tid; i < nBlk; i += 256)
				if(bm[i] <= lim) sel[atomicAdd(&sh[2], 1u)] = (unsigned short) i;
			__syncthreads();
			return (int) sh[2];
		};
		/* the nodes of sel[g0, g1) with lo < d_scan <= lim (lo == NONE: no lower limit): histogram (pass 0) or list at sh[3] (pass 1) */
		auto sweep = [&](int g0, int g1, uint32_t lo, uint32_t lim, int pass) __attribute__((always_inline)) {
#pragma unroll 1
			for(int s0 = g0; s0 < g1; s0 += 16) {     /* sixteen loads in flight per thread: scalar block base + one lane offset */
				uint32_t pv[16];
#pragma unroll
				for(int k = 0; k < 16; ++k) pv[k] = s0 + k < g1 ? (uint32_t) (dr + (size_t) __builtin_amdgcn_readfirstlane((int) sel[s0 + k]) * 256)[tid] : NONE;
#pragma unroll
				for(int k = 0; k < 16; ++k) {
					if(s0 + k >= g1) continue;
					const int node = __builtin_amdgcn_readfirstlane((int) sel[s0 + k]) * 256 + tid;
					if(node >= db.nNodes || node == db.root || pv[k] > lim) continue;
					if(pv[k] == 0 && !hu_cover_meets(db.nodeCover[node], rsp)) continue;        /* no shared position: N = 0 exactly, never a candidate */
					if(pass == 0) atomicAdd(&hist[pv[k]], 1u);
					else if(lo == NONE || pv[k] > lo) { const uint32_t slot = atomicAdd(&sh[3], 1u); if(slot < CAP) cn[slot] = (uint32_t) node; }
				}
			}
			__syncthreads();
		};
		uint32_t nb = 0;                          /* kept candidates: slots [0, nb) of ck / cp / cn, the max_nseed best so far once there are that many */
		uint32_t seen = 0;
		/* candidates with lo < d_scan <= lim join the kept set; `expect` = their number if known */
		auto absorb = [&](uint32_t lo, uint32_t lim, uint32_t expect) __attribute__((always_inline)) -> bool {
			const int nsel = choose(lim);
			const int group = expect != NONE && nb + expect <= CAP ? (nsel > 0 ? nsel : 1) : 3;
			for(int g0 = 0; g0 < nsel; g0 += group) {
				if(tid == 0) sh[3] = nb;
				__syncthreads();
				sweep(g0, min(g0 + group, nsel), lo, lim, 1);
				const uint32_t cnt = sh[3];
				if(cnt > CAP) return false;
				seen += cnt - nb;
				for(uint32_t i = nb + tid; i < cnt; i += 256) { const uint32_t c = pair_exact(db, R, read, (int) cn[i]); cp[i] = c; ck[i] = seed_key(c >> 16, c & 0xffffu, cn[i]); }
				__syncthreads();
				if(cnt >= need) { /* keep the max_nseed best, in order: rank = number of smaller keys (the keys are distinct) */
					uint32_t rk[CAP / 256], kn[CAP / 256], kp[CAP / 256]; unsigned long long kk[CAP / 256];
#pragma unroll
					for(int q = 0; q < (int)(CAP / 256); ++q) {
						const uint32_t i = q * 256 + tid;
						rk[q] = NONE;
						if(i < cnt) {
							const unsigned long long mine = ck[i];
							uint32_t rank = 0;
							for(uint32_t j = 0; j < cnt; ++j) rank += ck[j] < mine ? 1u : 0u;
							rk[q] = rank; kk[q] = mine; kn[q] = cn[i]; kp[q] = cp[i];
						}
					}
					__syncthreads();
#pragma unroll
					for(int q = 0; q < (int)(CAP / 256); ++q) if(rk[q] < need) { ck[rk[q]] = kk[q]; cn[rk[q]] = kn[q]; cp[rk[q]] = kp[q]; }
					__syncthreads();
					nb = need;
				}
				else nb = cnt;
			}
			return true;
		};
		for(int i = tid; i < (int) HB; i += 256) hist[i] = 0;
		const int nsel0 = choose(Dsel);
		sweep(0, nsel0, NONE, Dsel, 0);
		chunk[tid] = hist[tid * 4] + hist[tid * 4 + 1] + hist[tid * 4 + 2] + hist[tid * 4 + 3];
		__syncthreads();
		if(tid < 64) { /* prefix over the 64 x 16 bins on one wave -> Dk and the number of nodes with d_scan <= Dk */
			const uint32_t c16 = chunk[tid * 4] + chunk[tid * 4 + 1] + chunk[tid * 4 + 2] + chunk[tid * 4 + 3];
			uint32_t inc = c16;
			for(int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if(tid >= off) inc += o; }
			const unsigned long long reached = __ballot(inc >= need);
			if(!reached) { if(tid == 0) sh[1] = NONE; }
			else if(tid == __ffsll((long long) reached) - 1) {
				uint32_t cum = inc - c16; int bb = tid * 16;
				while(bb < tid * 16 + 15 && cum + hist[bb] < need) { cum += hist[bb]; ++bb; }
				sh[1] = (uint32_t) bb; sh[6] = cum + hist[bb];
			}
		}
		__syncthreads();
		const uint32_t Dk = sh[1], cnt1 = sh[6];
		if(Dk <= Dsel && absorb(NONE, Dk, cnt1) && nb == need) {
			const uint32_t c1p = cp[need - 1], d1 = c1p >> 16, N1 = c1p & 0xffffu;     /* the max_nseed-th smallest key of C1 */
			if(N1 != 0) {
				const unsigned long long D1l = (unsigned long long) d1 * L / N1;
				const uint32_t D1 = D1l > 0xfffffffeull ? 0xfffffffeu : (uint32_t) D1l;
				bool ok = true;
				if(D1 > Dk) {
					if(D1 >= DMAX) ok = false;
					else {
						if(tid == 0) { uint32_t e = NONE; if(D1 <= Dsel) { e = 0; for(uint32_t x = Dk + 1; x <= D1; ++x) e += hist[x]; } sh[7] = e; }
						__syncthreads();
						ok = absorb(Dk, D1, sh[7]);
					}
				}
				if(ok) {
					for(uint32_t i = tid; i < need; i += 256) { outId[i] = (int32_t) cn[i]; outDN[i] = cp[i]; outPar[i] = pair_exact(db, R, read, db.parent[cn[i]]); }
					if(tid == 0) { seedCnt[read] = (int32_t) need; if(stat) { atomicAdd(&stat[0], 1u); atomicAdd(&stat[1], (uint32_t) nsel0); atomicAdd(&stat[2], seen); } }
					served = true;
				}
			}
		}
	}
	if(served) return;
	if(tid == 0) retry[1 + atomicAdd(&retry[0], 1)] = read;        /* left to the general launch */
}

/* GENERAL = false (no longer launched: k_seed_topk_straight above took its place): the straight path only (one choice of blocks, no
 * candidate with N = 0 among the max_nseed best); a read it cannot serve is appended to `retry` ([0] = count) and left to a second launch
 * with GENERAL = true (one workgroup per listed read), which
 * widens, retries and finally recomputes: a read that needs the rare paths does not hold up the tail of the launch that serves the rest,
 * and the rare reads of a batch run side by side. */
template<class DT, bool GENERAL>
__global__ __launch_bounds__(256, 4) void k_seed_topk_d(HuDbDev db, const DT* __restrict__ dm, const uint32_t* __restrict__ bminD, HuReadPlanes R,
		int maxNSeed, int32_t* __restrict__ seedCnt, int32_t* __restrict__ seedId, uint32_t* __restrict__ seedDN, uint32_t* __restrict__ parDN,
		uint32_t* __restrict__ stat, int32_t* __restrict__ retry) {
	constexpr uint32_t DMAX = (uint32_t)(DT) ~(DT) 0;
	constexpr int NBITS = 8 * (int) sizeof(DT);
	constexpr uint32_t CAP = 1024, HB = 1024, NONE = 0xffffffffu;
	static_assert(3 * 256 + HU_MAX_SEEDS <= CAP, "three blocks and the kept set fit the candidate buffers");
	__shared__ unsigned long long keys[HU_TOPK_CAP];        /* 32 KB, carved up below */
	__shared__ uint32_t chunk[256];
	__shared__ uint32_t sh[10];
	uint32_t* bm = reinterpret_cast<uint32_t*>(keys);                                   /* [2048] block minima          0 ..  8 KB */
	unsigned short* sel = reinterpret_cast<unsigned short*>(keys + 1024);               /* [2048] chosen blocks         8 .. 12 KB */
	uint32_t* hist = reinterpret_cast<uint32_t*>(keys + 1536);                          /* [HB] histogram of d_scan    12 .. 16 KB */
	unsigned long long* ck = keys + 2048;                                               /* [CAP] candidate keys        16 .. 24 KB */
	uint32_t* cp = reinterpret_cast<uint32_t*>(keys + 3072);                            /* [CAP] candidate (d, N)      24 .. 28 KB */
	uint32_t* cn = reinterpret_cast<uint32_t*>(keys + 3584);                            /* [CAP] candidate nodes       28 .. 32 KB */
	/* GENERAL: a fixed grid walks the list (an empty list costs a thousand workgroups that leave at once, not one per read) */
	for(int item = (int) blockIdx.x; item < (GENERAL ? retry[0] : (int) gridDim.x); item += (int) gridDim.x) {
	if(GENERAL) __syncthreads();           /* the previous read's use of the shared arrays is over */
	const int read = GENERAL ? retry[1 + item] : item, tid = threadIdx.x;
	const size_t np = (size_t) db.nNodesPad;
	const DT* __restrict__ dr = dm + (size_t) read * np;
	const int nBlk = db.nNodesPad / 256;
	int32_t* outId = seedId + (size_t) read * HU_MAX_SEEDS;
	uint32_t* outDN = seedDN + (size_t) read * HU_MAX_SEEDS;
	uint32_t* outPar = parDN + (size_t) read * HU_MAX_SEEDS;
	const uint32_t need = (uint32_t) maxNSeed;
	const int per = (nBlk + 63) / 64;                                                   /* block minima per lane of wave 0: <= 32 */
	for(int i = tid; i < per * 64; i += 256) bm[i] = i < nBlk ? bminD[(size_t) read * nBlk + i] : NONE;
	if(tid == 0) { sh[0] = NONE; sh[2] = 0; sh[3] = 0; sh[9] = 0; sh[5] = read_bases(db, R, read); }
	__syncthreads();
	const uint32_t L = sh[5];
	const uint2 rsp = R.rspan[read];
	if(L == 0) { /* no base in the region (a read that was not aligned): every N is 0, the order is the node ids' */
		if(tid == 0) {
			int k = 0;
			for(int node = 0; node < db.nNodes && k < maxNSeed; ++node) if(node != db.root) { outId[k] = node; outDN[k] = 0; outPar[k] = 0; ++k; }
			seedCnt[read] = k;
		}
		continue;
	}
	/* the want-th smallest block minimum, bit by bit from the top on one wave: the largest r with #{x < r} < want */
	auto select_rank = [&](int want) __attribute__((always_inline)) {
		__syncthreads();
		if(tid < 64) {
			uint32_t x[32];
#pragma unroll
			for(int k = 0; k < 32; ++k) x[k] = k < per ? bm[k * 64 + tid] : NONE;
			uint32_t res = 0;
			if(per <= 16) {
				for(int bit = NBITS; bit >= 0; --bit) {
					const uint32_t trial = res | (1u << bit);
					int cnt = 0;
#pragma unroll
					for(int k = 0; k < 16; ++k) cnt += __popcll(__ballot(x[k] < trial));
					if(cnt < want) res = trial;
				}
			} else {
				for(int bit = NBITS; bit >= 0; --bit) {
					const uint32_t trial = res | (1u << bit);
					int cnt = 0;
#pragma unroll
					for(int k = 0; k < 32; ++k) cnt += __popcll(__ballot(x[k] < trial));
					if(cnt < want) res = trial;
				}
			}
			if(tid == 0) sh[0] = res;
		}
		__syncthreads();
		return sh[0];
	};
	bool served = false;
	{
		/* blocks with minimum <= lim -> sel; returns their number */
		auto choose = [&](uint32_t lim) __attribute__((always_inline)) {
			if(tid == 0) sh[2] = 0;
			__syncthreads();
			for(int i = tid; i < nBlk; i += 256)
				if(bm[i] <= lim) sel[atomicAdd(&sh[2], 1u)] = (unsigned short) i;
			__syncthreads();
			return (int) sh[2];
		};
		/* the nodes of sel[g0, g1) with lo < d_scan <= lim (lo == NONE: no lower limit): histogram (pass 0) or list at sh[3] (pass 1) */
		auto sweep = [&](int g0, int g1, uint32_t lo, uint32_t lim, int pass) __attribute__((always_inline)) {
#pragma unroll 1
			for(int s0 = g0; s0 < g1; s0 += 16) {     /* sixteen loads in flight per thread: scalar block base + one lane offset */
				uint32_t pv[16];
#pragma unroll
				for(int k = 0; k < 16; ++k) pv[k] = s0 + k < g1 ? (uint32_t) (dr + (size_t) __builtin_amdgcn_readfirstlane((int) sel[s0 + k]) * 256)[tid] : NONE;
#pragma unroll
				for(int k = 0; k < 16; ++k) {
					if(s0 + k >= g1) continue;
					const int node = __builtin_amdgcn_readfirstlane((int) sel[s0 + k]) * 256 + tid;
					bool take = node < db.nNodes && node != db.root && pv[k] <= lim;
					if(take && pv[k] == 0) take = hu_cover_meets(db.nodeCover[node], rsp);        /* no shared position: N = 0 exactly, never a candidate (sorts last) */
					if(!take) continue;
					if(pass == 0) atomicAdd(&hist[pv[k]], 1u);
					else if(lo == NONE || pv[k] > lo) { const uint32_t slot = atomicAdd(&sh[3], 1u); if(slot < CAP) cn[slot] = (uint32_t) node; }
				}
			}
			__syncthreads();
		};
		uint32_t nb = 0;                          /* kept candidates: slots [0, nb) of ck / cp / cn, the max_nseed best so far once there are that many */
		uint32_t seen = 0;
		/* candidates with lo < d_scan <= lim join the kept set; `expect` = their number if known */
		auto absorb = [&](uint32_t lo, uint32_t lim, uint32_t expect) __attribute__((always_inline)) -> bool {
			const int nsel = choose(lim);
			const int group = expect != NONE && nb + expect <= CAP ? (nsel > 0 ? nsel : 1) : 3;
			for(int g0 = 0; g0 < nsel; g0 += group) {
				if(tid == 0) sh[3] = nb;
				__syncthreads();
				sweep(g0, min(g0 + group, nsel), lo, lim, 1);
				const uint32_t cnt = sh[3];
				if(cnt > CAP) return false;
				seen += cnt - nb;
				for(uint32_t i = nb + tid; i < cnt; i += 256) {
					const uint32_t c = pair_exact(db, R, read, (int) cn[i]); cp[i] = c; ck[i] = seed_key(c >> 16, c & 0xffffu, cn[i]);
					if((c & 0xffffu) == 0) atomicAdd(&sh[9], 1u);
				}
				__syncthreads();
				if(cnt >= need) { /* keep the max_nseed best, in order: rank = number of smaller keys (the keys are distinct) */
					uint32_t rk[CAP / 256], kn[CAP / 256], kp[CAP / 256]; unsigned long long kk[CAP / 256];
#pragma unroll
					for(int q = 0; q < (int)(CAP / 256); ++q) {
						const uint32_t i = q * 256 + tid;
						rk[q] = NONE;
						if(i < cnt) {
							const unsigned long long mine = ck[i];
							uint32_t rank = 0;
							for(uint32_t j = 0; j < cnt; ++j) rank += ck[j] < mine ? 1u : 0u;
							rk[q] = rank; kk[q] = mine; kn[q] = cn[i]; kp[q] = cp[i];
						}
					}
					__syncthreads();
#pragma unroll
					for(int q = 0; q < (int)(CAP / 256); ++q) if(rk[q] < need) { ck[rk[q]] = kk[q]; cn[rk[q]] = kn[q]; cp[rk[q]] = kp[q]; }
					__syncthreads();
					nb = need;
				}
				else nb = cnt;
			}
			return true;
		};
		/* Dsel: a d_scan that at least max_nseed nodes of the read reach.  The block minima leave out the nodes that meet no read of
		 * the TILE; a node can still miss THIS read, so when the blocks up to the max_nseed-th smallest minimum do not hold max_nseed
		 * nodes that meet it, four times as many blocks are taken, up to all of them. */
#pragma unroll 1
		for(int attempt = 0; attempt < (GENERAL ? 2 : 1) && !served; ++attempt) {      /* a second time over ALL blocks when the first choice of blocks proved too narrow */
		nb = 0;
		__syncthreads();
		if(tid == 0) { sh[3] = 0; sh[9] = 0; }
		uint32_t Dsel = NONE; int nsel0 = 0; bool have = false, usedAll = false;
#pragma unroll 1
		for(uint32_t want = attempt ? (uint32_t) nBlk : need; ; want *= 4) {
			const bool all = want >= (uint32_t) nBlk;
			usedAll = all;
			Dsel = all ? min(DMAX, HB) - 1 : select_rank((int) want);
			if(!(Dsel < DMAX && Dsel < HB)) { if(all) break; continue; }
			__syncthreads();
			for(int i = tid; i < (int) HB; i += 256) hist[i] = 0;
			nsel0 = choose(Dsel);
			sweep(0, nsel0, NONE, Dsel, 0);
			chunk[tid] = hist[tid * 4] + hist[tid * 4 + 1] + hist[tid * 4 + 2] + hist[tid * 4 + 3];
			__syncthreads();
			if(tid < 64) { /* prefix over the 64 x 16 bins on one wave -> Dk and the number of nodes with d_scan <= Dk */
				const uint32_t c16 = chunk[tid * 4] + chunk[tid * 4 + 1] + chunk[tid * 4 + 2] + chunk[tid * 4 + 3];
				uint32_t inc = c16;
				for(int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if(tid >= off) inc += o; }
				const unsigned long long reached = __ballot(inc >= need);
				if(!reached) { if(tid == 0) sh[1] = NONE; }
				else if(tid == __ffsll((long long) reached) - 1) {
					uint32_t cum = inc - c16; int bb = tid * 16;
					while(bb < tid * 16 + 15 && cum + hist[bb] < need) { cum += hist[bb]; ++bb; }
					sh[1] = (uint32_t) bb; sh[6] = cum + hist[bb];
				}
			}
			__syncthreads();
			if(sh[1] != NONE) { have = true; break; }
			if(all || !GENERAL) break;
		}
		const uint32_t Dk = sh[1], cnt1 = sh[6];
		/* ONE call site of absorb (it is a large piece of code), driven by: C1 = {d_scan <= Dk}; then, while the max_nseed-th kept key is
		 * a candidate that shares no valid position with the read (N = 0: such nodes all have d_scan = 0 and pass the interval test only
		 * when their bases lie on both sides of the read; they sort last), the limit moves up to the d_scan that brings as many more
		 * nodes as there were such candidates; last, the nodes up to D1 = floor(d1 L / N1). */
		bool ok = have && Dk <= Dsel, done = false;
		uint32_t lo = NONE, lim = Dk, expect = cnt1, Dcur = Dk, cumCur = cnt1;
		int rounds = 0; bool last = false;
#pragma unroll 1
		while(ok && !done) {
			ok = absorb(lo, lim, expect) && nb == need;
			if(!ok) break;
			Dcur = lim;
			if(last) { done = true; break; }
			const uint32_t c1p = cp[need - 1], d1 = c1p >> 16, N1 = c1p & 0xffffu;     /* the max_nseed-th smallest key so far */
			__syncthreads();
			if(N1 == 0) {
				if(!GENERAL || ++rounds > 4) { ok = false; break; }
				if(tid == 0) {
					const uint32_t target = need + sh[9];        /* sh[9]: candidates with N = 0 met so far */
					uint32_t cum = cumCur, x = Dcur;
					while(x < Dsel && cum < target) { ++x; cum += hist[x]; }
					sh[7] = cum >= target ? x : NONE; sh[8] = cum;
				}
				__syncthreads();
				if(sh[7] == NONE) { ok = false; break; }
				lo = Dcur; lim = sh[7]; expect = sh[8] - cumCur; cumCur = sh[8];
				continue;
			}
			const unsigned long long D1l = (unsigned long long) d1 * L / N1;
			const uint32_t D1 = D1l > 0xfffffffeull ? 0xfffffffeu : (uint32_t) D1l;
			if(D1 <= Dcur) { done = true; break; }
			if(D1 >= DMAX) { ok = false; break; }
			if(tid == 0) { uint32_t e = NONE; if(D1 <= Dsel) { e = 0; for(uint32_t x = Dcur + 1; x <= D1; ++x) e += hist[x]; } sh[7] = e; }
			__syncthreads();
			lo = Dcur; lim = D1; expect = sh[7]; last = true;
		}
		if(ok && done) {
			for(uint32_t i = tid; i < need; i += 256) { outId[i] = (int32_t) cn[i]; outDN[i] = cp[i]; outPar[i] = pair_exact(db, R, read, db.parent[cn[i]]); }
			if(tid == 0) { seedCnt[read] = (int32_t) need; if(stat) { atomicAdd(&stat[GENERAL ? 8 : 0], 1u); atomicAdd(&stat[1], (uint32_t) nsel0); atomicAdd(&stat[2], seen); } }
			served = true;
		}
		if(usedAll) break;
		}
	}
	if(served) continue;
	if(!GENERAL) { if(tid == 0) retry[1 + atomicAdd(&retry[0], 1)] = read; continue; }
	if(tid == 0 && stat) atomicAdd(&stat[3], 1u);
	topk_exact_recompute(db, R, read, maxNSeed, keys, chunk, sh, outId, outDN, outPar, seedCnt);
	}
}

/* ------------------------------------------------------------------------------------------ */
__device__ inline double wave_sum(double x) {
	for(int m = 32; m > 0; m >>= 1) x += __shfl_xor(x, m);
	return x; /* butterfly: bitwise identical in every lane */
}
/* wave64 sum on the DPP network (quad_perm, row_half_mirror, row_mirror, row_bcast:15/31), total read
 * back from lane 63 into SGPRs: ~18 VALU operations instead of 12 ds_bpermute round trips, and the
 * result is wave-uniform by construction (loop control of the EM) */
template<int CTRL, int ROW_MASK>
__device__ inline double dpp_add(double v) {
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
	return v + __hiloint2double(hi, lo);
}
/* all lanes of the source are valid for the permutations within a row: no old value to keep (bound_ctrl) */
template<int CTRL>
__device__ inline double dpp_add_full(double v) {
	const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
	const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
	return v + __hiloint2double(hi, lo);
}
__device__ inline double wave_sum_uniform(double v) {
	v = dpp_add_full<0xB1>(v);   /* quad_perm:[1,0,3,2] */
	v = dpp_add_full<0x4E>(v);   /* quad_perm:[2,3,0,1] */
	v = dpp_add_full<0x141>(v);  /* row_half_mirror */
	v = dpp_add_full<0x140>(v);  /* row_mirror */
	v = dpp_add<0x142, 0xa>(v);  /* row_bcast:15 into rows 1,3 */
	v = dpp_add<0x143, 0xc>(v);  /* row_bcast:31 into rows 2,3 */
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
	return __hiloint2double(hi, lo);
}
__device__ inline double max4d(const double* v) { return fmax(fmax(v[0], v[1]), fmax(v[2], v[3])); }
__device__ inline int argmax4d(const double* v) { /* Eigen maxCoeff(&i): first strict maximum */
	int b = 0;
	if(v[1] > v[b]) b = 1;
	if(v[2] > v[b]) b = 2;
	if(v[3] > v[b]) b = 3;
	return b;
}
/* inferState on a computed message: components that are equal in exact arithmetic (all-gap columns
 * under the equal-frequency models K80/JC69, where P(t) rows are permutations of each other) differ
 * here only by rounding of the spectral products; Eigen's maxCoeff returns the FIRST maximum of an
 * exact tie, so values within 1e-10 (1 + |max|) of the maximum count as tied. */
__device__ inline int argmax4_tied(const double* v) {
	const double mx = max4d(v), tol = 1e-10 * (1.0 + fabs(mx));
	if(v[0] >= mx - tol) return 0;
	if(v[1] >= mx - tol) return 1;
	if(v[2] >= mx - tol) return 2;
	return 3;
}
__device__ inline double sel4(const double* v, int i) { return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3]; }

/* e = exp(M - max M), returns max M (the log scale); all -inf -> zeros */
__device__ inline double lin_msg(const double* M, double* e) {
	double mx = max4d(M);
	if(mx == -INFINITY) { e[0] = e[1] = e[2] = e[3] = 0; return mx; }
#pragma unroll
	for(int i = 0; i < 4; ++i) e[i] = exp(M[i] - mx);
	return mx;
}
/* a = U1 . e  (message in the eigenbasis) */
__device__ inline void to_eig(const HuModelDev& m, const double* e, double* a) {
#pragma unroll
	for(int k = 0; k < 4; ++k) a[k] = (m.U1[k*4+0] * e[0] + m.U1[k*4+1] * e[1]) + (m.U1[k*4+2] * e[2] + m.U1[k*4+3] * e[3]);
}
/* c = P(t) . e = U . (E (.) a), E = exp(lam t); clamped at 0 against spectral cancellation */
__device__ inline void conv_eig(const HuModelDev& m, const double* E, const double* a, double* c) {
	double s[4];
#pragma unroll
	for(int k = 0; k < 4; ++k) s[k] = E[k] * a[k];
#pragma unroll
	for(int i = 0; i < 4; ++i)
		c[i] = fmax((m.U[i*4+0] * s[0] + m.U[i*4+1] * s[1]) + (m.U[i*4+2] * s[2] + m.U[i*4+3] * s[3]), 0.0);
}
/* e = U a (back from the eigenbasis), clamped at 0 */
__device__ inline void from_eig(const HuModelDev& m, const double* a, double* e) {
#pragma unroll
	for(int i = 0; i < 4; ++i) e[i] = fmax((m.U[i*4+0] * a[0] + m.U[i*4+1] * a[1]) + (m.U[i*4+2] * a[2] + m.U[i*4+3] * a[3]), 0.0);
}
__device__ inline void load4(const double* p, double* v) {
	const double2 a = *reinterpret_cast<const double2*>(p), b = *reinterpret_cast<const double2*>(p + 2);
	v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

#define HU_LN2_HI 6.93147180369123816490e-01
#define HU_LN2_LO 1.90821492927058770002e-10
#define HU_LN2 0.693147180559945309417232121458

/* inferState in linear space (argmax of a product == argmax of the sum of logs); exact-arithmetic
 * ties (see argmax4_tied) resolve to the first index */
__device__ inline int argmax4_tied_lin(const double* z) {
	const double thr = max4d(z) * (1.0 - 1e-9);
	if(z[0] >= thr) return 0;
	if(z[1] >= thr) return 1;
	if(z[2] >= thr) return 2;
	return 3;
}

/* one-time packing of the log-space messages of the .ptu into the form of HuDbDev: linear space,
 * scaled by 2^-k (k = rint(max / ln 2), kept aside), and carried into the eigenbasis of the
 * substitution model, a = U^-1 e — every P(t) then acts on a message as a diagonal scaling */
__global__ __launch_bounds__(256) void k_pack_msgs(HuModelDev mdl, double* __restrict__ msg, int32_t* __restrict__ k2, size_t nsites) {
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if(i >= nsites) return;
	double M[4];
	load4(msg + i * 4, M);
	const double mx = max4d(M);
	int k = 0;
	double e[4] = {0, 0, 0, 0}, a[4];
	if(mx != -INFINITY) {
		k = (int) rint(mx * (1.0 / HU_LN2));
		for(int c = 0; c < 4; ++c) e[c] = exp(fma(-(double) k, HU_LN2_LO, fma(-(double) k, HU_LN2_HI, M[c])));
	}
	to_eig(mdl, e, a);
	*reinterpret_cast<double2*>(msg + i * 4) = make_double2(a[0], a[1]);
	*reinterpret_cast<double2*>(msg + i * 4 + 2) = make_double2(a[2], a[3]);
	k2[i] = k;
}

struct HuEstOut { double ratio, wnr, loglik; };

/* one wave per (read, seed); lanes stride the sites of [start, end].  UNR sites per loop trip: the
 * loads of all of them are issued before the first is consumed (the kernel is bound by memory
 * latency x occupancy, not by bandwidth: measured by capping waves per CU). */
template<int UNR>
__device__ inline void estimate_body(const HuDbDev& db, const HuModelDev& mdl, const int8_t* __restrict__ codes,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, const uint32_t* __restrict__ parDN,
		const int32_t* __restrict__ seedCnt, const int32_t* __restrict__ seedId, const uint32_t* __restrict__ seedDN,
		int weighted, HuEstOut* __restrict__ out) {
	const uint32_t slot = db.wideList ? db.wideList[blockIdx.x] : blockIdx.x;
	const int read = slot / HU_MAX_SEEDS, s = slot % HU_MAX_SEEDS, lane = threadIdx.x;
	if(s >= seedCnt[read] || hu_skip_width(db, rend[read] - rstart[read] + 1)) return;
	const int u = seedId[(size_t) read * HU_MAX_SEEDS + s];
	const int v = db.parent[u];
	const uint32_t dn = seedDN[(size_t) read * HU_MAX_SEEDS + s];
	const uint32_t pv = parDN[(size_t) read * HU_MAX_SEEDS + s];       /* (d, N) against the parent's sequence */
	const double cDist = (double)(dn >> 16) / (double)(dn & 0xffffu);
	const double pDist = (double)(pv >> 16) / (double)(pv & 0xffffu);
	double ratio = cDist / (cDist + pDist);
	if(isnan(ratio)) ratio = 0.5;
	const double w0 = db.blen[u];
	const double wur = w0 * ratio, wvr = w0 - wur;
	double Eu[4], Ev[4];
#pragma unroll
	for(int k = 0; k < 4; ++k) { Eu[k] = exp(mdl.lam[k] * wur); Ev[k] = exp(mdl.lam[k] * wvr); }
	const int start = rstart[read], end = rend[read];
	const int8_t* __restrict__ cd = codes + (size_t) read * db.csLen;
	const int64_t sOff = (int64_t) u * db.winLen - db.winStart; /* + j, j >= winStart */
	const int piMax = argmax4d(mdl.logpi);
	double piw[4]; /* inferWeight(log pi) */
	{ double mx = max4d(mdl.logpi), sm; for(int i = 0; i < 4; ++i) piw[i] = exp(mdl.logpi[i] - mx); sm = (piw[0] + piw[2]) + (piw[1] + piw[3]); for(int i = 0; i < 4; ++i) piw[i] /= sm; }

	/* z_i = (P(wur) e^U)_i (P(wvr) e^V)_i in linear space; log R_j = log z + (kU + kV) ln 2 */
	auto zOf = [&](const double* aU, const double* aV, double* z) { /* messages arrive in the eigenbasis */
		double c[4];
		conv_eig(mdl, Eu, aU, z);
		conv_eig(mdl, Ev, aV, c);
		for(int i = 0; i < 4; ++i) z[i] *= c[i];
	};
	double dsum = 0, nsum = 0;
	for(int j0 = start + lane; j0 <= end; j0 += 64 * UNR) {
		double eU[UNR][4], eV[UNR][4]; int bb[UNR];
#pragma unroll
		for(int t = 0; t < UNR; ++t) {
			const int j = j0 + 64 * t <= end ? j0 + 64 * t : j0;
			load4(db.up + (sOff + j) * 4, eU[t]); load4(db.down + (sOff + j) * 4, eV[t]); bb[t] = cd[j];
		}
#pragma unroll
		for(int t = 0; t < UNR; ++t) {
			if(j0 + 64 * t > end) continue;
			double z[4];
			zOf(eU[t], eV[t], z);
			const int b = bb[t];
			const int b1 = argmax4_tied_lin(z), b2 = b >= 0 ? b : piMax;
			if(!weighted) { if(b1 != b2) dsum += 1; }
			else {
				double w1 = sel4(z, b1) / ((z[0] + z[2]) + (z[1] + z[3]));
				double w2 = b >= 0 ? 1.0 : piw[b2];
				if(b1 != b2) dsum += w1 * w2;
				nsum += w1 * w2;
			}
		}
	}
	dsum = wave_sum(dsum);
	double wnr;
	if(!weighted) wnr = dsum / (double)(end - start + 1);
	else { nsum = wave_sum(nsum); wnr = dsum / nsum; }
	/* N*P(wnr): column b of P(wnr) for a base, P(wnr).pi for a gap */
	double En[4], Ppi[4];
#pragma unroll
	for(int k = 0; k < 4; ++k) En[k] = exp(mdl.lam[k] * wnr);
	{ double a[4]; to_eig(mdl, mdl.pi, a); if(wnr == 0) { for(int i = 0; i < 4; ++i) Ppi[i] = mdl.pi[i]; } else conv_eig(mdl, En, a, Ppi); }
	double ll = 0;
	long long ksum = 0;
	for(int j0 = start + lane; j0 <= end; j0 += 64 * UNR) {
		double eU[UNR][4], eV[UNR][4]; int bb[UNR]; int kk[UNR];
#pragma unroll
		for(int t = 0; t < UNR; ++t) {
			const int j = j0 + 64 * t <= end ? j0 + 64 * t : j0;
			load4(db.up + (sOff + j) * 4, eU[t]); load4(db.down + (sOff + j) * 4, eV[t]); bb[t] = cd[j];
			kk[t] = db.upK[sOff + j] + db.downK[sOff + j];
		}
#pragma unroll
		for(int t = 0; t < UNR; ++t) {
			if(j0 + 64 * t > end) continue;
			double z[4], c[4];
			zOf(eU[t], eV[t], z);
			const int b = bb[t];
			if(b >= 0) {
				if(wnr == 0) { for(int i = 0; i < 4; ++i) c[i] = i == b ? 1.0 : 0.0; }
				else { double a[4]; for(int k = 0; k < 4; ++k) a[k] = mdl.U1[k*4+b]; conv_eig(mdl, En, a, c); }
			}
			else for(int i = 0; i < 4; ++i) c[i] = Ppi[i];
			ll += log((mdl.pi[0] * z[0] * c[0] + mdl.pi[2] * z[2] * c[2]) + (mdl.pi[1] * z[1] * c[1] + mdl.pi[3] * z[3] * c[3]));
			ksum += (long long) kk[t];
		}
	}
	ll = wave_sum(ll);
	for(int m = 32; m > 0; m >>= 1) ksum += __shfl_xor(ksum, m);
	ll += (double) ksum * HU_LN2;
	if(lane == 0) { HuEstOut o; o.ratio = ratio; o.wnr = wnr; o.loglik = ll; out[(size_t) read * HU_MAX_SEEDS + s] = o; }
}

#define HU_EST_KERNEL(NAME, UNR, MINW) \
__global__ __launch_bounds__(64, MINW) void NAME(HuDbDev db, HuModelDev mdl, const int8_t* __restrict__ codes, \
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, const uint32_t* __restrict__ parDN, \
		const int32_t* __restrict__ seedCnt, const int32_t* __restrict__ seedId, const uint32_t* __restrict__ seedDN, \
		int weighted, HuEstOut* __restrict__ out) { \
	estimate_body<UNR>(db, mdl, codes, rstart, rend, parDN, seedCnt, seedId, seedDN, weighted, out); \
}
/* measured on MI355X (8192 reads, R = 1363): UNR 1 -> 12.3 ms, 2 -> 12.4, 4 -> 13.3 (register pressure
 * costs a wave per SIMD); forcing 5-6 waves per SIMD spills and doubles the time.  At UNR 1 the kernel
 * moves 76.6 GB per launch (PMC) = 6.2 TB/s, the achievable HBM rate. */
HU_EST_KERNEL(k_estimate, 1, 1)

/* ------------------------------------------------------------------------------------------ */
struct HuCand { int32_t read, node; double ratio0, wnr0; };
struct HuPlaceOut { double wnr, wur; int32_t iters, pad; };

/* k_root_loglik (hu_opts.fix_root_loglik): the log-likelihood the reference evidently meant placeSeq to return — the three
 * messages of u, v and the read meeting at the new root r at the optimised branch lengths, sum_j log pi . exp(loglik(r, j)),
 * loglik(r, j) with the dGamma averaging of src/PhyloTreeUnrooted.cpp:320-346 — instead of the constant it does return because the
 * root message is re-initialised first (:918-922).  One wave per candidate, the sites strided over the lanes; per
 * site and rate category two matvecs in the eigenbasis and a table look-up for the read's base (tab[k][b][i] = (P(w_nr r_k) c^b)_i,
 * c^b = e_b or pi), one log per site.  Off the default path. */
__global__ __launch_bounds__(64) void k_root_loglik(HuDbDev db, HuModelDev mdl, const int8_t* __restrict__ codes,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, const HuCand* __restrict__ cands,
		const HuPlaceOut* __restrict__ po, int nc, double* __restrict__ out) {
	__shared__ double Eu[HU_MAX_DGK][4], Ev[HU_MAX_DGK][4], tab[HU_MAX_DGK][5][4];
	const int c = blockIdx.x, lane = threadIdx.x;
	if(c >= nc) return;
	const HuCand cd = cands[c];
	const int read = cd.read, un = cd.node;
	const int start = rstart[read], end = rend[read];
	const int Kc = mdl.dgK > 0 ? mdl.dgK : 1;
	const double w0 = db.blen[un], wur = po[c].wur, wvr = w0 - wur, wnr = po[c].wnr;
	for(int e = lane; e < Kc * 4; e += 64) {
		const int k = e >> 2, m = e & 3;
		Eu[k][m] = exp(mdl.lam[m] * (wur * mdl.rate[k])); Ev[k][m] = exp(mdl.lam[m] * (wvr * mdl.rate[k]));
	}
	for(int e = lane; e < Kc * 20; e += 64) { /* (P(t) c^b)_i = sum_m U_im exp(lam_m t) (U^-1 c^b)_m */
		const int k = e / 20, b = (e % 20) >> 2, i = e & 3;
		const double t = wnr * mdl.rate[k];
		double v = 0;
		if(t == 0) v = b < 4 ? (i == b ? 1.0 : 0.0) : mdl.pi[i];
		else {
			for(int m = 0; m < 4; ++m) {
				double am = 0;
				if(b < 4) am = mdl.U1[m * 4 + b];
				else for(int x = 0; x < 4; ++x) am += mdl.U1[m * 4 + x] * mdl.pi[x];
				v += mdl.U[i * 4 + m] * exp(mdl.lam[m] * t) * am;
			}
			v = fmax(v, 0.0);
		}
		tab[k][b][i] = v;
	}
	__syncthreads();
	const int8_t* __restrict__ cdr = codes + (size_t) read * db.csLen;
	const int64_t sOff = (int64_t) un * db.winLen - db.winStart;
	const double rK = 1.0 / (double) Kc;
	double ll = 0; long long ksum = 0;
	for(int j = start + lane; j <= end; j += 64) {
		double aU[4], aV[4];
		load4(db.up + (sOff + j) * 4, aU); load4(db.down + (sOff + j) * 4, aV);
		const int b = cdr[j] >= 0 ? cdr[j] : 4;
		double lik = 0;
		for(int k = 0; k < Kc; ++k) {
			double cu[4], cv[4];
			conv_eig(mdl, Eu[k], aU, cu); conv_eig(mdl, Ev[k], aV, cv);
			const double* cn = tab[k][b];
			lik += (mdl.pi[0] * cu[0] * cv[0] * cn[0] + mdl.pi[2] * cu[2] * cv[2] * cn[2]) + (mdl.pi[1] * cu[1] * cv[1] * cn[1] + mdl.pi[3] * cu[3] * cv[3] * cn[3]);
		}
		ll += log(lik * rK);
		ksum += (long long) db.upK[sOff + j] + (long long) db.downK[sOff + j];
	}
	ll = wave_sum(ll);
	for(int m = 32; m > 0; m >>= 1) ksum += __shfl_xor(ksum, m);
	if(lane == 0) out[c] = ll + (double) ksum * HU_LN2;
}

/* Felsenstein's EM for one branch (src/PhyloTreeUnrooted.cpp:749-798) on the per-site ratios
 * rho_j = A_j / B_j kept in LDS: p <- mean_j p0 / (rho_j q0 + p0); NaN sites are skipped (their count
 * does not change between iterations and is taken once).  The quotient is a reciprocal refined by two
 * Newton steps (full double precision for finite non-zero denominators, IEEE division otherwise). */
__device__ inline double em_branch(const double* rho, int n, double w0, double maxL, int lane, int& emIters) {
	double q0 = exp(-w0), p0 = 1 - q0, p = p0, q = q0;
	double c = 0;
	for(int j = lane; j < n; j += 64) c += isnan(rho[j]) ? 0.0 : 1.0;
	c = wave_sum_uniform(c);
	for(int it = 0; it < HU_MAX_ITER && p >= 0 && p <= 1; ++it) {
		double s = 0;
		for(int j = lane; j < n; j += 64) {
			const double r = rho[j];
			const double x = fma(r, q0, p0);
			double t;
			if(x > 1e-300 && x < 1e300) {
				double y = __builtin_amdgcn_rcp(x);
				y = fma(y, fma(-x, y, 1.0), y);
				y = fma(y, fma(-x, y, 1.0), y);
				t = p0 * y;
			}
			else t = p0 / x;
			s += isnan(r) ? 0.0 : t;
		}
		s = wave_sum_uniform(s);
		p = s / c; q = 1 - p;
		++emIters;
		if(fabs(log(q) - log(q0)) < HU_BRANCH_EPS) break;
		p0 = p; q0 = q;
	}
	double w = -log(q);
	if(w > maxL) w = maxL;
	return w;
}

/* per-site bodies of the two sweeps of one outer iteration (see k_place).  With messages in the
 * eigenbasis, sum_k (P_k^u e^U)_i (P_k^v e^V)_i = sum_mn U_im U_in aU_m aV_n G_mn with the wave-uniform
 * G_mn = sum_k exp(lam_m w_u r_k) exp(lam_n w_v r_k): the rate categories cost nothing per site. */
struct PlaceCtx {
	const HuModelDev* mdl; const double* G; const double* Tb; double pi2;
};
/* X_i = sum_n U_in (sum_m U_im H_mn) */
__device__ inline void bilinear_U(const HuModelDev& m, const double* H, double* X) {
#pragma unroll
	for(int i = 0; i < 4; ++i) {
		double y[4];
#pragma unroll
		for(int n = 0; n < 4; ++n)
			y[n] = (m.U[i*4+0] * H[0*4+n] + m.U[i*4+1] * H[1*4+n]) + (m.U[i*4+2] * H[2*4+n] + m.U[i*4+3] * H[3*4+n]);
		X[i] = fmax((m.U[i*4+0] * y[0] + m.U[i*4+1] * y[1]) + (m.U[i*4+2] * y[2] + m.U[i*4+3] * y[3]), 0.0);
	}
}
__device__ inline double place_site_rn(const PlaceCtx& c, const double* aU, const double* aV, int b) {
	const HuModelDev& mdl = *c.mdl;
	double H[16], X[4];
#pragma unroll
	for(int m = 0; m < 4; ++m)
#pragma unroll
		for(int n = 0; n < 4; ++n) H[m*4+n] = (aU[m] * aV[n]) * c.G[m*4+n];
	bilinear_U(mdl, H, X);
	const double piX = (mdl.pi[0] * X[0] + mdl.pi[2] * X[2]) + (mdl.pi[1] * X[1] + mdl.pi[3] * X[3]);
	if(b >= 0) return sel4(X, b) / piX;
	return ((mdl.pi[0] * mdl.pi[0] * X[0] + mdl.pi[2] * mdl.pi[2] * X[2]) + (mdl.pi[1] * mdl.pi[1] * X[1] + mdl.pi[3] * mdl.pi[3] * X[3])) / (piX * c.pi2);
}
__device__ inline double place_site_ru(const PlaceCtx& c, const double* aU, const double* aV, int b) {
	const HuModelDev& mdl = *c.mdl;
	const double* T = c.Tb + (b >= 0 ? b : 4) * 16;
	double H[16], X[4], eU[4];
#pragma unroll
	for(int m = 0; m < 4; ++m)
#pragma unroll
		for(int n = 0; n < 4; ++n) H[m*4+n] = aV[m] * T[m*4+n];
	bilinear_U(mdl, H, X);
	from_eig(mdl, aU, eU);
	const double piX = (mdl.pi[0] * X[0] + mdl.pi[2] * X[2]) + (mdl.pi[1] * X[1] + mdl.pi[3] * X[3]);
	const double piU = (mdl.pi[0] * eU[0] + mdl.pi[2] * eU[2]) + (mdl.pi[1] * eU[1] + mdl.pi[3] * eU[3]);
	const double A = (mdl.pi[0] * X[0] * eU[0] + mdl.pi[2] * X[2] * eU[2]) + (mdl.pi[1] * X[1] * eU[1] + mdl.pi[3] * X[3] * eU[3]);
	return A / (piX * piU);
}

/* One wave per candidate placement.  Messages stay in linear space (packed at load time); P(t r_k)
 * acts in the eigenbasis so a category costs 4 mul + 16 fma per message.  The r->v message of the
 * reference's outer iteration is never read by anything and is not evaluated.
 * LDS: rho[n] + E tables + leaf-conv table.  PAIR: two sites per loop trip (loads of both issued
 * before either is consumed). */
template<bool PAIR>
__device__ inline void place_body(const HuDbDev& db, const HuModelDev& mdl, const int8_t* __restrict__ codes,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend,
		const HuCand* __restrict__ cands, HuPlaceOut* __restrict__ out, double* lds) {
	const int lane = threadIdx.x;
	const uint32_t ci = db.wideList ? db.wideList[blockIdx.x] : blockIdx.x;
	const HuCand cd = cands[ci];
	const int read = cd.read, u = cd.node;
	const int start = rstart[read], end = rend[read], n = end - start + 1;
	if(hu_skip_width(db, n)) return;
	const int Kc = mdl.dgK > 0 ? mdl.dgK : 1;
	double* Gtab = lds;                    /* [16]    G_mn for the (u, v) pair of branches                 */
	double* Ttab = Gtab + 16;              /* [5][16] T^b_mn for the (v, n) pair and each leaf vector      */
	double* ctab = Ttab + 80;              /* lam[4] rate[16] cb[5][4]: model constants that the table builders
	                                        * index by lane (dynamic indexing of kernel arguments would push the
	                                        * whole model block into VGPRs/scratch) */
	double* rho = lds + 3 * HU_MAX_DGK * 4 + HU_MAX_DGK * 5 * 4; /* (offset kept: the host sizes LDS by it) */
	const int8_t* __restrict__ cdr = codes + (size_t) read * db.csLen + start;
	const int64_t mOff = ((int64_t) u * db.winLen + (start - db.winStart)) * 4;
	const double* __restrict__ Ub = db.up + mOff;
	const double* __restrict__ Vb = db.down + mOff;
	const double w0 = db.blen[u];
	double lenUR = w0 * cd.ratio0, lenVR = w0 * (1 - cd.ratio0), lenNR = cd.wnr0;
	double wur0 = lenUR, wnr0 = lenNR;
	const double w0j = lenUR + lenVR;
	double wur = wur0, wnr = wnr0;
	double pi2 = 0;
	for(int i = 0; i < 4; ++i) pi2 += mdl.pi[i] * mdl.pi[i];
	double api[4];
	to_eig(mdl, mdl.pi, api);
	if(lane == 0) {
#pragma unroll
		for(int i = 0; i < 4; ++i) ctab[i] = mdl.lam[i];
#pragma unroll
		for(int i = 0; i < HU_MAX_DGK; ++i) ctab[4 + i] = mdl.rate[i];
#pragma unroll
		for(int b = 0; b < 4; ++b)
#pragma unroll
			for(int n = 0; n < 4; ++n) ctab[20 + b * 4 + n] = mdl.U1[n*4+b];
#pragma unroll
		for(int n = 0; n < 4; ++n) ctab[20 + 16 + n] = api[n];
	}
	__syncthreads();
	const double* clam = ctab; const double* crate = ctab + 4; const double* ccb = ctab + 20;
	int iter = 0, emIters = 0;
	for(; iter < HU_MAX_ITER && 0 <= wur && wur <= w0j; ++iter) {
		/* G_mn = sum_k exp(lam_m w_ur r_k) exp(lam_n w_vr r_k) / Kc */
		if(lane < 16) {
			const int m = lane >> 2, n = lane & 3;
			double g = 0;
			for(int k = 0; k < Kc; ++k) g += exp(clam[m] * (lenUR * crate[k])) * exp(clam[n] * (lenVR * crate[k]));
			Gtab[lane] = g / Kc;
		}
		__syncthreads();
		PlaceCtx pc = { &mdl, Gtab, Ttab, pi2 };
		/* (i) message r->n from children u, v; EM on the n-r branch against the read's leaf message */
		if(PAIR) {
			for(int j = lane; j < n; j += 128) {
				const int j2 = j + 64; const bool two = j2 < n; const int jj = two ? j2 : j;
				double eU[4], eV[4], fU[4], fV[4];
				load4(Ub + (size_t) j * 4, eU); load4(Vb + (size_t) j * 4, eV);
				load4(Ub + (size_t) jj * 4, fU); load4(Vb + (size_t) jj * 4, fV);
				const int b = cdr[j], b2 = cdr[jj];
				rho[j] = place_site_rn(pc, eU, eV, b);
				if(two) rho[j2] = place_site_rn(pc, fU, fV, b2);
			}
		}
		else for(int j = lane; j < n; j += 64) {
			double eU[4], eV[4];
			load4(Ub + (size_t) j * 4, eU); load4(Vb + (size_t) j * 4, eV);
			rho[j] = place_site_rn(pc, eU, eV, cdr[j]);
		}
		__syncthreads();
		wnr = em_branch(rho, n, lenNR, 1.0, lane, emIters);
		lenNR = wnr;
		__syncthreads();
		/* T^b_mn = c^b_n sum_k exp(lam_m w_vr r_k) exp(lam_n w_nr r_k) / Kc, c^b = U^-1 e_b (b < 4) or U^-1 pi (gap) */
		for(int i = lane; i < 80; i += 64) {
			const int b = i >> 4, m = (i >> 2) & 3, n = i & 3;
			double g = 0;
			for(int k = 0; k < Kc; ++k) g += exp(clam[m] * (lenVR * crate[k])) * exp(clam[n] * (lenNR * crate[k]));
			Ttab[i] = ccb[b * 4 + n] * (g / Kc);
		}
		__syncthreads();
		/* (ii) message r->u from children v, n; EM on the u-r branch against u's own message */
		if(PAIR) {
			for(int j = lane; j < n; j += 128) {
				const int j2 = j + 64; const bool two = j2 < n; const int jj = two ? j2 : j;
				double eU[4], eV[4], fU[4], fV[4];
				load4(Ub + (size_t) j * 4, eU); load4(Vb + (size_t) j * 4, eV);
				load4(Ub + (size_t) jj * 4, fU); load4(Vb + (size_t) jj * 4, fV);
				const int b = cdr[j], b2 = cdr[jj];
				rho[j] = place_site_ru(pc, eU, eV, b);
				if(two) rho[j2] = place_site_ru(pc, fU, fV, b2);
			}
		}
		else for(int j = lane; j < n; j += 64) {
			double eU[4], eV[4];
			load4(Ub + (size_t) j * 4, eU); load4(Vb + (size_t) j * 4, eV);
			rho[j] = place_site_ru(pc, eU, eV, cdr[j]);
		}
		__syncthreads();
		wur = em_branch(rho, n, lenUR, w0j, lane, emIters);
		lenUR = wur;
		lenVR = w0j - wur;
		__syncthreads();
		if(fabs(wur - wur0) < HU_BRANCH_EPS && fabs(wnr - wnr0) < HU_BRANCH_EPS) { ++iter; break; }
		wur0 = wur; wnr0 = wnr;
	}
	if(lane == 0) { HuPlaceOut o; o.wnr = lenNR; o.wur = lenUR; o.iters = iter; o.pad = emIters; out[ci] = o; }
}

#define HU_PLACE_KERNEL(NAME, PAIR, MINW) \
__global__ __launch_bounds__(64, MINW) void NAME(HuDbDev db, HuModelDev mdl, const int8_t* __restrict__ codes, \
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, const HuCand* __restrict__ cands, HuPlaceOut* __restrict__ out) { \
	extern __shared__ double lds[]; \
	place_body<PAIR>(db, mdl, codes, rstart, rend, cands, out, lds); \
}
/* measured (8192 reads, R = 1363, mean 25.6 candidates): per-category convolutions 26.3 ms; the G/T-table
 * form 23.2 ms with one site per trip (3 waves per SIMD) and 23.9 ms with two (2 waves per SIMD) */
HU_PLACE_KERNEL(k_place, false, 1)
