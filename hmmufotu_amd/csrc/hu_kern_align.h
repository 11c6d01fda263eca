// HIP kernels of the banded-HMM alignment stage (gfx950, wave64).  Included by hu_engine.hip.
//
//   k_viterbi      prepareViterbiScores + calcViterbiScores (banded or full) + S + minCoeff +
//                  buildViterbiTrace   (src/BandedHMMP7.cpp:735-892, 943-1006)
//   k_align_rows   buildGlobalAlign + getPaddingSeq (src/BandedHMMP7.cpp:1008-1081, 1137-1186)
//   k_merge_rows   PE orientation check + HmmAlignment::merge (src/hmmufotu.cpp:629-639,
//                  src/BandedHMMP7.cpp:1188-1213)
//   k_encode_rows  DigitalSeq(abc, id, aln.align) (src/DigitalSeq.cpp:41-48) + bit-planes of the
//                  aligned read restricted to [csStart-1, csEnd-1] for the seed scan
//
// The DP is evaluated phase by phase exactly as the reference orders it (upstream rectangle,
// seed band, ..., downstream rectangle without the B-entry term), each phase as an
// anti-diagonal wavefront: M(i,j) needs (i-1,j-1), I(i,j) needs (i-1,j), D(i,j) needs (i,j-1),
// so every cell of one anti-diagonal is independent and every value is produced by the same
// additions and minima as the reference's j-major/i-minor loop => bit-identical costs.
// All three matrices of every computed cell are kept (per-read scratch in HBM) because the
// reference's traceback re-evaluates the candidate sums on the FINAL matrices.
#pragma once
#include "hu_common.h"

struct HuVitOut { int32_t alnStart, alnEnd, alnFrom, alnTo, traceLen, status; double minScore; };

__device__ inline bool reg_contains(const HuRegion& g, int i, int j) {
	if(j < g.j0 || j > g.j1 || i < g.i0 || i > g.i1) return false;
	if(g.band) { int dist = (i - g.from) - (j - g.start); if(!(dist <= g.nIns && dist >= -g.nDel)) return false; }
	return true;
}

struct VitCtx {
	const HuReadDesc* rd;
	const double* scr;   /* the read's scratch: 3 doubles per cell */
	double tNN, tNB;
};

/* value of (M, I, D)(i, j) as the reference's dense matrices would hold it, looking at the
 * first `upto` phases (later phases overwrite earlier ones) */
__device__ inline void vit_lookup(const VitCtx& c, int upto, int i, int j, double& m, double& ii, double& d) {
	const HuReadDesc& rd = *c.rd;
	for(int r = upto - 1; r >= 0; --r) {
		const HuRegion& g = rd.reg[r];
		if(reg_contains(g, i, j)) {
			const int64_t idx = (g.off + (int64_t)(j - g.j0) * (g.i1 - g.i0 + 1) + (i - g.i0)) * 3;
			m = c.scr[idx]; ii = c.scr[idx + 1]; d = c.scr[idx + 2];
			return;
		}
	}
	if(j == 0 && i >= 1) { /* B state column (src/BandedHMMP7.cpp:735-746) */
		double v = (i == 1) ? 0.0 : __dmul_rn(c.tNN, (double)(i - 1));
		v = __dadd_rn(v, c.tNB);
		m = v; ii = v; d = INFINITY;
		return;
	}
	m = ii = d = INFINITY;
}
__device__ inline double vit_bcol(const VitCtx& c, int i) {
	double v = (i == 1) ? 0.0 : __dmul_rn(c.tNN, (double)(i - 1));
	return __dadd_rn(v, c.tNB);
}

__constant__ int8_t c_sym_map[128];

__global__ __launch_bounds__(64) void k_viterbi(HuDbDev db, const HuReadDesc* __restrict__ descs, const char* __restrict__ bases,
		double* __restrict__ scratch, char* __restrict__ traces, double tNN, double tNB, double tEC, double tCC,
		HuVitOut* __restrict__ outs) {
	const int s = blockIdx.x, lane = threadIdx.x;
	const HuReadDesc& rd = descs[s];
	const int L = rd.len, K = db.K;
	if(rd.nRegions <= 0) { if(lane == 0) { HuVitOut o = {0, 0, 0, 0, 0, HU_READ_INVALID, INFINITY}; outs[s] = o; } return; }
	const char* __restrict__ x = bases + rd.baseOff;
	double* scr = scratch + rd.scratchOff * 3;
	VitCtx ctx = { &rd, scr, tNN, tNB };
	for(int r = 0; r < rd.nRegions; ++r) {
		const HuRegion g = rd.reg[r];
		const int ni = g.i1 - g.i0 + 1, nj = g.j1 - g.j0 + 1;
		if(ni <= 0 || nj <= 0) continue;
		for(int dg = 0; dg <= ni + nj - 2; ++dg) {
			const int lo = dg - (nj - 1) > 0 ? dg - (nj - 1) : 0, hi = dg < ni - 1 ? dg : ni - 1;
			for(int q = lo + lane; q <= hi; q += 64) {
				const int i = g.i0 + q, j = g.j0 + dg - q;
				if(g.band) { int dist = (i - g.from) - (j - g.start); if(!(dist <= g.nIns && dist >= -g.nDel)) continue; }
				const int b = c_sym_map[(int) x[i - 1] & 127];
				double mD, iD, dD, mU, iU, dU, mL, iL, dL;
				vit_lookup(ctx, r + 1, i - 1, j - 1, mD, iD, dD);
				vit_lookup(ctx, r + 1, i - 1, j, mU, iU, dU);
				vit_lookup(ctx, r + 1, i, j - 1, mL, iL, dL);
				const double* tp = db.T + (size_t)(j - 1) * 8;
				const double* tj = db.T + (size_t) j * 8;
				double best = fmin(mD + tp[0], fmin(iD + tp[3], dD + tp[5]));
				if(g.withB) best = fmin(vit_bcol(ctx, i) + db.entryC[j], best);
				const double M = db.EM[(size_t) j * 4 + b] + best;
				const double I = db.EI[(size_t) j * 4 + b] + fmin(mU + tj[1], iU + tj[4]);
				/* D1 and DK are retracted: no phase ever assigns them, they stay +inf */
				const double D = (j > 1 && j < K) ? fmin(mL + tp[2], dL + tp[6]) : INFINITY;
				const int64_t idx = (g.off + (int64_t)(j - g.j0) * ni + q) * 3;
				scr[idx] = M; scr[idx + 1] = I; scr[idx + 2] = D;
			}
			__syncthreads();
		}
	}
	/* S = M + exit (+ I(.,K) + t_K(I,M) in the extra column) + E->C + C->C loops; minCoeff with
	 * Eigen's column-major first-minimum rule; a cell counts in the LAST phase that covers it */
	double bestS = INFINITY; int bestCol = 0x7fffffff, bestRow = 0x7fffffff;
	for(int r = 0; r < rd.nRegions; ++r) {
		const HuRegion g = rd.reg[r];
		const int ni = g.i1 - g.i0 + 1, nj = g.j1 - g.j0 + 1;
		if(ni <= 0 || nj <= 0) continue;
		const int64_t ncell = (int64_t) ni * nj;
		for(int64_t cidx = lane; cidx < ncell; cidx += 64) {
			const int i = g.i0 + (int)(cidx % ni), j = g.j0 + (int)(cidx / ni);
			if(!reg_contains(g, i, j)) continue;
			bool later = false;
			for(int r2 = r + 1; r2 < rd.nRegions; ++r2) if(reg_contains(rd.reg[r2], i, j)) later = true;
			if(later) continue;
			const int64_t idx = (g.off + cidx) * 3;
			const double cc = (i < L) ? __dmul_rn(tCC, (double)(L - i)) : 0.0;
			double sv = __dadd_rn(__dadd_rn(scr[idx], db.exitC[j]), tEC);
			if(i < L) sv = __dadd_rn(sv, cc);
			if(sv < bestS || (sv == bestS && (j < bestCol || (j == bestCol && i < bestRow)))) { bestS = sv; bestCol = j; bestRow = i; }
			if(j == K) {
				double s2 = __dadd_rn(__dadd_rn(scr[idx + 1], db.T[(size_t) K * 8 + 3]), tEC);
				if(i < L) s2 = __dadd_rn(s2, cc);
				if(s2 < bestS || (s2 == bestS && (K + 1 < bestCol || (K + 1 == bestCol && i < bestRow)))) { bestS = s2; bestCol = K + 1; bestRow = i; }
			}
		}
	}
	for(int m = 32; m > 0; m >>= 1) {
		const double os = __shfl_xor(bestS, m); const int oc = __shfl_xor(bestCol, m), orow = __shfl_xor(bestRow, m);
		if(os < bestS || (os == bestS && (oc < bestCol || (oc == bestCol && orow < bestRow)))) { bestS = os; bestCol = oc; bestRow = orow; }
	}
	if(lane != 0) return;
	HuVitOut o;
	o.minScore = bestS; o.traceLen = 0; o.alnStart = o.alnEnd = o.alnFrom = o.alnTo = 0;
	if(!(bestS < INFINITY)) { o.status = HU_READ_NEEDS_FULL; outs[s] = o; return; }
	/* buildViterbiTrace (src/BandedHMMP7.cpp:943-1006) */
	char* tr = traces + rd.traceOff;
	int n = 0;
	char st = bestCol <= K ? 'M' : 'I';
	int i = bestRow, j = bestCol <= K ? bestCol : K;
	o.alnEnd = j; o.alnTo = bestRow;
	const int R = rd.nRegions;
	tr[n++] = 'E';
	while(i >= 1 && j >= 0) {
		tr[n++] = st;
		if(st == 'M') {
			double mD, iD, dD;
			vit_lookup(ctx, R, i - 1, j - 1, mD, iD, dD);
			const double* tp = db.T + (size_t)(j - 1) * 8;
			const double pB = vit_bcol(ctx, i) + db.entryC[j];
			double mn = INFINITY; char nx = 'B';
			if(j > 1) {
				const double pM = mD + tp[0], pI = iD + tp[3], pD = dD + tp[5];
				if(pB < mn) { nx = 'B'; mn = pB; }
				if(pM < mn) { nx = 'M'; mn = pM; }
				if(pI < mn) { nx = 'I'; mn = pI; }
				if(pD < mn) { nx = 'D'; mn = pD; }
			}
			else {
				const double pI = iD + tp[3];
				if(pB < mn) { nx = 'B'; mn = pB; }
				if(pI < mn) { nx = 'I'; mn = pI; }
			}
			st = nx; i--; j--;
		}
		else if(st == 'I') {
			double mU, iU, dU;
			vit_lookup(ctx, R, i - 1, j, mU, iU, dU);
			const double* tj = db.T + (size_t) j * 8;
			double mn = INFINITY; char nx;
			if(j > 0) {
				nx = 'M';
				const double pM = mU + tj[1], pI = iU + tj[4];
				if(pM < mn) { nx = 'M'; mn = pM; }
				if(pI < mn) { nx = 'I'; mn = pI; }
			}
			else {
				nx = 'B';
				const double pB = vit_bcol(ctx, i) + db.T[1], pI = iU + tj[4];
				if(pB < mn) { nx = 'B'; mn = pB; }
				if(pI < mn) { nx = 'I'; mn = pI; }
			}
			st = nx; i--;
		}
		else if(st == 'D') {
			double mL, iL, dL;
			vit_lookup(ctx, R, i, j - 1, mL, iL, dL);
			const double* tp = db.T + (size_t)(j - 1) * 8;
			double mn = INFINITY; char nx = 'M';
			const double pM = mL + tp[2], pD = dL + tp[6];
			if(pM < mn) { nx = 'M'; mn = pM; }
			if(pD < mn) { nx = 'D'; mn = pD; }
			st = nx; j--;
		}
		else break;
	}
	o.alnStart = j + 1; o.alnFrom = i + 1;
	if(tr[n - 1] != 'B') tr[n++] = 'B';
	for(int a = 0, b = n - 1; a < b; ++a, --b) { char t = tr[a]; tr[a] = tr[b]; tr[b] = t; }
	o.traceLen = n;
	o.status = (o.alnStart > 0 && o.alnFrom > 0) ? HU_READ_OK : HU_READ_INVALID;
	outs[s] = o;
}

/* traceback shared by both fill kernels: buildViterbiTrace (src/BandedHMMP7.cpp:943-1006) */
__device__ inline void vit_trace(const HuDbDev& db, const VitCtx& ctx, const HuReadDesc& rd, char* __restrict__ traces,
		double bestS, int bestCol, int bestRow, HuVitOut& o) {
	const int K = db.K;
	o.minScore = bestS; o.traceLen = 0; o.alnStart = o.alnEnd = o.alnFrom = o.alnTo = 0;
	if(!(bestS < INFINITY)) { o.status = HU_READ_NEEDS_FULL; return; }
	char* tr = traces + rd.traceOff;
	int n = 0;
	char st = bestCol <= K ? 'M' : 'I';
	int i = bestRow, j = bestCol <= K ? bestCol : K;
	o.alnEnd = j; o.alnTo = bestRow;
	const int R = rd.nRegions;
	tr[n++] = 'E';
	while(i >= 1 && j >= 0) {
		tr[n++] = st;
		if(st == 'M') {
			double mD, iD, dD;
			vit_lookup(ctx, R, i - 1, j - 1, mD, iD, dD);
			const double* tp = db.T + (size_t)(j - 1) * 8;
			const double pB = vit_bcol(ctx, i) + db.entryC[j];
			double mn = INFINITY; char nx = 'B';
			if(j > 1) {
				const double pM = mD + tp[0], pI = iD + tp[3], pD = dD + tp[5];
				if(pB < mn) { nx = 'B'; mn = pB; }
				if(pM < mn) { nx = 'M'; mn = pM; }
				if(pI < mn) { nx = 'I'; mn = pI; }
				if(pD < mn) { nx = 'D'; mn = pD; }
			}
			else {
				const double pI = iD + tp[3];
				if(pB < mn) { nx = 'B'; mn = pB; }
				if(pI < mn) { nx = 'I'; mn = pI; }
			}
			st = nx; i--; j--;
		}
		else if(st == 'I') {
			double mU, iU, dU;
			vit_lookup(ctx, R, i - 1, j, mU, iU, dU);
			const double* tj = db.T + (size_t) j * 8;
			double mn = INFINITY; char nx;
			if(j > 0) {
				nx = 'M';
				const double pM = mU + tj[1], pI = iU + tj[4];
				if(pM < mn) { nx = 'M'; mn = pM; }
				if(pI < mn) { nx = 'I'; mn = pI; }
			}
			else {
				nx = 'B';
				const double pB = vit_bcol(ctx, i) + db.T[1], pI = iU + tj[4];
				if(pB < mn) { nx = 'B'; mn = pB; }
				if(pI < mn) { nx = 'I'; mn = pI; }
			}
			st = nx; i--;
		}
		else if(st == 'D') {
			double mL, iL, dL;
			vit_lookup(ctx, R, i, j - 1, mL, iL, dL);
			const double* tp = db.T + (size_t)(j - 1) * 8;
			double mn = INFINITY; char nx = 'M';
			const double pM = mL + tp[2], pD = dL + tp[6];
			if(pM < mn) { nx = 'M'; mn = pM; }
			if(pD < mn) { nx = 'D'; mn = pD; }
			st = nx; j--;
		}
		else break;
	}
	o.alnStart = j + 1; o.alnFrom = i + 1;
	if(tr[n - 1] != 'B') tr[n++] = 'B';
	for(int a = 0, b = n - 1; a < b; ++a, --b) { char t = tr[a]; tr[a] = tr[b]; tr[b] = t; }
	o.traceLen = n;
	o.status = (o.alnStart > 0 && o.alnFrom > 0) ? HU_READ_OK : HU_READ_INVALID;
}

/* LDS-staged variant: the last two anti-diagonals of the running phase live in LDS (three rotating
 * buffers x {M,I,D} x read length), so the inner loop touches HBM only to file each cell for the
 * traceback (fire-and-forget stores) and for the few perimeter cells that read an earlier phase.
 * The S minimum is tracked while filling.  One workgroup of 4 waves per sequence: a 250-bp read has
 * anti-diagonals of up to ~210 cells, so 256 lanes take a whole diagonal per step, and the LDS
 * footprint (72 B x read length) caps a CU at 8 sequences = 32 waves, the full complement.
 * Between diagonals only LDS must be ordered: s_waitcnt lgkmcnt(0) + s_barrier, never vmcnt. */
#define HU_VIT_THREADS 256
__device__ inline void vit_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(HU_VIT_THREADS) void k_viterbi_lds(HuDbDev db, const HuReadDesc* __restrict__ descs, const char* __restrict__ bases,
		double* __restrict__ scratch, char* __restrict__ traces, double tNN, double tNB, double tEC, double tCC,
		HuVitOut* __restrict__ outs, int ldsRows, int onlyStatus) {
	extern __shared__ double vsh[];
	__shared__ double redS[HU_VIT_THREADS / 64];
	__shared__ int redC[HU_VIT_THREADS / 64], redR[HU_VIT_THREADS / 64];
	const int s = blockIdx.x, tid = threadIdx.x;
	if(onlyStatus && outs[s].status != onlyStatus) return;   /* redo pass: only the sequences flagged by k_viterbi_trace_dec */
	const HuReadDesc& rd = descs[s];
	const int L = rd.len, K = db.K;
	if(rd.nRegions <= 0) { if(tid == 0) { HuVitOut o = {0, 0, 0, 0, 0, HU_READ_INVALID, INFINITY}; outs[s] = o; } return; }
	const char* __restrict__ x = bases + rd.baseOff;
	double* scr = scratch + rd.scratchOff * 3;
	VitCtx ctx = { &rd, scr, tNN, tNB };
	double bestS = INFINITY; int bestCol = 0x7fffffff, bestRow = 0x7fffffff;
	const int nR = rd.nRegions;
	for(int r = 0; r < nR; ++r) {
		const HuRegion g = rd.reg[r];
		const int ni = g.i1 - g.i0 + 1, nj = g.j1 - g.j0 + 1;
		if(ni <= 0 || nj <= 0) continue;
		for(int dg = 0; dg <= ni + nj - 2; ++dg) {
			double* cur = vsh + (size_t)(dg % 3) * 3 * ldsRows;
			const double* p1 = vsh + (size_t)((dg + 2) % 3) * 3 * ldsRows;
			const double* p2 = vsh + (size_t)((dg + 1) % 3) * 3 * ldsRows;
			const int lo = dg - (nj - 1) > 0 ? dg - (nj - 1) : 0, hi = dg < ni - 1 ? dg : ni - 1;
			for(int q = lo + tid; q <= hi; q += HU_VIT_THREADS) {
				const int i = g.i0 + q, j = g.j0 + dg - q;
				const int dist = (i - g.from) - (j - g.start);
				if(g.band && !(dist <= g.nIns && dist >= -g.nDel)) continue;
				const int b = c_sym_map[(int) x[i - 1] & 127];
				double mD, iD, dD, mU, iU, dU, mL, iL, dL;
				/* (i-1, j-1): band distance unchanged */
				if(q >= 1 && j - 1 >= g.j0) { mD = p2[q - 1]; iD = p2[ldsRows + q - 1]; dD = p2[2 * ldsRows + q - 1]; }
				else vit_lookup(ctx, r, i - 1, j - 1, mD, iD, dD);
				/* (i-1, j): distance - 1 */
				if(q >= 1 && (!g.band || dist - 1 >= -g.nDel)) { mU = p1[q - 1]; iU = p1[ldsRows + q - 1]; }
				else vit_lookup(ctx, r, i - 1, j, mU, iU, dU);
				/* (i, j-1): distance + 1 */
				if(j - 1 >= g.j0 && (!g.band || dist + 1 <= g.nIns)) { mL = p1[q]; dL = p1[2 * ldsRows + q]; }
				else vit_lookup(ctx, r, i, j - 1, mL, iL, dL);
				const double* tp = db.T + (size_t)(j - 1) * 8;
				const double* tj = db.T + (size_t) j * 8;
				double best = fmin(mD + tp[0], fmin(iD + tp[3], dD + tp[5]));
				if(g.withB) best = fmin(vit_bcol(ctx, i) + db.entryC[j], best);
				const double M = db.EM[(size_t) j * 4 + b] + best;
				const double I = db.EI[(size_t) j * 4 + b] + fmin(mU + tj[1], iU + tj[4]);
				const double D = (j > 1 && j < K) ? fmin(mL + tp[2], dL + tp[6]) : INFINITY;
				cur[q] = M; cur[ldsRows + q] = I; cur[2 * ldsRows + q] = D;
				const int64_t idx = (g.off + (int64_t)(j - g.j0) * ni + q) * 3;
				scr[idx] = M; scr[idx + 1] = I; scr[idx + 2] = D;
				/* S candidates of this cell unless a later phase recomputes it */
				bool later = false;
				for(int r2 = r + 1; r2 < nR; ++r2) if(reg_contains(rd.reg[r2], i, j)) later = true;
				if(!later) {
					const double cc = (i < L) ? __dmul_rn(tCC, (double)(L - i)) : 0.0;
					double sv = __dadd_rn(__dadd_rn(M, db.exitC[j]), tEC);
					if(i < L) sv = __dadd_rn(sv, cc);
					if(sv < bestS || (sv == bestS && (j < bestCol || (j == bestCol && i < bestRow)))) { bestS = sv; bestCol = j; bestRow = i; }
					if(j == K) {
						double s2 = __dadd_rn(__dadd_rn(I, db.T[(size_t) K * 8 + 3]), tEC);
						if(i < L) s2 = __dadd_rn(s2, cc);
						if(s2 < bestS || (s2 == bestS && (K + 1 < bestCol || (K + 1 == bestCol && i < bestRow)))) { bestS = s2; bestCol = K + 1; bestRow = i; }
					}
				}
			}
			vit_lds_barrier();
		}
	}
	for(int m = 32; m > 0; m >>= 1) {
		const double os = __shfl_xor(bestS, m); const int oc = __shfl_xor(bestCol, m), orow = __shfl_xor(bestRow, m);
		if(os < bestS || (os == bestS && (oc < bestCol || (oc == bestCol && orow < bestRow)))) { bestS = os; bestCol = oc; bestRow = orow; }
	}
	if((tid & 63) == 0) { redS[tid >> 6] = bestS; redC[tid >> 6] = bestCol; redR[tid >> 6] = bestRow; }
	__syncthreads(); /* also: every cell filed in HBM before thread 0 walks back through them */
	if(tid != 0) return;
	for(int wv = 1; wv < HU_VIT_THREADS / 64; ++wv) {
		const double os = redS[wv]; const int oc = redC[wv], orow = redR[wv];
		if(os < bestS || (os == bestS && (oc < bestCol || (oc == bestCol && orow < bestRow)))) { bestS = os; bestCol = oc; bestRow = orow; }
	}
	/* the traceback runs in k_viterbi_trace, one LANE per sequence: it is a chain of dependent
	 * loads, so thousands of them must be in flight at once, not one per workgroup */
	HuVitOut o;
	o.minScore = bestS; o.alnEnd = bestCol; o.alnTo = bestRow; o.alnStart = o.alnFrom = 0; o.traceLen = -1; o.status = HU_READ_NEEDS_FULL;
	outs[s] = o;
}

__global__ __launch_bounds__(64) void k_viterbi_trace(HuDbDev db, const HuReadDesc* __restrict__ descs, const double* __restrict__ scratch,
		char* __restrict__ traces, double tNN, double tNB, HuVitOut* __restrict__ outs, int nSeq) {
	const int s = blockIdx.x * 64 + threadIdx.x;
	if(s >= nSeq) return;
	HuVitOut o = outs[s];
	if(o.traceLen != -1 || o.status == HU_READ_NEEDS_VALUES) return; /* invalid read, already traced, or waiting for the redo pass */
	const HuReadDesc& rd = descs[s];
	VitCtx ctx = { &rd, scratch + rd.scratchOff * 3, tNN, tNB };
	const double bestS = o.minScore; const int bestCol = o.alnEnd, bestRow = o.alnTo;
	vit_trace(db, ctx, rd, traces, bestS, bestCol, bestRow, o);
	outs[s] = o;
}

/* ------------------------------------------------------------------------------------------------
 * Decision-byte variant of the LDS wavefront.  The reference's traceback re-evaluates, at every cell of
 * the path, the candidate sums of its predecessor on the FINAL matrices (whichMin, first strict minimum
 * in the order B, M, I, D).  Those sums are known when the cell is filled, and they are the final ones
 * unless a LATER phase rewrites the predecessor — which the traceback checks per step from the region
 * descriptors.  So the fill kernel files ONE byte per cell (what the traceback would choose in state M, I
 * and D at that cell) instead of three doubles, in anti-diagonal order so that a step's stores are
 * consecutive bytes, and the (M, I, D) values only of the cells inside the boxes of later phases (their
 * perimeter look-ups).  k_viterbi_lds wrote 21 GB per 8192 reads of 250 bp and ran at the speed of those
 * stores; this one writes 0.4 GB.  A sequence whose traceback meets a rewritten predecessor, or leaves the
 * computed cells, is flagged HU_READ_NEEDS_VALUES and redone by the value-filing kernels.
 * Profile fields come from the per-field arrays (HuDbDev::Tt/EMt/EIt): consecutive lanes, consecutive columns. */
__global__ __launch_bounds__(HU_VIT_THREADS) void k_viterbi_dec(HuDbDev db, const HuReadDesc* __restrict__ descs, const char* __restrict__ bases,
		double* __restrict__ scratch, uint8_t* __restrict__ dec, double tNN, double tNB, double tEC, double tCC,
		HuVitOut* __restrict__ outs, int ldsRows) {
	extern __shared__ double vsh[];
	__shared__ double redS[HU_VIT_THREADS / 64];
	__shared__ int redC[HU_VIT_THREADS / 64], redR[HU_VIT_THREADS / 64];
	const int s = blockIdx.x, tid = threadIdx.x;
	const HuReadDesc& rd = descs[s];
	const int L = rd.len, K = db.K;
	const size_t K1 = (size_t) K + 1;
	if(rd.nRegions <= 0) { if(tid == 0) { HuVitOut o = {0, 0, 0, 0, 0, HU_READ_INVALID, INFINITY}; outs[s] = o; } return; }
	const char* __restrict__ x = bases + rd.baseOff;
	double* scr = scratch + rd.scratchOff * 3;
	uint8_t* dcs = dec + rd.decOff;
	VitCtx ctx = { &rd, scr, tNN, tNB };
	double bestS = INFINITY; int bestCol = 0x7fffffff, bestRow = 0x7fffffff;
	const int nR = rd.nRegions;
	for(int r = 0; r < nR; ++r) {
		const HuRegion g = rd.reg[r];
		const int ni = g.i1 - g.i0 + 1, nj = g.j1 - g.j0 + 1;
		if(ni <= 0 || nj <= 0) continue;
		for(int dg = 0; dg <= ni + nj - 2; ++dg) {
			double* cur = vsh + (size_t)(dg % 3) * 3 * ldsRows;
			const double* p1 = vsh + (size_t)((dg + 2) % 3) * 3 * ldsRows;
			const double* p2 = vsh + (size_t)((dg + 1) % 3) * 3 * ldsRows;
			const int lo = dg - (nj - 1) > 0 ? dg - (nj - 1) : 0, hi = dg < ni - 1 ? dg : ni - 1;
			for(int q = lo + tid; q <= hi; q += HU_VIT_THREADS) {
				const int i = g.i0 + q, j = g.j0 + dg - q;
				const int dist = (i - g.from) - (j - g.start);
				if(g.band && !(dist <= g.nIns && dist >= -g.nDel)) continue;
				const int b = c_sym_map[(int) x[i - 1] & 127];
				double mD, iD, dD, mU, iU, dU, mL, iL, dL;
				if(q >= 1 && j - 1 >= g.j0) { mD = p2[q - 1]; iD = p2[ldsRows + q - 1]; dD = p2[2 * ldsRows + q - 1]; }
				else vit_lookup(ctx, r, i - 1, j - 1, mD, iD, dD);
				if(q >= 1 && (!g.band || dist - 1 >= -g.nDel)) { mU = p1[q - 1]; iU = p1[ldsRows + q - 1]; }
				else vit_lookup(ctx, r, i - 1, j, mU, iU, dU);
				if(j - 1 >= g.j0 && (!g.band || dist + 1 <= g.nIns)) { mL = p1[q]; dL = p1[2 * ldsRows + q]; }
				else vit_lookup(ctx, r, i, j - 1, mL, iL, dL);
				const double* tp = db.Tt + (j - 1);   /* field f of column j - 1: tp[f * K1] */
				const double* tj = db.Tt + j;
				const double pB = vit_bcol(ctx, i) + db.entryC[j];
				const double pM = mD + tp[0], pI = iD + tp[3 * K1], pD = dD + tp[5 * K1];
				double best = fmin(pM, fmin(pI, pD));
				if(g.withB) best = fmin(pB, best);
				const double uM = mU + tj[1 * K1], uI = iU + tj[4 * K1];
				const double lM = mL + tp[2 * K1], lD = dL + tp[6 * K1];
				const double M = db.EMt[(size_t) b * K1 + j] + best;
				const double I = db.EIt[(size_t) b * K1 + j] + fmin(uM, uI);
				const double D = (j > 1 && j < K) ? fmin(lM, lD) : INFINITY;
				cur[q] = M; cur[ldsRows + q] = I; cur[2 * ldsRows + q] = D;
				/* what buildViterbiTrace (src/BandedHMMP7.cpp:956-1000) chooses at this cell: B is a candidate in
				 * every phase there, and column 1 has no M / D predecessor */
				int dM = 0;
				{
					double mn = INFINITY;
					if(pB < mn) { dM = 0; mn = pB; }
					if(j > 1 && pM < mn) { dM = 1; mn = pM; }
					if(pI < mn) { dM = 2; mn = pI; }
					if(j > 1 && pD < mn) { dM = 3; mn = pD; }
				}
				const int dI = uI < uM ? 1 : 0, dDd = lD < lM ? 1 : 0;
				dcs[g.doff + (int64_t) dg * ((ni + 15) & ~15) + q] = (uint8_t)(dM | (dI << 2) | (dDd << 3));
				bool later = false, near = false;
				for(int r2 = r + 1; r2 < nR; ++r2) {
					const HuRegion& g2 = rd.reg[r2];
					if(reg_contains(g2, i, j)) later = true;
					if(i >= g2.i0 - 1 && i <= g2.i1 && j >= g2.j0 - 1 && j <= g2.j1) near = true;
				}
				if(near) { /* a later phase may look this cell up */
					const int64_t idx = (g.off + (int64_t)(j - g.j0) * ni + q) * 3;
					scr[idx] = M; scr[idx + 1] = I; scr[idx + 2] = D;
				}
				if(!later) {
					const double cc = (i < L) ? __dmul_rn(tCC, (double)(L - i)) : 0.0;
					double sv = __dadd_rn(__dadd_rn(M, db.exitC[j]), tEC);
					if(i < L) sv = __dadd_rn(sv, cc);
					if(sv < bestS || (sv == bestS && (j < bestCol || (j == bestCol && i < bestRow)))) { bestS = sv; bestCol = j; bestRow = i; }
					if(j == K) {
						double s2 = __dadd_rn(__dadd_rn(I, db.T[(size_t) K * 8 + 3]), tEC);
						if(i < L) s2 = __dadd_rn(s2, cc);
						if(s2 < bestS || (s2 == bestS && (K + 1 < bestCol || (K + 1 == bestCol && i < bestRow)))) { bestS = s2; bestCol = K + 1; bestRow = i; }
					}
				}
			}
			vit_lds_barrier();
		}
		/* the values filed for later phases must have left this CU's write path before they are looked up */
		__syncthreads();
	}
	for(int m = 32; m > 0; m >>= 1) {
		const double os = __shfl_xor(bestS, m); const int oc = __shfl_xor(bestCol, m), orow = __shfl_xor(bestRow, m);
		if(os < bestS || (os == bestS && (oc < bestCol || (oc == bestCol && orow < bestRow)))) { bestS = os; bestCol = oc; bestRow = orow; }
	}
	if((tid & 63) == 0) { redS[tid >> 6] = bestS; redC[tid >> 6] = bestCol; redR[tid >> 6] = bestRow; }
	__syncthreads();
	if(tid != 0) return;
	for(int wv = 1; wv < HU_VIT_THREADS / 64; ++wv) {
		const double os = redS[wv]; const int oc = redC[wv], orow = redR[wv];
		if(os < bestS || (os == bestS && (oc < bestCol || (oc == bestCol && orow < bestRow)))) { bestS = os; bestCol = oc; bestRow = orow; }
	}
	HuVitOut o;
	o.minScore = bestS; o.alnEnd = bestCol; o.alnTo = bestRow; o.alnStart = o.alnFrom = 0; o.traceLen = -1; o.status = HU_READ_NEEDS_FULL;
	outs[s] = o;
}

/* The same with one DP row per thread (reads of at most THREADS bases): the row's base, B-column term and C-loop
 * term are per-phase constants, the profile column advances by one per step, so the eleven profile values of the
 * NEXT step's cell are requested while the current cell is computed (no load sits between two barriers), and the
 * later-phase bookkeeping is skipped for cells outside the bounding corner of all later phases. */
template<int THREADS>
__global__ __launch_bounds__(THREADS) void k_viterbi_dec2(HuDbDev db, const HuReadDesc* __restrict__ descs, const char* __restrict__ bases,
		double* __restrict__ scratch, uint8_t* __restrict__ dec, double tNN, double tNB, double tEC, double tCC,
		HuVitOut* __restrict__ outs, int ldsRows, int haloW) {
	extern __shared__ double vsh[];
	double* halo = vsh + (size_t) 9 * ldsRows;   /* [3][haloW]: (M, I, D) of row i0 - 1, columns j0 - 1 .. j1 */
	/* decision bytes of the last 32 anti-diagonals, [slot][row]; 16 at a time leave for HBM as 16-byte stores.
	 * vmcnt retires in issue order, so a store per step would put its round trip to L2 in front of the next
	 * step's prefetched profile values: one store per 16 steps instead */
	uint8_t* ring = reinterpret_cast<uint8_t*>(halo + (size_t) 3 * haloW);
	__shared__ double redS[THREADS / 64];
	__shared__ int redC[THREADS / 64], redR[THREADS / 64];
	__shared__ HuRegion sreg[HU_MAX_REGIONS];
	const int s = blockIdx.x, tid = threadIdx.x;
	const HuReadDesc& rd = descs[s];
	const int L = rd.len, K = db.K;
	const size_t K1 = (size_t) K + 1;
	const int nR = rd.nRegions;
	if(nR <= 0) { if(tid == 0) { HuVitOut o = {0, 0, 0, 0, 0, HU_READ_INVALID, INFINITY}; outs[s] = o; } return; }
	if(tid < nR) sreg[tid] = rd.reg[tid];
	__syncthreads();
	const char* __restrict__ x = bases + rd.baseOff;
	double* scr = scratch + rd.scratchOff * 3;
	uint8_t* dcs = dec + rd.decOff;
	VitCtx ctx = { &rd, scr, tNN, tNB };
	/* vit_lookup on the LDS copy of the descriptor */
	auto look = [&](int upto, int ii, int jj, double& m, double& iv, double& d) {
		for(int rr = upto - 1; rr >= 0; --rr) {
			const HuRegion& gg = sreg[rr];
			if(reg_contains(gg, ii, jj)) {
				const int64_t idx = (gg.off + (int64_t)(jj - gg.j0) * (gg.i1 - gg.i0 + 1) + (ii - gg.i0)) * 3;
				m = scr[idx]; iv = scr[idx + 1]; d = scr[idx + 2];
				return;
			}
		}
		if(jj == 0 && ii >= 1) { const double vv = vit_bcol(ctx, ii); m = vv; iv = vv; d = INFINITY; return; }
		m = iv = d = INFINITY;
	};
	double bestS = INFINITY; int bestCol = 0x7fffffff, bestRow = 0x7fffffff;
	const double tKIM = db.T[(size_t) K * 8 + 3];
	for(int r = 0; r < nR; ++r) {
		const HuRegion g = sreg[r];
		const int ni = g.i1 - g.i0 + 1, nj = g.j1 - g.j0 + 1;
		if(ni <= 0 || nj <= 0) continue;
		const int P = (ni + 15) & ~15;
		/* cells with i < nearI or j < nearJ are neither inside nor next to any later phase */
		int nearI = 0x7fffffff, nearJ = 0x7fffffff;
		for(int r2 = r + 1; r2 < nR; ++r2) {
			const int a = sreg[r2].i0 - 1, c = sreg[r2].j0 - 1;
			if(sreg[r2].i1 >= sreg[r2].i0 && sreg[r2].j1 >= sreg[r2].j0) { nearI = a < nearI ? a : nearI; nearJ = c < nearJ ? c : nearJ; }
		}
		const int q = tid, i = g.i0 + q;
		const bool row = q < ni;
		const int b = row ? c_sym_map[(int) x[i - 1] & 127] : 0;
		const double bcol = vit_bcol(ctx, i);
		const double cc = (i < L) ? __dmul_rn(tCC, (double)(L - i)) : 0.0;
		const double* __restrict__ emb = db.EMt + (size_t) b * K1;
		const double* __restrict__ eib = db.EIt + (size_t) b * K1;
		/* profile values of the cell this thread computes next: column jc */
		double nT0, nT3, nT5, nT2, nT6, nJ1, nJ4, nEM, nEI, nEN, nEX;
		auto fetch = [&](int jc) {
			const double* tp = db.Tt + (jc - 1);
			nT0 = tp[0]; nT3 = tp[3 * K1]; nT5 = tp[5 * K1]; nT2 = tp[2 * K1]; nT6 = tp[6 * K1];
			nJ1 = tp[1 * K1 + 1]; nJ4 = tp[4 * K1 + 1];
			nEM = emb[jc]; nEI = eib[jc]; nEN = db.entryC[jc]; nEX = db.exitC[jc];
		};
		nT0 = nT3 = nT5 = nT2 = nT6 = nJ1 = nJ4 = nEM = nEI = nEN = nEX = 0;
		if(row && q == 0) fetch(g.j0);
		/* every value of an earlier phase that this phase will look at, fetched now and in parallel: the first
		 * column's neighbours and the band's outer neighbours of this thread's row into registers, row i0 - 1 into
		 * LDS.  Inside the wavefront no step waits on global memory. */
		double cDm = INFINITY, cDi = INFINITY, cDd = INFINITY, cLm = INFINITY, cLd = INFINITY;
		double eUm = INFINITY, eUi = INFINITY, eLm = INFINITY, eLd = INFINITY, tmp;
		if(row) {
			look(r, i - 1, g.j0 - 1, cDm, cDi, cDd);
			look(r, i, g.j0 - 1, cLm, tmp, cLd);
			if(g.band) {
				const int jU = (i - g.from) + g.nDel + g.start, jL = (i - g.from) - g.nIns + g.start;
				if(jU >= g.j0 && jU <= g.j1) look(r, i - 1, jU, eUm, eUi, tmp);
				if(jL >= g.j0 && jL <= g.j1) look(r, i, jL - 1, eLm, tmp, eLd);
			}
		}
		for(int c = tid; c <= nj; c += THREADS) {
			double hm, hi2, hd;
			look(r, g.i0 - 1, g.j0 - 1 + c, hm, hi2, hd);
			halo[c] = hm; halo[haloW + c] = hi2; halo[2 * haloW + c] = hd;
		}
		__syncthreads();
		for(int dg = 0; dg <= ni + nj - 2; ++dg) {
			double* cur = vsh + (size_t)(dg % 3) * 3 * ldsRows;
			const double* p1 = vsh + (size_t)((dg + 2) % 3) * 3 * ldsRows;
			const double* p2 = vsh + (size_t)((dg + 1) % 3) * 3 * ldsRows;
			const int j = g.j0 + dg - q;
			const bool cell = row && j >= g.j0 && j <= g.j1;
			const double T0 = nT0, T3 = nT3, T5 = nT5, T2 = nT2, T6 = nT6, J1 = nJ1, J4 = nJ4, EMv = nEM, EIv = nEI, ENv = nEN, EXv = nEX;
			if(row && j + 1 >= g.j0 && j + 1 <= g.j1) fetch(j + 1);
			const int dist = (i - g.from) - (j - g.start);
			if(cell && !(g.band && !(dist <= g.nIns && dist >= -g.nDel))) {
				double mD, iD, dD, mU, iU, mL, dL;
				if(q >= 1 && j - 1 >= g.j0) { mD = p2[q - 1]; iD = p2[ldsRows + q - 1]; dD = p2[2 * ldsRows + q - 1]; }
				else if(q == 0) { mD = halo[j - g.j0]; iD = halo[haloW + j - g.j0]; dD = halo[2 * haloW + j - g.j0]; }   /* (i0 - 1, j - 1) */
				else { mD = cDm; iD = cDi; dD = cDd; }                                                            /* (i - 1, j0 - 1) */
				if(q >= 1 && (!g.band || dist - 1 >= -g.nDel)) { mU = p1[q - 1]; iU = p1[ldsRows + q - 1]; }
				else if(q == 0) { mU = halo[j - g.j0 + 1]; iU = halo[haloW + j - g.j0 + 1]; }                   /* (i0 - 1, j) */
				else { mU = eUm; iU = eUi; }                                                                     /* above the band */
				if(j - 1 >= g.j0 && (!g.band || dist + 1 <= g.nIns)) { mL = p1[q]; dL = p1[2 * ldsRows + q]; }
				else if(j == g.j0) { mL = cLm; dL = cLd; }                                                       /* (i, j0 - 1) */
				else { mL = eLm; dL = eLd; }                                                                     /* left of the band */
				const double pB = bcol + ENv;
				const double pM = mD + T0, pI = iD + T3, pD = dD + T5;
				double best = fmin(pM, fmin(pI, pD));
				if(g.withB) best = fmin(pB, best);
				const double uM = mU + J1, uI = iU + J4;
				const double lM = mL + T2, lD = dL + T6;
				const double M = EMv + best;
				const double I = EIv + fmin(uM, uI);
				const double D = (j > 1 && j < K) ? fmin(lM, lD) : INFINITY;
				cur[q] = M; cur[ldsRows + q] = I; cur[2 * ldsRows + q] = D;
				int dM = 0;
				{
					double mn = INFINITY;
					if(pB < mn) { dM = 0; mn = pB; }
					if(j > 1 && pM < mn) { dM = 1; mn = pM; }
					if(pI < mn) { dM = 2; mn = pI; }
					if(j > 1 && pD < mn) { dM = 3; mn = pD; }
				}
				const int dI = uI < uM ? 1 : 0, dDd = lD < lM ? 1 : 0;
				ring[(dg & 31) * THREADS + q] = (uint8_t)(dM | (dI << 2) | (dDd << 3));
				bool later = false;
				if(i >= nearI && j >= nearJ) {
					bool near = false;
					for(int r2 = r + 1; r2 < nR; ++r2) {
						const HuRegion& g2 = sreg[r2];
						if(reg_contains(g2, i, j)) later = true;
						if(i >= g2.i0 - 1 && i <= g2.i1 && j >= g2.j0 - 1 && j <= g2.j1) near = true;
					}
					if(near) { /* a later phase may look this cell up */
						const int64_t idx = (g.off + (int64_t)(j - g.j0) * ni + q) * 3;
						scr[idx] = M; scr[idx + 1] = I; scr[idx + 2] = D;
					}
				}
				if(!later) {
					double sv = __dadd_rn(__dadd_rn(M, EXv), tEC);
					if(i < L) sv = __dadd_rn(sv, cc);
					if(sv < bestS || (sv == bestS && (j < bestCol || (j == bestCol && i < bestRow)))) { bestS = sv; bestCol = j; bestRow = i; }
					if(j == K) {
						double s2 = __dadd_rn(__dadd_rn(I, tKIM), tEC);
						if(i < L) s2 = __dadd_rn(s2, cc);
						if(s2 < bestS || (s2 == bestS && (K + 1 < bestCol || (K + 1 == bestCol && i < bestRow)))) { bestS = s2; bestCol = K + 1; bestRow = i; }
					}
				}
			}
			vit_lds_barrier();
			if((dg & 15) == 15 || dg == ni + nj - 2) { /* file the diagonals dg0 .. dg: pitch P bytes, P / 16 units of 16 bytes each */
				const int dg0 = dg & ~15, units = P >> 4, cnt = (dg - dg0 + 1) * units;
				if(tid < cnt) {
					const int dd = tid / units, c = tid - dd * units;
					const uint4 v = *reinterpret_cast<const uint4*>(ring + ((dg0 + dd) & 31) * THREADS + c * 16);
					*reinterpret_cast<uint4*>(dcs + g.doff + (int64_t)(dg0 + dd) * P + c * 16) = v;
				}
			}
		}
		__syncthreads();
	}
	for(int m = 32; m > 0; m >>= 1) {
		const double os = __shfl_xor(bestS, m); const int oc = __shfl_xor(bestCol, m), orow = __shfl_xor(bestRow, m);
		if(os < bestS || (os == bestS && (oc < bestCol || (oc == bestCol && orow < bestRow)))) { bestS = os; bestCol = oc; bestRow = orow; }
	}
	if((tid & 63) == 0) { redS[tid >> 6] = bestS; redC[tid >> 6] = bestCol; redR[tid >> 6] = bestRow; }
	__syncthreads();
	if(tid != 0) return;
	for(int wv = 1; wv < THREADS / 64; ++wv) {
		const double os = redS[wv]; const int oc = redC[wv], orow = redR[wv];
		if(os < bestS || (os == bestS && (oc < bestCol || (oc == bestCol && orow < bestRow)))) { bestS = os; bestCol = oc; bestRow = orow; }
	}
	HuVitOut o;
	o.minScore = bestS; o.alnEnd = bestCol; o.alnTo = bestRow; o.alnStart = o.alnFrom = 0; o.traceLen = -1; o.status = HU_READ_NEEDS_FULL;
	outs[s] = o;
}

/* ------------------------------------------------------------------------------------------------
 * k_viterbi_wave: the decision-byte DP with ONE WAVE per sequence and no barrier.
 *
 * Rectangular phases: lane l owns the RPL consecutive rows i0 + l RPL .. of the phase and walks the columns;
 * at step t it is at column j0 + t - l, so the lanes form a wavefront over blocks of RPL x 1 cells.  Inside a
 * lane the column's cells are computed top to bottom from registers: the left neighbour of a cell is the
 * lane's own value of the previous step, its upper neighbour the cell just computed, its diagonal neighbour the
 * previous-step value of the row above.  Only row 0 of a lane looks outside: its upper and diagonal
 * neighbours are the last row of lane l - 1 one resp. two steps ago — ONE triple of doubles moved down a
 * lane per step over DPP (wave_shr:1), used as "up" now and as "diagonal" at the next step.  Lane 0 takes
 * that triple from the row above the phase (values of earlier phases, put into LDS when the phase starts),
 * and every row starts from the column left of the phase.
 * Band phases (a seed: <= 64 rows) run one row per lane on anti-diagonals with the same triple; the cells
 * just outside the band (one above and one left of the band per row) are looked up when the phase starts.
 * Per step a lane asks for the profile values of its NEXT column (9 + 2 RPL doubles from the per-field
 * arrays: consecutive lanes, consecutive columns) and files RPL decision bytes as one 32/64-bit store.
 * k_viterbi_dec2 (a workgroup per sequence, a barrier per anti-diagonal) spent ~300 wave-instructions per
 * anti-diagonal and wave; this kernel spends ~60 per cell and has nothing to wait for but its own loads. */
template<int RPL> struct HuDecWord { typedef uint32_t type; };
template<> struct HuDecWord<8> { typedef unsigned long long type; };

__device__ inline double dpp_wave_shr1(double v, double lane0) { /* value of lane l - 1; lane 0 gets `lane0` */
	const int lo = __builtin_amdgcn_update_dpp(__double2loint(lane0), __double2loint(v), 0x138, 0xf, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), __double2hiint(v), 0x138, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}

/* running minimum of S with Eigen's column-major first-minimum rule as one unsigned key (column - 1) 2^16 + row
 * (columns 1 .. K + 1 <= 65536, rows <= 65535); branch-free.  A candidate of +inf never displaces anything: when
 * every candidate is +inf the sequence has no path and its position is not used. */
struct VwBest { double s; uint32_t key; };
__device__ inline void vw_cand(VwBest& b, double sv, int j, int i, bool ok) {
	const uint32_t key = ((uint32_t)(j - 1) << 16) + (uint32_t) i;
	const bool better = ok & ((sv < b.s) | ((sv == b.s) & (sv < INFINITY) & (key < b.key)));
	b.s = better ? sv : b.s; b.key = better ? key : b.key;
}

template<int RPL, int DIAG = 0>
__global__ __launch_bounds__(64) void k_viterbi_wave(HuDbDev db, const HuReadDesc* __restrict__ descs, const char* __restrict__ bases,
		double* __restrict__ scratch, uint8_t* __restrict__ dec, double tNN, double tNB, double tEC, double tCC,
		HuVitOut* __restrict__ outs, int haloW) {
	extern __shared__ double halo[];                 /* [3][haloW]: (M, I, D) of row i0 - 1, columns j0 - 1 .. j1 */
	__shared__ HuRegion sreg[HU_MAX_REGIONS];
	typedef typename HuDecWord<RPL>::type dword_t;
	const int s = blockIdx.x, lane = threadIdx.x;
	const HuReadDesc& rd = descs[s];
	const int L = rd.len, K = db.K;
	const size_t K1 = (size_t) K + 1;
	const int nR = rd.nRegions;
	if(nR <= 0) { if(lane == 0) { HuVitOut o = {0, 0, 0, 0, 0, HU_READ_INVALID, INFINITY}; outs[s] = o; } return; }
	if(lane < nR) sreg[lane] = rd.reg[lane];
	__syncthreads();
	for(int r = 0; r < nR; ++r) if(sreg[r].band && sreg[r].i1 - sreg[r].i0 + 1 > 64) { /* a band wider than a wave: the value-filing kernels take the sequence */
		if(lane == 0) { HuVitOut o = {0, 0, 0, 0, -1, HU_READ_NEEDS_VALUES, INFINITY}; outs[s] = o; }
		return;
	}
	const char* __restrict__ x = bases + rd.baseOff;
	double* scr = scratch + rd.cornerOff * 3;        /* the CORNER scratch: only cells a later phase looks up are filed (HuRegion::coff) */
	uint8_t* dcs = dec + rd.decOff;
	VitCtx ctx = { &rd, scr, tNN, tNB };
	auto look = [&](int upto, int ii, int jj, double& m, double& iv, double& d) {
		for(int rr = upto - 1; rr >= 0; --rr) {
			const HuRegion& gg = sreg[rr];
			if(reg_contains(gg, ii, jj)) {
				/* every cell this kernel looks up lies in the row above / the column left of / inside a LATER phase than gg, hence in gg's
				 * corner block (rows >= ci0, columns >= cj0); a cell outside it was never filed and reads as "not computed" */
				if(ii < gg.ci0 || jj < gg.cj0) { m = iv = d = INFINITY; return; }
				const int64_t idx = (gg.coff + (int64_t)(jj - gg.cj0) * (gg.i1 - gg.ci0 + 1) + (ii - gg.ci0)) * 3;
				m = scr[idx]; iv = scr[idx + 1]; d = scr[idx + 2];
				return;
			}
		}
		if(jj == 0 && ii >= 1) { const double vv = vit_bcol(ctx, ii); m = vv; iv = vv; d = INFINITY; return; }
		m = iv = d = INFINITY;
	};
	VwBest best = { INFINITY, 0xffffffffu };
	const double tKIM = db.T[(size_t) K * 8 + 3];
	for(int r = 0; r < nR; ++r) {
		const HuRegion g = sreg[r];
		const int ni = g.i1 - g.i0 + 1, nj = g.j1 - g.j0 + 1;
		if(ni <= 0 || nj <= 0) continue;
		int nearI = 0x7fffffff, nearJ = 0x7fffffff;
		for(int r2 = r + 1; r2 < nR; ++r2) {
			const int a = sreg[r2].i0 - 1, c = sreg[r2].j0 - 1;
			if(sreg[r2].i1 >= sreg[r2].i0 && sreg[r2].j1 >= sreg[r2].j0) { nearI = a < nearI ? a : nearI; nearJ = c < nearJ ? c : nearJ; }
		}
		/* row i0 - 1 of the earlier phases: column j0 - 1 + c at halo[c] */
		const bool useHalo = nj + 2 <= haloW;
		if(useHalo) for(int c = lane; c <= nj; c += 64) {
			double hm, hi2, hd;
			look(r, g.i0 - 1, g.j0 - 1 + c, hm, hi2, hd);
			halo[c] = hm; halo[haloW + c] = hi2; halo[2 * haloW + c] = hd;
		}
		__syncthreads();
		/* what a cell files and offers: decision byte and S candidates.  Cells in the bounding corner of the later
		 * phases (i >= nearI and j >= nearJ: a handful per phase) are left to corner(): they may belong to a later
		 * phase (no S candidate) or be looked up by one (values filed). */
		auto finish = [&](int i, int j, double M, double I, double pB, double pM, double pI, double pD,
				double uM, double uI, double lM, double lD, double EXv) -> int {
			int dM = 0;
			{
				double mn = INFINITY;
				if(pB < mn) { dM = 0; mn = pB; }
				if(j > 1 && pM < mn) { dM = 1; mn = pM; }
				if(pI < mn) { dM = 2; mn = pI; }
				if(j > 1 && pD < mn) { dM = 3; mn = pD; }
			}
			{
				const bool notCorner = !(i >= nearI && j >= nearJ);
				const double ccv = __dmul_rn(tCC, (double)(L - i));
				const double s0 = __dadd_rn(__dadd_rn(M, EXv), tEC), s1 = __dadd_rn(s0, ccv);
				vw_cand(best, i < L ? s1 : s0, j, i, notCorner);
				if(j == K) {
					double s2 = __dadd_rn(__dadd_rn(I, tKIM), tEC);
					if(i < L) s2 = __dadd_rn(s2, ccv);
					vw_cand(best, s2, K + 1, i, notCorner);
				}
			}
			return dM | ((uI < uM ? 1 : 0) << 2) | ((lD < lM ? 1 : 0) << 3);
		};
		auto corner = [&](int i, int j, double M, double I, double D) {
			bool later = false, near = false;
			for(int r2 = r + 1; r2 < nR; ++r2) {
				const HuRegion& g2 = sreg[r2];
				if(reg_contains(g2, i, j)) later = true;
				if(i >= g2.i0 - 1 && i <= g2.i1 && j >= g2.j0 - 1 && j <= g2.j1) near = true;
			}
			if(near) { /* the corner block's origin is read from LDS here, on the rare path, instead of living in four registers for the whole phase */
				const HuRegion& gc = sreg[r];
				if(i >= gc.ci0 && j >= gc.cj0) {
					const int64_t idx = (gc.coff + (int64_t)(j - gc.cj0) * (gc.i1 - gc.ci0 + 1) + (i - gc.ci0)) * 3;
					scr[idx] = M; scr[idx + 1] = I; scr[idx + 2] = D;
				}
			}
			if(!later) {
				const double ccv = __dmul_rn(tCC, (double)(L - i));
				double sv = __dadd_rn(__dadd_rn(M, db.exitC[j]), tEC);
				if(i < L) sv = __dadd_rn(sv, ccv);
				vw_cand(best, sv, j, i, true);
				if(j == K) {
					double s2 = __dadd_rn(__dadd_rn(I, tKIM), tEC);
					if(i < L) s2 = __dadd_rn(s2, ccv);
					vw_cand(best, s2, K + 1, i, true);
				}
			}
		};
		if(g.band && ni <= 64) {
			/* ---- band phase: one row per lane, anti-diagonal t: column j0 + t - lane */
			const int i = g.i0 + lane;
			const bool row = lane < ni;
			const int b = row ? c_sym_map[(int) x[i - 1] & 127] : 0;
			const double bcol = vit_bcol(ctx, i);
			const double* __restrict__ emb = db.EMt + (size_t) b * K1;
			const double* __restrict__ eib = db.EIt + (size_t) b * K1;
			double pm = INFINITY, pi = INFINITY, pd = INFINITY;            /* (i, column - 1): starts as (i, j0 - 1) */
			double eUm = INFINITY, eUi = INFINITY, eLm = INFINITY, eLd = INFINITY, tmp;
			const int jU = (i - g.from) + g.nDel + g.start, jL = (i - g.from) - g.nIns + g.start;
			if(row) {
				look(r, i, g.j0 - 1, pm, pi, pd);
				if(jU >= g.j0 && jU <= g.j1) look(r, i - 1, jU, eUm, eUi, tmp);
				if(jL >= g.j0 && jL <= g.j1) look(r, i, jL - 1, eLm, tmp, eLd);
			}
			double fm, fi, fd;                                             /* (i - 1, column) of this step's column */
			double gm, gi, gd;                                             /* the same one step ago = (i - 1, column - 1) */
			look(r, i - 1, g.j0 - 1, gm, gi, gd);
			for(int t = 0; t <= ni + nj - 2; ++t) {
				const int j = g.j0 + t - lane;
				const bool cell = row && j >= g.j0 && j <= g.j1;
				const int dist = (i - g.from) - (j - g.start);
				const bool in = cell && dist <= g.nIns && dist >= -g.nDel;
				/* the triple of the lane above: its (M, I, D) at its previous column = this lane's column */
				double hm = INFINITY, hi2 = INFINITY, hd = INFINITY;
				if(lane == 0 && j >= g.j0 - 1 && j <= g.j1) {
					if(useHalo) { hm = halo[j - g.j0 + 1]; hi2 = halo[haloW + j - g.j0 + 1]; hd = halo[2 * haloW + j - g.j0 + 1]; }
					else look(r, g.i0 - 1, j, hm, hi2, hd);
				}
				fm = dpp_wave_shr1(pm, hm); fi = dpp_wave_shr1(pi, hi2); fd = dpp_wave_shr1(pd, hd);
				double M = INFINITY, I = INFINITY, D = INFINITY;
				if(in) {
					const double* tp = db.Tt + (j - 1);
					const double T0 = tp[0], T3 = tp[3 * K1], T5 = tp[5 * K1], T2 = tp[2 * K1], T6 = tp[6 * K1];
					const double J1 = tp[1 * K1 + 1], J4 = tp[4 * K1 + 1];
					const double ENv = db.entryC[j], EXv = db.exitC[j];
					/* up: above the band -> looked up at the start; left: left of the band likewise */
					const double mU = dist - 1 >= -g.nDel || lane == 0 ? fm : eUm, iU = dist - 1 >= -g.nDel || lane == 0 ? fi : eUi;
					const double mL = dist + 1 <= g.nIns || j == g.j0 ? pm : eLm, dL = dist + 1 <= g.nIns || j == g.j0 ? pd : eLd;
					const double pB = bcol + ENv;
					const double pM = gm + T0, pI = gi + T3, pD = gd + T5;
					double bst = fmin(pM, fmin(pI, pD));
					if(g.withB) bst = fmin(pB, bst);
					const double uM = mU + J1, uI = iU + J4;
					const double lM = mL + T2, lD = dL + T6;
					M = emb[j] + bst;
					I = eib[j] + fmin(uM, uI);
					D = (j > 1 && j < K) ? fmin(lM, lD) : INFINITY;
					const int by = finish(i, j, M, I, pB, pM, pI, pD, uM, uI, lM, lD, EXv);
					dcs[g.doff + (int64_t) t * 64 + lane] = (uint8_t) by;
					if(i >= nearI && j >= nearJ) corner(i, j, M, I, D);
				}
				/* the lane's (i, column) becomes (i, column - 1); a cell outside the band leaves what a look-up of it
				 * would have given (it is only ever read through eU / eL, never through these registers) */
				gm = fm; gi = fi; gd = fd;
				if(cell) { pm = M; pi = I; pd = D; }
			}
		}
		else {
			/* ---- rectangular phase: RPL rows per lane, step t: column j0 + t - lane */
			const int nL = (ni + RPL - 1) / RPL;
			const int ib = g.i0 + lane * RPL;
			int bk[RPL];
			double pm[RPL], pi[RPL], pd[RPL];                              /* (i_k, column - 1) */
#pragma unroll
			for(int k = 0; k < RPL; ++k) {
				const int i = ib + k;
				const bool row = lane * RPL + k < ni;
				bk[k] = row ? c_sym_map[(int) x[i - 1] & 127] : 0;
				pm[k] = pi[k] = pd[k] = INFINITY;
				if(row) look(r, i, g.j0 - 1, pm[k], pi[k], pd[k]);
			}
			double gm, gi, gd;                                             /* (ib - 1, column - 1) */
			look(r, ib - 1, g.j0 - 1, gm, gi, gd);
			/* profile values of the column this lane computes next */
			double nT0 = 0, nT3 = 0, nT5 = 0, nT2 = 0, nT6 = 0, nJ1 = 0, nJ4 = 0, nEN = 0, nEX = 0, nEM[RPL], nEI[RPL];
#pragma unroll
			for(int k = 0; k < RPL; ++k) nEM[k] = nEI[k] = 0;
			auto fetch = [&](int jc) {
				const double* tp = db.Tt + (jc - 1);
				nT0 = tp[0]; nT3 = tp[3 * K1]; nT5 = tp[5 * K1]; nT2 = tp[2 * K1]; nT6 = tp[6 * K1];
				nJ1 = tp[1 * K1 + 1]; nJ4 = tp[4 * K1 + 1];
				nEN = db.entryC[jc]; nEX = db.exitC[jc];
#pragma unroll
				for(int k = 0; k < RPL; ++k) { nEM[k] = db.EMt[(size_t) bk[k] * K1 + jc]; nEI[k] = db.EIt[(size_t) bk[k] * K1 + jc]; }
			};
			if(lane == 0) fetch(g.j0);
			for(int t = 0; t <= nj + nL - 2; ++t) {
				const int j = g.j0 + t - lane;
				const bool col = lane < nL && j >= g.j0 && j <= g.j1;
				const double T0 = nT0, T3 = nT3, T5 = nT5, T2 = nT2, T6 = nT6, J1 = nJ1, J4 = nJ4, ENv = nEN, EXv = nEX;
				double (&EMv)[RPL] = nEM; double (&EIv)[RPL] = nEI;
				double hm = INFINITY, hi2 = INFINITY, hd = INFINITY;
				if(lane == 0 && j >= g.j0 - 1 && j <= g.j1) {
					if(useHalo) { hm = halo[j - g.j0 + 1]; hi2 = halo[haloW + j - g.j0 + 1]; hd = halo[2 * haloW + j - g.j0 + 1]; }
					else look(r, g.i0 - 1, j, hm, hi2, hd);
				}
				/* (ib - 1, j): last row of the lane above at ITS previous column */
				const double fm = dpp_wave_shr1(pm[RPL - 1], hm), fi = dpp_wave_shr1(pi[RPL - 1], hi2), fd = dpp_wave_shr1(pd[RPL - 1], hd);
				if(col) {
					dword_t word = 0;
					double dm = gm, di = gi, dd = gd;          /* diagonal neighbour of row k: (i_k - 1, j - 1) */
					double um = fm, ui = fi;                    /* upper neighbour of row k: (i_k - 1, j) */
#pragma unroll
					for(int k = 0; k < RPL; ++k) {
						const int i = ib + k;
						const double om = pm[k], oi = pi[k], od = pd[k];   /* (i_k, j - 1): left neighbour, and the next row's diagonal */
						const bool in = lane * RPL + k < ni;
						const double mU = um, iU = ui, mL = om, dL = od, dM_ = dm, dI_ = di, dD_ = dd;
						double M = INFINITY, I = INFINITY, D = INFINITY;
						if(in) {
							const double pB = vit_bcol(ctx, i) + ENv;
							const double pM = dM_ + T0, pI = dI_ + T3, pD = dD_ + T5;
							double bst = fmin(pM, fmin(pI, pD));
							if(g.withB) bst = fmin(pB, bst);
							const double uM = mU + J1, uI = iU + J4;
							const double lM = mL + T2, lD = dL + T6;
							M = EMv[k] + bst;
							I = EIv[k] + fmin(uM, uI);
							D = (j > 1 && j < K) ? fmin(lM, lD) : INFINITY;
							const int by = finish(i, j, M, I, pB, pM, pI, pD, uM, uI, lM, lD, EXv);
							word |= (dword_t) by << (8 * k);
						}
						if(lane * RPL + k < ni) { pm[k] = M; pi[k] = I; pd[k] = D; }
						dm = om; di = oi; dd = od;
						um = M; ui = I;
					}
					if(DIAG != 1) *reinterpret_cast<dword_t*>(dcs + g.doff + ((int64_t) t * 64 + lane) * RPL) = word;
					else if(word == 0x7fffffffu) dcs[0] = 1;
					if(j >= nearJ && ib + RPL - 1 >= nearI) {
#pragma unroll
						for(int k = 0; k < RPL; ++k) if(lane * RPL + k < ni && ib + k >= nearI) corner(ib + k, j, pm[k], pi[k], pd[k]);
					}
				}
				gm = fm; gi = fi; gd = fd;
				/* the next column's profile values are requested when this column's are dead: their registers are
				 * reused, and the other waves of the SIMD cover the wait (three or four fit instead of two) */
				if(DIAG != 2) { if(lane < nL && j + 1 >= g.j0 && j + 1 <= g.j1) fetch(j + 1); }
			}
		}
		__syncthreads();   /* values filed for later phases are looked up by other lanes */
	}
	for(int m = 32; m > 0; m >>= 1) {
		const double os = __shfl_xor(best.s, m); const uint32_t ok = (uint32_t) __shfl_xor((int) best.key, m);
		if(os < best.s || (os == best.s && ok < best.key)) { best.s = os; best.key = ok; }
	}
	if(lane != 0) return;
	HuVitOut o;
	o.minScore = best.s; o.alnEnd = (int)(best.key >> 16) + 1; o.alnTo = (int)(best.key & 0xffffu); o.alnStart = o.alnFrom = 0; o.traceLen = -1; o.status = HU_READ_NEEDS_FULL;
	outs[s] = o;
}

/* traceback on the decision bytes, one lane per sequence */
__global__ __launch_bounds__(64) void k_viterbi_trace_dec(HuDbDev db, const HuReadDesc* __restrict__ descs, const uint8_t* __restrict__ dec,
		char* __restrict__ traces, HuVitOut* __restrict__ outs, int nSeq, int forceRedo, int rpl) {
	const int s = blockIdx.x * 64 + threadIdx.x;
	if(s >= nSeq) return;
	HuVitOut o = outs[s];
	if(o.traceLen != -1 || o.status == HU_READ_NEEDS_VALUES) return; /* invalid read, or left to the value-filing kernels by the fill kernel */
	const HuReadDesc& rd = descs[s];
	const int K = db.K, R = rd.nRegions;
	const double bestS = o.minScore; const int bestCol = o.alnEnd, bestRow = o.alnTo;
	o.traceLen = 0; o.alnStart = o.alnEnd = o.alnFrom = o.alnTo = 0;
	if(!(bestS < INFINITY)) { o.status = HU_READ_NEEDS_FULL; outs[s] = o; return; }
	const uint8_t* dcs = dec + rd.decOff;
	char* tr = traces + rd.traceOff;
	int n = 0;
	char st = bestCol <= K ? 'M' : 'I';
	int i = bestRow, j = bestCol <= K ? bestCol : K;
	o.alnEnd = j; o.alnTo = bestRow;
	tr[n++] = 'E';
	bool redo = forceRedo != 0; /* test hook: exercise the redo pass */
	while(i >= 1 && j >= 0 && !redo) {
		tr[n++] = st;
		if(st != 'M' && st != 'I' && st != 'D') break;
		int rc = -1;
		for(int r = R - 1; r >= 0; --r) if(reg_contains(rd.reg[r], i, j)) { rc = r; break; }
		const int pi = st == 'D' ? i : i - 1, pj = st == 'I' ? j : j - 1;
		if(rc < 0) { redo = true; break; }
		for(int r2 = rc + 1; r2 < R; ++r2) if(reg_contains(rd.reg[r2], pi, pj)) redo = true;
		if(redo) break;
		const HuRegion& g = rd.reg[rc];
		const int ni = g.i1 - g.i0 + 1;
		int64_t at;                                   /* the layout of the fill kernel that ran: rpl = 0 anti-diagonals of a workgroup, */
		if(rpl == 0) at = (int64_t)((i - g.i0) + (j - g.j0)) * ((ni + 15) & ~15) + (i - g.i0);      /* else steps of k_viterbi_wave<rpl> */
		else if(g.band && ni <= 64) at = (int64_t)((i - g.i0) + (j - g.j0)) * 64 + (i - g.i0);
		else { const int ln = (i - g.i0) / rpl, k = (i - g.i0) % rpl; at = ((int64_t)((j - g.j0) + ln) * 64 + ln) * rpl + k; }
		const int by = dcs[g.doff + at];
		if(st == 'M') { const int d = by & 3; st = d == 0 ? 'B' : d == 1 ? 'M' : d == 2 ? 'I' : 'D'; }
		else if(st == 'I') st = (by >> 2) & 1 ? 'I' : 'M';
		else st = (by >> 3) & 1 ? 'D' : 'M';
		i = pi; j = pj;
	}
	if(redo) { o.status = HU_READ_NEEDS_VALUES; o.traceLen = -1; o.minScore = bestS; o.alnEnd = bestCol; o.alnTo = bestRow; outs[s] = o; return; }
	o.alnStart = j + 1; o.alnFrom = i + 1;
	if(tr[n - 1] != 'B') tr[n++] = 'B';
	for(int a = 0, b = n - 1; a < b; ++a, --b) { char t = tr[a]; tr[a] = tr[b]; tr[b] = t; }
	o.traceLen = n;
	o.status = (o.alnStart > 0 && o.alnFrom > 0) ? HU_READ_OK : HU_READ_INVALID;
	outs[s] = o;
}

/* getPaddingSeq(..., JUSTIFIED) of a non-empty insert (src/BandedHMMP7.cpp:1168-1178) */
__device__ inline void pad_justified(char* dst, int L, const char* ins, int n) {
	if(n >= L) {
		const int h0 = L / 2, h1 = L - L / 2; /* floor, ceil */
		for(int a = 0; a < h0; ++a) dst[a] = (char)(ins[a] | 0x20);
		for(int a = 0; a < h1; ++a) dst[h0 + a] = (char)(ins[n - h1 + a] | 0x20);
	}
	else { /* reference quirk: the tail repeats the FIRST ceil(n/2) characters */
		const int h0 = n / 2, h1 = n - n / 2;
		for(int a = 0; a < h0; ++a) dst[a] = (char)(ins[a] | 0x20);
		for(int a = 0; a < L - n; ++a) dst[h0 + a] = '-';
		for(int a = 0; a < h1; ++a) dst[h0 + L - n + a] = (char)(ins[a] | 0x20);
	}
}

struct HuAlnDev { int32_t seqStart, seqEnd, hmmStart, hmmEnd, csStart, csEnd, status, usedFull; double cost; };

/* one wave per sequence: row = '.' * csLen, then lane 0 replays the trace */
__global__ __launch_bounds__(64) void k_align_rows(HuDbDev db, const HuReadDesc* __restrict__ descs, const char* __restrict__ bases,
		const char* __restrict__ traces, const HuVitOut* __restrict__ vit, char* __restrict__ rows, HuAlnDev* __restrict__ alns) {
	const int s = blockIdx.x, lane = threadIdx.x;
	const HuReadDesc& rd = descs[s];
	const HuVitOut v = vit[s];
	char* row = rows + (size_t) s * db.csLen;
	for(int c = lane; c < db.csLen; c += 64) row[c] = '.';
	__syncthreads();
	if(lane != 0) return;
	HuAlnDev a;
	a.status = v.status; a.cost = v.minScore; a.usedFull = 0;
	a.seqStart = v.alnFrom; a.seqEnd = v.alnTo; a.hmmStart = v.alnStart; a.hmmEnd = v.alnEnd;
	a.csStart = a.csEnd = 0;
	if(v.status != HU_READ_OK) { alns[s] = a; return; }
	const char* __restrict__ x = bases + rd.baseOff;
	const char* __restrict__ tr = traces + rd.traceOff;
	const int Lr = rd.len, L = db.csLen;
	const int csStart = db.p2cs[v.alnStart], csEnd = db.p2cs[v.alnEnd];
	a.csStart = csStart; a.csEnd = csEnd;
	int pos = 0, j = 0, k = 0, insFrom = 0, insLen = 0;
	for(int t = 0; t < v.traceLen; ++t) {
		const char st = tr[t];
		if(st == 'B') {
			const int nN = v.alnFrom - 1, room = csStart - 1;
			const int cnt = nN < room ? nN : room;
			for(int q = 0; q < cnt; ++q) row[room - cnt + q] = x[nN - cnt + q];
			pos = room; j = v.alnFrom; k = v.alnStart;
		}
		else if(st == 'M') {
			if(k > 1 && t > 1) {
				const int gap = db.p2cs[k] - db.p2cs[k - 1] - 1;
				if(gap > 0) {
					if(insLen > 0) pad_justified(row + pos, gap, x + insFrom, insLen);
					else for(int q = 0; q < gap; ++q) row[pos + q] = '-';
					pos += gap;
				}
			}
			insLen = 0;
			row[pos++] = x[j - 1];
			j++; k++;
		}
		else if(st == 'I') {
			insFrom = j - 1; insLen = 0;
			while(t < v.traceLen && tr[t] == 'I') { insLen++; j++; t++; }
			t--;
		}
		else if(st == 'D') {
			if(k > 1) {
				const int gap = db.p2cs[k] - db.p2cs[k - 1] - 1;
				for(int q = 0; q < gap; ++q) row[pos + q] = '-';
				if(gap > 0) pos += gap;
			}
			row[pos++] = '-';
			k++;
		}
		else if(st == 'E') {
			const int nC = Lr - v.alnTo > 0 ? Lr - v.alnTo : 0, room = L - csEnd;
			const int cnt = nC < room ? nC : room;
			for(int q = 0; q < cnt; ++q) row[pos + q] = x[v.alnTo + q];
		}
	}
	alns[s] = a;
}

/* PE: row r <- merge(row r, row n + r) unless the orientation check fails */
__global__ __launch_bounds__(256) void k_merge_rows(HuDbDev db, int n, int ignoreOrient, char* __restrict__ rows, HuAlnDev* __restrict__ alns) {
	const int r = blockIdx.x;
	__shared__ int ok;
	HuAlnDev a = alns[r];
	const HuAlnDev b = alns[n + r];
	if(threadIdx.x == 0) {
		int st = a.status;
		if(a.status == HU_READ_OK) {
			if(b.status != HU_READ_OK) st = HU_READ_INVALID;
			else if(!ignoreOrient && !(a.csStart <= b.csStart && a.csEnd <= b.csEnd)) st = HU_READ_CHIMERA;
		}
		ok = (st == HU_READ_OK);
		if(ok) {
			if(b.seqStart < a.seqStart) a.seqStart = b.seqStart;
			if(b.seqEnd > a.seqEnd) a.seqEnd = b.seqEnd;
			if(b.hmmStart < a.hmmStart) a.hmmStart = b.hmmStart;
			if(b.hmmEnd > a.hmmEnd) a.hmmEnd = b.hmmEnd;
			if(b.csStart < a.csStart) a.csStart = b.csStart;
			if(b.csEnd > a.csEnd) a.csEnd = b.csEnd;
			a.cost = a.cost + b.cost;
			a.usedFull |= b.usedFull;
		}
		a.status = st;
		alns[r] = a;
	}
	__syncthreads();
	if(!ok) return;
	char* ra = rows + (size_t) r * db.csLen;
	const char* rb = rows + (size_t)(n + r) * db.csLen;
	for(int c = threadIdx.x; c < db.csLen; c += 256) if(ra[c] == '.' && rb[c] != '.') ra[c] = rb[c];
}

/* DigitalSeq codes of the aligned row + region + bit-planes for the seed scan (in scan order, see
 * HuDbDev::posCol) + the read's bitmap of quads that hold at least one of its bases.
 * rp layout: rp[((tile*WQ + q) * T + t) * 16 + p*4 + w], zero outside [csStart-1, csEnd-1]. */
/* slot: the read's place in the scan's tiling — reads sorted by the first column of their region (k_tile_keys + radix sort), so that the
 * sixteen reads of a tile cover the same few quads whatever order the reads came in (with uniform read starts a tile in read order
 * is the union of sixteen unrelated windows: 10.4 ms against 3.6 ms for the scan); quad bitmap and insert list stay indexed by read */
__device__ inline void planes_of_codes(const HuDbDev& db, const int8_t* __restrict__ cd, int start, int end, int r, int slot, int lane,
		uint32_t* __restrict__ rp, uint32_t* __restrict__ rq, int32_t* __restrict__ ins, uint2* __restrict__ rspan) {
	const int tile = slot / HU_READ_TILE, t = slot % HU_READ_TILE;
	const int nw32 = (db.WQ + 31) / 32;
	uint32_t* qbits = rq + (size_t) r * nw32;
	int32_t* il = ins + (size_t) r * (HU_MAX_INS + 1);   /* [0] = count, then (scan position << 2 | base code) */
	/* Positions >= QM*128 hold non-profile columns only: a read has a base there only where the alignment put an
	 * insert.  Up to HU_MAX_INS of them travel as a list (one node word each in the scan); a read with more keeps
	 * them in its bit-planes and the quads join the tile's quad list. */
	const int QM = (db.K + 127) / 128;
	int nIns = 0;
	for(int base = QM * 128; base < db.WQ * 128; base += 64) {
		const int c = db.posCol[base + lane];
		const bool in = c >= start && c <= end && cd[c >= 0 ? c : 0] >= 0 && c >= 0;
		nIns += __popcll(__ballot(in));
	}
	const bool dense = nIns > HU_MAX_INS;
	uint32_t acc = 0;
	int li = 0;
	uint32_t f1 = 0xffffu, l1 = 0, f2 = 0xffffu, l2 = 0; bool any1 = false, any2 = false;   /* first / last position with a base: profile block, non-profile block */
	for(int base = 0; base < db.WQ * 128; base += 64) {
		const int c = db.posCol[base + lane];
		const int8_t code = c >= 0 ? cd[c] : (int8_t) -2;
		const bool in = c >= start && c <= end && code >= 0;
		const int q = base / 128, w = (base % 128) / 32;
		const bool asList = q >= QM && !dense;
		unsigned long long b0 = __ballot(in && (code & 1)), b1 = __ballot(in && (code & 2)), bv = __ballot(in);
		if(lane == 0 && bv) {
			const uint32_t lo = (uint32_t) base + (uint32_t)(__ffsll((long long) bv) - 1), hi = (uint32_t) base + 63u - (uint32_t) __clzll((long long) bv);
			if(q < QM) { if(!any1) { f1 = lo; any1 = true; } l1 = hi; }
			else { if(!any2) { f2 = lo; any2 = true; } l2 = hi; }
		}
		if(lane == 0) {
			if(asList) {
				unsigned long long m = bv;
				while(m) { const int bit = __ffsll((long long) m) - 1; m &= m - 1; il[1 + li++] = ((base + bit) << 2) | (int)(((b0 >> bit) & 1) | (((b1 >> bit) & 1) << 1)); }
				b0 = b1 = bv = 0;
			}
			uint32_t* dst = rp + (((size_t) tile * db.WQ + q) * HU_READ_TILE + t) * 16;
			dst[0 + w] = (uint32_t) b0; dst[0 + w + 1] = (uint32_t)(b0 >> 32);
			dst[4 + w] = (uint32_t) b1; dst[4 + w + 1] = (uint32_t)(b1 >> 32);
			dst[8 + w] = (uint32_t) bv; dst[8 + w + 1] = (uint32_t)(bv >> 32);
			if(bv) acc |= 1u << (q & 31);
			if((q & 31) == 31 && w == 2) { qbits[q >> 5] = acc; acc = 0; }
		}
	}
	if(lane == 0) {
		if(db.WQ & 31) qbits[(db.WQ - 1) >> 5] = acc; il[0] = li;
		rspan[r] = make_uint2((any1 ? f1 : 0xffffu) | ((any1 ? l1 : 0u) << 16), (any2 ? f2 : 0xffffu) | ((any2 ? l2 : 0u) << 16));
	}
}

__global__ __launch_bounds__(64) void k_encode_rows(HuDbDev db, const char* __restrict__ rows, HuAlnDev* __restrict__ alns,
		int8_t* __restrict__ codes, int32_t* __restrict__ rstart, int32_t* __restrict__ rend, uint32_t* __restrict__ rp, uint32_t* __restrict__ rq,
		int32_t* __restrict__ ins, const int32_t* __restrict__ readSlot, uint2* __restrict__ rspan) {
	const int r = blockIdx.x, lane = threadIdx.x;
	const HuAlnDev a = alns[r];
	const char* row = rows + (size_t) r * db.csLen;
	int8_t* cd = codes + (size_t) r * db.csLen;
	bool ok = a.status == HU_READ_OK;
	/* a read whose region leaves the resident message window cannot be placed: marked per read, with an empty region, so that
	 * the later stages never index the messages with it */
	if(ok && (a.csStart - 1 < db.winStart || a.csEnd - 1 >= db.winStart + db.winLen)) {
		ok = false;
		if(lane == 0) alns[r].status = HU_READ_OUT_OF_WINDOW;
	}
	/* the empty region of a read that is not placed sits at the first resident column: the later stages form message
	 * addresses from a region's start even when it has no column */
	const int start = ok ? a.csStart - 1 : (int) db.winStart, end = ok ? a.csEnd - 1 : (int) db.winStart - 1;
	if(lane == 0) { rstart[r] = start; rend[r] = end; }
	for(int c = lane; c < db.csLen; c += 64) {
		char ch = row[c];
		if(ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
		cd[c] = c_sym_map[(int) ch & 127];
	}
	__syncthreads();
	planes_of_codes(db, cd, start, end, r, readSlot[r], lane, rp, rq, ins, rspan);
}

/* same, when the caller supplies DigitalSeq codes directly (hu_batch_set_aligned) */
__global__ __launch_bounds__(64) void k_planes_from_codes(HuDbDev db, const int8_t* __restrict__ codes,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, uint32_t* __restrict__ rp, uint32_t* __restrict__ rq,
		int32_t* __restrict__ ins, const int32_t* __restrict__ readSlot, uint2* __restrict__ rspan) {
	const int r = blockIdx.x, lane = threadIdx.x;
	planes_of_codes(db, codes + (size_t) r * db.csLen, rstart[r], rend[r], r, readSlot[r], lane, rp, rq, ins, rspan);
}

/* sort keys of the tiling: first column of the region, reads without one last; vals = read index */
__global__ void k_tile_keys(int n, const HuAlnDev* __restrict__ alns, const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend,
		uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
	const int r = blockIdx.x * 256 + threadIdx.x;
	if(r >= n) return;
	uint32_t k;
	if(alns) k = alns[r].status == HU_READ_OK ? (uint32_t) alns[r].csStart : 0x7fffffffu;
	else k = rend[r] >= rstart[r] ? (uint32_t) rstart[r] + 1u : 0x7fffffffu;
	key[r] = k; val[r] = (uint32_t) r;
}
/* slotRead[slot] = read (-1 past n), readSlot[read] = slot */
__global__ void k_tile_slots(int n, int nSlots, const uint32_t* __restrict__ sortedRead, int32_t* __restrict__ slotRead, int32_t* __restrict__ readSlot) {
	const int s = blockIdx.x * 256 + threadIdx.x;
	if(s >= nSlots) return;
	if(s < n) { const int r = (int) sortedRead[s]; slotRead[s] = r; readSlot[r] = s; }
	else slotRead[s] = -1;
}

/* per scan tile: the quads in which any of its reads has a base -> tileQ[tile][0] = count, [1..] = quads; and the
 * reads' insert lists as one list, tileIns[tile][0] = count, [1..] = read t << 24 | (scan position << 2 | base) */
__global__ __launch_bounds__(64) void k_tile_lists(HuDbDev db, int n, const uint32_t* __restrict__ rq, const int32_t* __restrict__ ins,
		int32_t* __restrict__ tileQ, int32_t* __restrict__ tileIns, const int32_t* __restrict__ slotRead, const uint2* __restrict__ rspan, uint2* __restrict__ tileSpan) {
	const int tile = blockIdx.x, lane = threadIdx.x;
	const int nw32 = (db.WQ + 31) / 32;
	int32_t* out = tileQ + (size_t) tile * (db.WQ + 1);
	int cnt = 0;
	for(int w = 0; w < nw32; ++w) { /* WQ <= 512: at most 16 words; lane 0 writes the compacted list */
		uint32_t m = 0;
		if(lane < HU_READ_TILE) { const int r = slotRead[tile * HU_READ_TILE + lane]; if(r >= 0) m = rq[(size_t) r * nw32 + w]; }
		for(int s = 32; s > 0; s >>= 1) m |= __shfl_xor(m, s);
		if(lane == 0) while(m) { const int b = __ffs(m) - 1; m &= m - 1; out[1 + cnt++] = w * 32 + b; }
	}
	if(lane == 0) {
		out[0] = cnt;
		int32_t* til = tileIns + (size_t) tile * (HU_READ_TILE * HU_MAX_INS + 1);
		int ne = 0;
		for(int t = 0; t < HU_READ_TILE; ++t) {
			const int r = slotRead[tile * HU_READ_TILE + t];
			if(r < 0) continue;
			const int32_t* il = ins + (size_t) r * (HU_MAX_INS + 1);
			for(int e = 0; e < il[0]; ++e) til[1 + ne++] = (t << 24) | il[1 + e];
		}
		til[0] = ne;
		/* the union of the reads' position intervals, block by block */
		uint32_t f1 = 0xffffu, l1 = 0, f2 = 0xffffu, l2 = 0; bool a1 = false, a2 = false;
		for(int t = 0; t < HU_READ_TILE; ++t) {
			const int r = slotRead[tile * HU_READ_TILE + t];
			if(r < 0) continue;
			const uint2 s2 = rspan[r];
			if((s2.x & 0xffffu) <= (s2.x >> 16)) { f1 = a1 ? min(f1, s2.x & 0xffffu) : (s2.x & 0xffffu); l1 = a1 ? max(l1, s2.x >> 16) : (s2.x >> 16); a1 = true; }
			if((s2.y & 0xffffu) <= (s2.y >> 16)) { f2 = a2 ? min(f2, s2.y & 0xffffu) : (s2.y & 0xffffu); l2 = a2 ? max(l2, s2.y >> 16) : (s2.y >> 16); a2 = true; }
		}
		tileSpan[tile] = make_uint2((a1 ? f1 : 0xffffu) | ((a1 ? l1 : 0u) << 16), (a2 ? f2 : 0xffffu) | ((a2 ? l2 : 0u) << 16));
	}
}
