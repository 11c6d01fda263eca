// Register-resident variants of k_estimate / k_place (gfx950, wave64).  Included by hu_engine.hip.
//
// One workgroup of 4 waves per (read, seed) resp. candidate.  Thread t owns the sites t, t + 256, ...
// (SPT of them) and keeps their two messages IN THE EIGENBASIS of the substitution model
// (a = U^-1 e, 8 doubles per site) in registers for the whole kernel: every message byte crosses HBM
// exactly once (the streaming kernels of hu_kern_sep.h re-read them once per pass / per sweep: PMC
// showed 76.6 GB and 91.9 GB per launch against 36 GB and 18 GB of algorithmic bytes).  Sums over the
// sites are DPP wave sums combined across the four waves through 32 bytes of LDS.
#pragma once
#include "hu_kern_sep.h"

#define HU_BLK_THREADS 256

/* orders a wave's own LDS writes before its own later LDS reads (a wave's LDS queue is in order; this waits for
 * the returns and keeps the compiler from moving accesses across) */
__device__ inline void lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* sum over the workgroup, identical in every thread; `red` is a [2][4] LDS scratch, `phase` toggles */
template<int NW = 4>
__device__ inline double block_sum(double v, double* red, int& phase) {
	v = wave_sum_uniform(v);
	double* r = red + (phase & 1) * NW;
	if((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
	lds_barrier();
	double s = r[0];
#pragma unroll
	for(int w = 1; w < NW; ++w) s += r[w];
	phase ^= 1;
	return s;
}
template<int NW = 4>
__device__ inline long long block_sum_ll(long long v, long long* red, int& phase) {
	for(int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
	long long* r = red + (phase & 1) * NW;
	if((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
	lds_barrier();
	long long s = r[0];
#pragma unroll
	for(int w = 1; w < NW; ++w) s += r[w];
	phase ^= 1;
	return s;
}

template<int SPT, int THREADS>
__global__ __launch_bounds__(THREADS) void k_estimate_blk(HuDbDev db, HuModelDev mdl, const int8_t* __restrict__ codes,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, const uint32_t* __restrict__ parDN,
		const int32_t* __restrict__ seedCnt, const int32_t* __restrict__ seedId, const uint32_t* __restrict__ seedDN,
		int weighted, HuEstOut* __restrict__ out) {
	__shared__ double red[2 * THREADS / 64];
	__shared__ long long redl[2 * THREADS / 64];
	constexpr int NW = THREADS / 64;
	const uint32_t slot = db.wideList ? db.wideList[blockIdx.x] : blockIdx.x;
	const int read = slot / HU_MAX_SEEDS, s = slot % HU_MAX_SEEDS, tid = threadIdx.x;
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
	const int start = rstart[read], end = rend[read], n = end - start + 1;
	const int8_t* __restrict__ cd = codes + (size_t) read * db.csLen + start;
	const int64_t sOff = (int64_t) u * db.winLen + (start - db.winStart);
	const int piMax = argmax4d(mdl.logpi);
	double piw[4];
	{ double mx = max4d(mdl.logpi), sm; for(int i = 0; i < 4; ++i) piw[i] = exp(mdl.logpi[i] - mx); sm = (piw[0] + piw[2]) + (piw[1] + piw[3]); for(int i = 0; i < 4; ++i) piw[i] /= sm; }
	/* one pass over HBM: z_i = (P(wur) e^U)_i (P(wvr) e^V)_i of the thread's sites stays in registers */
	double z[SPT][4]; int bb[SPT]; long long ksum = 0;
	{
		double eU[SPT][4], eV[SPT][4];
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			const int j = tid + THREADS * t;
			const int jj = j < n ? j : 0;
			load4(db.up + (sOff + jj) * 4, eU[t]); load4(db.down + (sOff + jj) * 4, eV[t]);
			bb[t] = cd[jj];
			if(j < n) ksum += (long long) db.upK[sOff + jj] + (long long) db.downK[sOff + jj];
		}
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			double c[4];                                   /* messages arrive in the eigenbasis */
			conv_eig(mdl, Eu, eU[t], z[t]);
			conv_eig(mdl, Ev, eV[t], c);
			for(int i = 0; i < 4; ++i) z[t][i] *= c[i];
		}
	}
	int phase = 0;
	double dsum = 0, nsum = 0;
#pragma unroll
	for(int t = 0; t < SPT; ++t) {
		if(tid + THREADS * t >= n) continue;
		const int b = bb[t];
		const int b1 = argmax4_tied_lin(z[t]), b2 = b >= 0 ? b : piMax;
		if(!weighted) { if(b1 != b2) dsum += 1; }
		else {
			const double w1 = sel4(z[t], b1) / ((z[t][0] + z[t][2]) + (z[t][1] + z[t][3]));
			const double w2 = b >= 0 ? 1.0 : piw[b2];
			if(b1 != b2) dsum += w1 * w2;
			nsum += w1 * w2;
		}
	}
	dsum = block_sum<NW>(dsum, red, phase);
	double wnr;
	if(!weighted) wnr = dsum / (double) n;
	else { nsum = block_sum<NW>(nsum, red, phase); wnr = dsum / nsum; }
	double En[4], Ppi[4];
#pragma unroll
	for(int k = 0; k < 4; ++k) En[k] = exp(mdl.lam[k] * wnr);
	{ double a[4]; to_eig(mdl, mdl.pi, a); if(wnr == 0) { for(int i = 0; i < 4; ++i) Ppi[i] = mdl.pi[i]; } else conv_eig(mdl, En, a, Ppi); }
	double ll = 0;
#pragma unroll
	for(int t = 0; t < SPT; ++t) {
		if(tid + THREADS * t >= n) continue;
		const int b = bb[t];
		double c[4];
		if(b >= 0) {
			if(wnr == 0) { for(int i = 0; i < 4; ++i) c[i] = i == b ? 1.0 : 0.0; }
			else { double a[4]; for(int k = 0; k < 4; ++k) a[k] = mdl.U1[k*4+b]; conv_eig(mdl, En, a, c); }
		}
		else for(int i = 0; i < 4; ++i) c[i] = Ppi[i];
		ll += log((mdl.pi[0] * z[t][0] * c[0] + mdl.pi[2] * z[t][2] * c[2]) + (mdl.pi[1] * z[t][1] * c[1] + mdl.pi[3] * z[t][3] * c[3]));
	}
	ll = block_sum<NW>(ll, red, phase);
	int lphase = 0;
	const long long kt = block_sum_ll<NW>(ksum, redl, lphase);
	ll += (double) kt * HU_LN2;
	if(tid == 0) { HuEstOut o; o.ratio = ratio; o.wnr = wnr; o.loglik = ll; out[(size_t) read * HU_MAX_SEEDS + s] = o; }
}

/* ------------------------------------------------------------------------------------------------
 * k_estimate_prod: estimateSeq (src/PhyloTreeUnrooted.cpp:849-877) with the per-site logarithm removed.
 *
 * sum_j log x_j = log prod_j x_j: each site contributes its mantissa to a running product and its binary
 * exponent to an integer sum (v_frexp_mant_f64, v_frexp_exp_i32_f64, v_mul_f64, v_add_i32: 4 instructions
 * instead of the ~50 of log()); products are renormalised per thread, multiplied across the wave on the DPP
 * network and across the waves through LDS, and ONE log() per (read, seed) closes the sum, together with the
 * integer exponents of the packed messages (k of 2^-k).  Relative error of the product of R mantissas is
 * <= R 2^-53, i.e. an absolute error of ~1.5e-13 on a log-likelihood of ~-1e3.
 * Pass 2 is table-driven: x_j = sum_i Q^b_i z_i with Q^b_i = pi_i (P(wnr) e_b)_i for a base b and
 * pi_i (P(wnr) pi)_i for a gap; every wave builds its own copy of the 5 x 4 table (no workgroup barrier).
 * Component 0 of a message in the eigenbasis is pi . e and U(i,0) = 1, lam_0 = 0 (hu_model_prepare):
 * (P(t) e)_i = a_0 + sum_{m>0} U_im exp(lam_m t) a_m. */
template<int CTRL>
__device__ inline double dpp_mul_full(double v) {
	const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
	const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
	return v * __hiloint2double(hi, lo);
}
/* product over the 64 lanes (in every lane): butterfly of xor-shuffles where DPP has no pattern */
__device__ inline double wave_prod(double v) {
	v = dpp_mul_full<0xB1>(v);   /* quad_perm:[1,0,3,2] */
	v = dpp_mul_full<0x4E>(v);   /* quad_perm:[2,3,0,1] */
	v = dpp_mul_full<0x141>(v);  /* row_half_mirror */
	v = dpp_mul_full<0x140>(v);  /* row_mirror */
	v *= __shfl_xor(v, 16);
	v *= __shfl_xor(v, 32);
	return v;
}
__device__ inline int wave_sum_i32(int v) {
	for(int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
	return v;
}

/* Position in the node-sorted list for workgroup b: blocks are dealt round-robin over the 8 XCDs (observed, not promised),
 * so b % 8 names the XCD and XCD x walks the CONTIGUOUS eighth x of the list: workgroups that want the same node's
 * messages share one L2 instead of meeting in the Infinity Cache.  xmap = 0: position b. */
__device__ inline uint32_t hu_xcd_pos(uint32_t b, uint32_t nb, int xmap) {
	if(!xmap) return b;
	const uint32_t per = nb >> 3;
	return b < per * 8 ? (b & 7u) * per + (b >> 3) : b;
}

template<int SPT, int NW, int OCC = 1>
__global__ __launch_bounds__(64 * NW, OCC) void k_estimate_prod(HuDbDev db, HuModelDev mdl, const int8_t* __restrict__ codes,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, const uint32_t* __restrict__ parDN,
		const int32_t* __restrict__ seedCnt, const int32_t* __restrict__ seedId, const uint32_t* __restrict__ seedDN,
		int weighted, HuEstOut* __restrict__ out, const uint32_t* __restrict__ order, int xmap = 0) {
	constexpr int THREADS = 64 * NW;
	__shared__ double redd[2 * NW];
	__shared__ int redi[2 * NW];
	__shared__ __attribute__((aligned(16))) double Qtab[NW][24];   /* [wave][0..3] = exp(lam_m wnr), [4 + b * 4 + i] = Q^b_i */
	/* `order`: the (read, seed) slots sorted by seed NODE, so that the workgroups reading one node's messages run
	 * together and all but the first find them in the Infinity Cache / L2 (reads of one sample share their seeds) */
	const uint32_t slot = order ? order[hu_xcd_pos(blockIdx.x, gridDim.x, xmap)] : blockIdx.x;
	const int read = slot / HU_MAX_SEEDS, s = slot % HU_MAX_SEEDS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if(s >= seedCnt[read] || hu_skip_width(db, rend[read] - rstart[read] + 1)) return;
	const int un = seedId[(size_t) read * HU_MAX_SEEDS + s];
	const int vn = db.parent[un];
	const uint32_t dn = seedDN[(size_t) read * HU_MAX_SEEDS + s];
	const uint32_t pv = parDN[(size_t) read * HU_MAX_SEEDS + s];      /* (d, N) against the parent's sequence */
	const double cDist = (double)(dn >> 16) / (double)(dn & 0xffffu);
	const double pDist = (double)(pv >> 16) / (double)(pv & 0xffffu);
	double ratio = cDist / (cDist + pDist);
	if(isnan(ratio)) ratio = 0.5;
	const double w0 = db.blen[un];
	const double wur = w0 * ratio, wvr = w0 - wur;
	/* the six exponentials exp(lam_m w_ur), exp(lam_m w_vr), m = 1..3: ONE exp() per wave, lane l < 6 evaluating the l-th of them,
	 * read back into scalar registers — every lane evaluating all six cost 5 x ~45 wave-instructions of the ~1,070 a wave executes */
	double Eu[3], Ev[3];
	{
		const int l6 = lane < 6 ? lane : 0, m3 = l6 % 3;
		const double e = exp((m3 == 0 ? mdl.lam[1] : m3 == 1 ? mdl.lam[2] : mdl.lam[3]) * (l6 < 3 ? wur : wvr));
		const int lo = __double2loint(e), hi = __double2hiint(e);
#pragma unroll
		for(int k = 0; k < 3; ++k) {
			Eu[k] = __hiloint2double(__builtin_amdgcn_readlane(hi, k), __builtin_amdgcn_readlane(lo, k));
			Ev[k] = __hiloint2double(__builtin_amdgcn_readlane(hi, 3 + k), __builtin_amdgcn_readlane(lo, 3 + k));
		}
	}
	const int start = rstart[read], end = rend[read], n = end - start + 1;
	const int8_t* __restrict__ cd = codes + (size_t) read * db.csLen + start;
	const int64_t sOff = (int64_t) un * db.winLen + (start - db.winStart);
	const int piMax = argmax4d(mdl.logpi);
	/* one pass over HBM: z_i = (P(wur) e^U)_i (P(wvr) e^V)_i of the thread's sites stays in registers */
	double z[SPT][4]; unsigned long long bop = 0; int ksum = 0;
	{
		double aU[SPT][4], aV[SPT][4];
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			const int j = tid + THREADS * t;
			const int jj = j < n ? j : 0;
			load4(db.up + (sOff + jj) * 4, aU[t]); load4(db.down + (sOff + jj) * 4, aV[t]);
			const int b = cd[jj];
			bop |= (unsigned long long)(b >= 0 ? b : 4) << (3 * t);
			if(j < n) ksum += db.upK[sOff + jj] + db.downK[sOff + jj];
		}
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			const double su1 = Eu[0] * aU[t][1], su2 = Eu[1] * aU[t][2], su3 = Eu[2] * aU[t][3];
			const double sv1 = Ev[0] * aV[t][1], sv2 = Ev[1] * aV[t][2], sv3 = Ev[2] * aV[t][3];
#pragma unroll
			for(int i = 0; i < 4; ++i) {
				const double cu = fmax(fma(mdl.U[i*4+3], su3, fma(mdl.U[i*4+2], su2, fma(mdl.U[i*4+1], su1, aU[t][0]))), 0.0);
				const double cv = fmax(fma(mdl.U[i*4+3], sv3, fma(mdl.U[i*4+2], sv2, fma(mdl.U[i*4+1], sv1, aV[t][0]))), 0.0);
				z[t][i] = cu * cv;
			}
		}
	}
	/* wnr: fraction of sites whose inferred state differs from the read's (method "unweighted"), or the
	 * weighted form of src/PhyloTreeUnrooted.cpp:1034-1052 */
	double wnr;
	if(!weighted) {
		int nd = 0;
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			const int bi = (int)((bop >> (3 * t)) & 7u);
			const int b1 = argmax4_tied_lin(z[t]), b2 = bi < 4 ? bi : piMax;
			nd += __popcll(__ballot(tid + THREADS * t < n && b1 != b2));
		}
		if(lane == 0) redi[wave] = nd;
		lds_barrier();
		int tot = redi[0];
#pragma unroll
		for(int w = 1; w < NW; ++w) tot += redi[w];
		wnr = (double) tot / (double) n;
	}
	else {
		double piw[4];
		{ double mx = max4d(mdl.logpi), sm; for(int i = 0; i < 4; ++i) piw[i] = exp(mdl.logpi[i] - mx); sm = (piw[0] + piw[2]) + (piw[1] + piw[3]); for(int i = 0; i < 4; ++i) piw[i] /= sm; }
		double dsum = 0, nsum = 0;
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			if(tid + THREADS * t >= n) continue;
			const int bi = (int)((bop >> (3 * t)) & 7u);
			const int b1 = argmax4_tied_lin(z[t]), b2 = bi < 4 ? bi : piMax;
			const double w1 = sel4(z[t], b1) / ((z[t][0] + z[t][2]) + (z[t][1] + z[t][3]));
			const double w2 = bi < 4 ? 1.0 : piw[b2];
			if(b1 != b2) dsum += w1 * w2;
			nsum += w1 * w2;
		}
		dsum = wave_sum(dsum); nsum = wave_sum(nsum);
		if(lane == 0) { redd[wave] = dsum; redd[NW + wave] = nsum; }
		lds_barrier();
		double a = redd[0], c = redd[NW];
#pragma unroll
		for(int w = 1; w < NW; ++w) { a += redd[w]; c += redd[NW + w]; }
		wnr = a / c;
		lds_barrier();   /* redd is reused below */
	}
	/* Q^b_i = pi_i (P(wnr) c^b)_i, c^b = e_b or pi; this wave's own copy */
	double* Q = Qtab[wave];
	if(lane < 3) Q[1 + lane] = exp((lane == 0 ? mdl.lam[1] : lane == 1 ? mdl.lam[2] : mdl.lam[3]) * wnr);
	lds_wave_sync();
	if(lane < 20) {
		const int b = lane >> 2, i = lane & 3;
		const double pii = sel4(mdl.pi, i);
		double c;
		if(wnr == 0) c = b < 4 ? (i == b ? 1.0 : 0.0) : pii;
		else {
			/* a = U^-1 c^b: column b of U1, or U1 . pi */
			double a0, a1, a2, a3;
			if(b < 4) { a0 = sel4(mdl.U1 + 0, b); a1 = sel4(mdl.U1 + 4, b); a2 = sel4(mdl.U1 + 8, b); a3 = sel4(mdl.U1 + 12, b); }
			else {
				a0 = (mdl.U1[0] * mdl.pi[0] + mdl.U1[1] * mdl.pi[1]) + (mdl.U1[2] * mdl.pi[2] + mdl.U1[3] * mdl.pi[3]);
				a1 = (mdl.U1[4] * mdl.pi[0] + mdl.U1[5] * mdl.pi[1]) + (mdl.U1[6] * mdl.pi[2] + mdl.U1[7] * mdl.pi[3]);
				a2 = (mdl.U1[8] * mdl.pi[0] + mdl.U1[9] * mdl.pi[1]) + (mdl.U1[10] * mdl.pi[2] + mdl.U1[11] * mdl.pi[3]);
				a3 = (mdl.U1[12] * mdl.pi[0] + mdl.U1[13] * mdl.pi[1]) + (mdl.U1[14] * mdl.pi[2] + mdl.U1[15] * mdl.pi[3]);
			}
			const double Ui1 = i == 0 ? mdl.U[1] : i == 1 ? mdl.U[5] : i == 2 ? mdl.U[9] : mdl.U[13];
			const double Ui2 = i == 0 ? mdl.U[2] : i == 1 ? mdl.U[6] : i == 2 ? mdl.U[10] : mdl.U[14];
			const double Ui3 = i == 0 ? mdl.U[3] : i == 1 ? mdl.U[7] : i == 2 ? mdl.U[11] : mdl.U[15];
			c = fmax(fma(Ui3, Q[3] * a3, fma(Ui2, Q[2] * a2, fma(Ui1, Q[1] * a1, a0))), 0.0);
		}
		Q[4 + b * 4 + i] = pii * c;
	}
	lds_wave_sync();
	/* log-likelihood: product of the mantissas, sum of the exponents */
	double mant = 1.0; int esum = ksum;
#pragma unroll
	for(int t = 0; t < SPT; ++t) {
		const double* q = Q + 4 + ((bop >> (3 * t)) & 7u) * 4;
		double x = fmax(fma(q[3], z[t][3], fma(q[2], z[t][2], fma(q[1], z[t][1], q[0] * z[t][0]))), 0.0);
		if(tid + THREADS * t >= n) x = 1.0;
		mant *= __builtin_amdgcn_frexp_mant(x);
		esum += x == 0.0 ? 0 : __builtin_amdgcn_frexp_exp(x);
	}
	esum += __builtin_amdgcn_frexp_exp(mant); mant = __builtin_amdgcn_frexp_mant(mant);   /* in [0.5, 1): 64 of them stay normal */
	mant = wave_prod(mant);
	esum = wave_sum_i32(esum);
	if(lane == 0) { redd[wave] = mant; redi[NW + wave] = esum; }
	lds_barrier();
	if(tid == 0) {
		double m = redd[0]; long long e = redi[NW];
#pragma unroll
		for(int w = 1; w < NW; ++w) { m *= redd[w]; e += redi[NW + w]; }
		HuEstOut o; o.ratio = ratio; o.wnr = wnr; o.loglik = log(m) + (double) e * HU_LN2;
		out[(size_t) read * HU_MAX_SEEDS + s] = o;
	}
}

/* ------------------------------------------------------------------------------------------------
 * k_place_blk: joint branch-length optimisation of one candidate placement per workgroup
 * (src/PhyloTreeUnrooted.cpp:800-847 outer loop, :749-798 EM), table-driven.
 *
 * With both messages of a site in the eigenbasis (a_U, a_V) and V orthonormal (U = Pi^-1/2 V, so that
 * sum_i pi_i U_im U_in = delta_mn), every per-site quantity of the two sweeps is a bilinear form:
 *   sweep (i)   rho = (a_U^T N^b a_V) / (sum_m G_mm a_Um a_Vm),   N^b_mn = W^b_mn G_mn
 *   sweep (ii)  rho = (a_V^T Z^b a_U) / (sum_m T^b_mm a_Vm),       Z^b_mk = sum_n T^b_mn C_mnk, T^b_mn = c^b_n G'_mn
 * b = the read's base at the site or 4 for a gap, G / G' the category-averaged products of exponentials of the
 * two branch lengths involved.  Component 0 of a message in the eigenbasis is pi . e (hu_model_prepare) and both
 * ratios are invariant under a scaling of either message, so each message is divided by its component 0 and only
 * the other three are kept: 6 doubles per site, register-resident for the whole optimisation together with the
 * per-site ratios — every message byte crosses HBM once (the streaming k_place re-reads them per sweep).
 * The five 4x4 tables are rebuilt per sweep by <= 100 threads; a site costs ~21 FMA + one division instead of
 * three 4x4 matvecs per rate category.  The EM step is p0 / (rho_j q0 + p0) summed over the sites: fma,
 * v_rcp_f64, Newton step(s), add. */
#define HU_RHO_SKIP 1e60         /* sentinel ratio of a site the EM skips (NaN ratio in the reference, padding) */
#define HU_TP 18                 /* doubles per table row block in LDS (16 + 2: rows of different b on different banks) */
#define HU_EXP_MEPS 0.99999000004999983333   /* exp(-1e-5) */
#define HU_EXP_PEPS 1.00001000005000016667   /* exp(+1e-5) */

__device__ inline double fast_div(double a, double b) { /* a / b to ~1 ulp for finite non-zero b; NaN for b = 0 */
	double y = __builtin_amdgcn_rcp(b);
	y = fma(fma(-b, y, 1.0), y, y);
	y = fma(fma(-b, y, 1.0), y, y);
	const double q = a * y;
	return fma(fma(-b, q, a), y, q);
}

/* exp / log as real calls: inlined, their ~20 polynomial coefficients are hoisted out of the candidate's outer loop
 * and sit in ~45 VGPRs for the whole kernel (64-bit literals cannot be VOP3 operands) — the difference between two
 * and four workgroups per CU.  They run a few times per EM call, never per site. */
__device__ __attribute__((noinline)) double hu_exp_call(double x) { return exp(x); }
__device__ __attribute__((noinline)) double hu_log_call(double x) { return log(x); }

/* EMV: Newton steps on v_rcp_f64 (2^-23 or better): 1 -> 2^-46 per term (a 1e-14 relative bias on the branch
 * length, eight orders below the 1e-6 bar), 2 -> full double precision.  RED: unused (0).
 * Measured and not kept (round 2): s_memtime stamps put a step of the 12-site kernel at 680 ticks of arithmetic + 378 of wave reduction and
 * exchange + 200 of tail.  Shortening the second half — row sums only on the DPP network, the 4 NW row sums handed over through LDS as a
 * broadcast read, the tail's factors computed before the exchange — changed nothing (8.80 against 8.86 ms), nor did the stage-by-stage
 * arithmetic below (8.90 against 8.87): a step lasts as long as the SLOWER of the two waves needs for its arithmetic beside whatever shares
 * its SIMD; the chain behind it is waiting for the partner, not instruction latency. */
template<int SPT, int NW, int EMV, int RED>
__device__ inline double em_branch_blk(const double (&rho)[SPT], int nvalidWave, double w0, double maxL,
		double* red, double* redc, int& phase, int& emIters, long long* st = nullptr) {
	/* st (diagnostic build only): s_memtime ticks of a step's arithmetic, wave reduction, exchange between the waves, tail */
	long long tl = st ? (long long) __builtin_amdgcn_s_memtime() : 0;
	auto stamp = [&](int i) { if(st) { const long long t = (long long) __builtin_amdgcn_s_memtime(); st[i] += t - tl; tl = t; } };
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	double q0 = hu_exp_call(-w0), p0 = 1 - q0, p = p0, q = q0, rc = 0;
	if(lane == 0) redc[(phase & 1) * NW + wave] = (double) nvalidWave;
	for(int it = 0; it < HU_MAX_ITER && p >= 0 && p <= 1; ++it) {
		double s = 0;
		const bool fast = q0 >= 1e-30 && p0 >= 1e-30; /* x = rho q0 + p0 in [1e-30, 1e60]; skipped sites add < 1e-30 */
		if(fast) {
			if(EMV == 3 && SPT % 4 == 0) { /* one reciprocal per four sites: 1/a + 1/b + 1/c + 1/d = ((a + b) cd + (c + d) ab) / (abcd);
				                             * a .. d in [1e-30, 1e60] (skipped sites: 1e60 q0 + p0), so the products stay in range.
				                             * Written STAGE BY STAGE over the groups of four (same operations, same order of the sum): left to
				                             * itself the compiler emits one group after the other, each a chain of ~8 dependent FP64
				                             * operations, and an in-order wave with one partner on its SIMD stalls at every link */
				constexpr int G = SPT / 4;
				double x[SPT], ab[G], cd[G], X[G], nm[G], y[G], w[G];
#pragma unroll
				for(int t = 0; t < SPT; ++t) x[t] = fma(rho[t], q0, p0);
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for(int g = 0; g < G; ++g) { ab[g] = x[4 * g] * x[4 * g + 1]; cd[g] = x[4 * g + 2] * x[4 * g + 3]; }
#pragma unroll
				for(int g = 0; g < G; ++g) { X[g] = ab[g] * cd[g]; x[4 * g] += x[4 * g + 1]; x[4 * g + 2] += x[4 * g + 3]; }
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for(int g = 0; g < G; ++g) { y[g] = __builtin_amdgcn_rcp(X[g]); nm[g] = x[4 * g + 2] * ab[g]; }
#pragma unroll
				for(int g = 0; g < G; ++g) nm[g] = fma(x[4 * g], cd[g], nm[g]);
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for(int g = 0; g < G; ++g) { w[g] = fma(-X[g], y[g], 2.0); nm[g] *= y[g]; }
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for(int g = 0; g < G; ++g) s = fma(nm[g], w[g], s);
			}
			else if(EMV == 4 && SPT % 2 == 0) { /* one reciprocal per two sites: 1/a + 1/b = (a + b) / (ab) */
#pragma unroll
				for(int t = 0; t < SPT; t += 2) {
					const double a = fma(rho[t], q0, p0), bb = fma(rho[t + 1], q0, p0);
					const double x = a * bb, y = __builtin_amdgcn_rcp(x);
					s = fma((a + bb) * y, fma(-x, y, 2.0), s);
				}
			}
			else {
#pragma unroll
				for(int t = 0; t < SPT; ++t) {
					const double x = fma(rho[t], q0, p0);
					const double y = __builtin_amdgcn_rcp(x);
					if(EMV == 1 || EMV == 3 || EMV == 4) s = fma(y, fma(-x, y, 2.0), s);
					else { const double y1 = fma(fma(-x, y, 1.0), y, y); s = fma(y1, fma(-x, y1, 2.0), s); }
				}
			}
		}
		else { /* degenerate branch lengths (p0 = 0, q0 = 0): the reference's expression as written */
#pragma unroll
			for(int t = 0; t < SPT; ++t) {
				const double r = rho[t];
				const double tt = p0 / fma(r, q0, p0);
				s += r == HU_RHO_SKIP ? 0.0 : tt;
			}
		}
		if(st) { if(s == 1.2345e-300) st[7] = 1; }   /* wait for the arithmetic */
		stamp(0);
		s = wave_sum_uniform(s);       /* (RED = 1 once chose a wave sum on two v_mfma_f64_16x16x4: measured slower than the DPP butterfly, removed in round 3) */
		stamp(1);
		double* r = red + (phase & 1) * NW;
		if(lane == 0) r[wave] = s;
		lds_barrier();
		s = r[0];
#pragma unroll
		for(int w = 1; w < NW; ++w) s += r[w];
		if(it == 0) {
			const double* c = redc + (phase & 1) * NW;
			double cnt = c[0];
#pragma unroll
			for(int w = 1; w < NW; ++w) cnt += c[w];
			rc = 1.0 / cnt;
		}
		phase ^= 1;
		if(st) { if(s == 1.2345e-300) st[7] = 1; }
		stamp(2);
		if(fast) s *= p0;
		p = s * rc; q = 1 - p;
		++emIters;
		if(q0 * HU_EXP_MEPS < q && q < q0 * HU_EXP_PEPS) { stamp(3); break; } /* |log q - log q0| < BRANCH_EPS */
		p0 = p; q0 = q;
		stamp(3);
	}
	double w = -hu_log_call(q);
	if(w > maxL) w = maxL;
	return w;
}

/* DBG: diagnostic build, s_memtime stamps per phase into dbg[block][8] (load, tables, sweeps, EM, total) */
/* VL = 3: the whole normalised v message lives in (dynamic) LDS, 3 x SPT x THREADS doubles: with u and the ratios
 * the per-thread state is 48 registers at SPT = 6, the kernel fits 128 VGPRs and FOUR workgroups of four waves share
 * a CU — a SIMD needs four waves to issue FP64 at its full rate (one wave alone issues every ~5 cycles, measured).
 * VL = 1: the third component of the normalised v message lives in LDS (SPT x THREADS doubles) instead of
 * registers: 2 x 12 sites x 7 doubles do not fit 256 VGPRs beside the temporaries, and the spills were reloaded
 * inside the serial table phases */
/* site list of every read for the split placement kernel: offsets (from the region's first column) of its gap sites in
 * [0, gapCap), of its base sites in [gapCap, gapCap + baseCap), and the two counts.  One wave per read.  The caps are chosen
 * by the host from the counts of k_site_count, so nothing is clipped. */
__global__ __launch_bounds__(64) void k_site_count(HuDbDev db, int n, const int8_t* __restrict__ codes, const int32_t* __restrict__ rstart,
		const int32_t* __restrict__ rend, int32_t* __restrict__ cnt) {
	const int read = blockIdx.x, lane = threadIdx.x;
	const int start = rstart[read], len = rend[read] - start + 1;
	int g = 0, bc = 0;
	for(int j0 = 0; j0 < len; j0 += 64) {
		const int j = j0 + lane;
		const bool val = j < len;
		const int b = val ? codes[(size_t) read * db.csLen + start + j] : 0;
		g += __popcll(__ballot(val && b < 0)); bc += __popcll(__ballot(val && b >= 0));
	}
	if(lane == 0) { cnt[2 * read] = g; cnt[2 * read + 1] = bc; }
}
__global__ __launch_bounds__(64) void k_site_perm(HuDbDev db, int n, const int8_t* __restrict__ codes, const int32_t* __restrict__ rstart,
		const int32_t* __restrict__ rend, int gapCap, int baseCap, uint16_t* __restrict__ perm) {
	const int read = blockIdx.x, lane = threadIdx.x;
	const int start = rstart[read], len = rend[read] - start + 1;
	if(hu_skip_width(db, len)) return;
	uint16_t* __restrict__ pr = perm + (size_t) read * (gapCap + baseCap);
	const unsigned long long lt = (1ull << lane) - 1ull;
	int g = 0, bc = 0;
	for(int j0 = 0; j0 < len; j0 += 64) {
		const int j = j0 + lane;
		const bool val = j < len;
		const int b = val ? codes[(size_t) read * db.csLen + start + j] : 0;
		const unsigned long long mg = __ballot(val && b < 0), mb = __ballot(val && b >= 0);
		if(val && b < 0) { const int at = g + __popcll(mg & lt); if(at < gapCap) pr[at] = (uint16_t) j; }
		if(val && b >= 0) { const int at = bc + __popcll(mb & lt); if(at < baseCap) pr[gapCap + at] = (uint16_t) j; }
		g += __popcll(mg); bc += __popcll(mb);
	}
}

/* GS > 0: the sites of a read are taken in the order of its site list (k_site_perm): gap sites in slots 0 .. GS-1, base sites
 * in slots GS .. SPT-1.  The sweeps over the gap slots are straight-line code on ONE table held in registers (every lane
 * would read the same 160 bytes per site and sweep from LDS: the sweeps are bound by the LDS return path, and 82 % of a
 * region's sites are gaps); only the base slots read their per-site table. */
template<int SPT, int NW, int EMV, int RED, int OCC, bool DBG = false, int VL = 0, int GS = 0>
__global__ __launch_bounds__(64 * NW, OCC) void k_place_blk(HuDbDev db, HuModelDev mdl, const int8_t* __restrict__ codes,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend,
		const HuCand* __restrict__ cands, HuPlaceOut* __restrict__ out, long long* __restrict__ dbg = nullptr, const uint32_t* __restrict__ order = nullptr,
		const uint16_t* __restrict__ perm = nullptr, const int32_t* __restrict__ permCnt = nullptr, int xmap = 0) {
	constexpr int THREADS = 64 * NW;
	long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, tl = 0;
	long long te[8] = {0, 0, 0, 0, 0, 0, 0, 0};        /* DBG: inside the EM steps */
	auto stamp = [&](int slot) { if(DBG) { const long long t = (long long) __builtin_amdgcn_s_memtime(); tk[slot] += t - tl; tl = t; } };
	if(DBG) { t0 = tl = (long long) __builtin_amdgcn_s_memtime(); }
	__shared__ double red[2 * NW], redc[2 * NW];
	__shared__ double cst[HU_PC_COUNT];
	__shared__ double Etab[3 * HU_MAX_DGK * 4];          /* [which][k][m] = exp(lam_m len_which rate_k) */
	__shared__ double Gtab[16];
	__shared__ __attribute__((aligned(16))) double tabM[5 * HU_TP];
	__shared__ __attribute__((aligned(16))) double tabD[6 * 4];             /* sweep (i): row 0 = G_mm; sweep (ii): row b = T^b_mm */
	__shared__ double vl[VL == 1 ? SPT * 64 * NW : 1];
	extern __shared__ double vdyn[];                   /* VL == 3: [3][SPT * THREADS] */
	const int tid = threadIdx.x;
	/* the serial table work of a workgroup runs in one or two of its waves: rotate which by workgroup so that
	 * the workgroups sharing a CU load different SIMDs with it */
	const int vt = (tid + 64 * (blockIdx.x % NW)) % THREADS;
	const uint32_t ci = order ? order[hu_xcd_pos(blockIdx.x, gridDim.x, xmap)] : blockIdx.x;     /* candidates in node order, an eighth of the list per XCD: see k_estimate_prod */
	const HuCand cd = cands[ci];
	const int read = cd.read, un = cd.node;
	const int start = rstart[read], end = rend[read], n = end - start + 1;
	if(hu_skip_width(db, n)) return;
	const int Kc = mdl.dgK > 0 ? mdl.dgK : 1;
	const double rKc = 1.0 / (double) Kc;
	const int8_t* __restrict__ cdr = codes + (size_t) read * db.csLen + start;
	const int64_t mOff = ((int64_t) un * db.winLen + (start - db.winStart)) * 4;
	const double* __restrict__ Ub = db.up + mOff;
	const double* __restrict__ Vb = db.down + mOff;
	const int nGap = GS > 0 ? permCnt[2 * read] : 0, nBase = GS > 0 ? permCnt[2 * read + 1] : 0;
	double u[SPT][3], v[SPT][3]; unsigned long long bop = 0, bop2 = 0;   /* base / gap code of slot t: 3 bits, slots 21.. in bop2 */
	{
		double aU[SPT][4], aV[SPT][4];
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			int jj;
			if(GS > 0) { /* slot t of the site list: [0, GS x THREADS) gap sites, then the base sites */
				const int i = t < GS ? tid + THREADS * t : tid + THREADS * (t - GS);
				const bool val = i < (t < GS ? nGap : nBase);
				jj = val ? (int) perm[(size_t) read * (SPT * THREADS) + (t < GS ? 0 : GS * THREADS) + i] : 0;
			}
			else { const int j = tid + THREADS * t; jj = j < n ? j : 0; }
			load4(Ub + (size_t) jj * 4, aU[t]); load4(Vb + (size_t) jj * 4, aV[t]);
			const int b = cdr[jj];
			if(t < 21) bop |= (unsigned long long)(b >= 0 ? b : 4) << (3 * t);
			else bop2 |= (unsigned long long)(b >= 0 ? b : 4) << (3 * (t - 21));
		}
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			const double iu = 1.0 / aU[t][0], iv = 1.0 / aV[t][0];
#pragma unroll
			for(int m = 0; m < 3; ++m) { u[t][m] = aU[t][m + 1] * iu; v[t][m] = aV[t][m + 1] * iv; }
			if(VL == 1) { vl[t * THREADS + tid] = v[t][2]; v[t][2] = 0; }
			if(VL == 3) {
#pragma unroll
				for(int m = 0; m < 3; ++m) { vdyn[(m * SPT + t) * THREADS + tid] = v[t][m]; v[t][m] = 0; }
			}
		}
	}
	for(int i = tid; i < HU_PC_COUNT; i += THREADS) cst[i] = db.placeConst[i];
	lds_barrier();
	const double* clam = cst + HU_PC_LAM; const double* crate = cst + HU_PC_RATE; const double* cW = cst + HU_PC_W;
	const double* cC = cst + HU_PC_C; const double* ccb = cst + HU_PC_CB;
	const double w0 = db.blen[un];
	double lenUR = w0 * cd.ratio0, lenVR = w0 * (1 - cd.ratio0), lenNR = cd.wnr0;
	double wur0 = lenUR, wnr0 = lenNR;
	const double w0j = lenUR + lenVR;
	double wur = wur0, wnr = wnr0;
	int iter = 0, emIters = 0, phase = 0;
	double rho[SPT];
	if(DBG) { double x = 0; for(int t = 0; t < SPT; ++t) x += u[t][0] + u[t][1]; if(x == 1.2345e-300) tk[7] = 1; } /* wait for the loads */
	stamp(0);
	for(; iter < HU_MAX_ITER && 0 <= wur && wur <= w0j; ++iter) {
		if(vt < 8 * Kc) {
			const int which = vt / (4 * Kc), k = (vt >> 2) % Kc, m = vt & 3;
			Etab[(which * HU_MAX_DGK + k) * 4 + m] = hu_exp_call(clam[m] * ((which ? lenVR : lenUR) * crate[k]));
		}
		lds_barrier();
		if(vt < 16) { /* G_mn = mean_k exp(lam_m w_ur r_k) exp(lam_n w_vr r_k) */
			const int m = vt >> 2, nn = vt & 3;
			double g = 0;
			for(int k = 0; k < Kc; ++k) g += Etab[k * 4 + m] * Etab[(HU_MAX_DGK + k) * 4 + nn];
			Gtab[vt] = g * rKc;
		}
		lds_barrier();
		for(int e = vt; e < 84; e += THREADS) { /* one entry per thread when the workgroup has two or more waves */
			if(e < 80) tabM[(e >> 4) * HU_TP + (e & 15)] = cW[e] * Gtab[e & 15];
			else tabD[e - 80] = Gtab[(e - 80) * 5];
		}
		lds_barrier();
		stamp(1);
		/* (i) message r->n from children u, v against the read's leaf message; EM on the n-r branch */
		int nv = 0;
		{
			unsigned long long bq = bop, bq2 = bop2;
			asm volatile("" : "+v"(bq), "+v"(bq2));  /* the per-site table addresses are recomputed per sweep, not kept live */
			const double g0 = tabD[0], g1 = tabD[1], g2 = tabD[2], g3 = tabD[3];
			double Mg[GS > 0 ? 16 : 1];
			if(GS > 0) {
#pragma unroll
				for(int e = 0; e < 16; ++e) Mg[e] = tabM[4 * HU_TP + e];
			}
#pragma unroll
			for(int t = 0; t < SPT; ++t) {
				const double* M = (GS > 0 && t < GS) ? Mg : tabM + (unsigned)(((t < 21 ? bq >> (3 * t) : bq2 >> (3 * (t - 21)))) & 7u) * HU_TP;
				const double v0 = VL == 3 ? vdyn[(0 * SPT + t) * THREADS + tid] : v[t][0], v1 = VL == 3 ? vdyn[(1 * SPT + t) * THREADS + tid] : v[t][1];
				const double v2 = VL == 3 ? vdyn[(2 * SPT + t) * THREADS + tid] : VL == 1 ? vl[t * THREADS + tid] : v[t][2];
				double num = fma(M[3], v2, fma(M[2], v1, fma(M[1], v0, M[0])));
#pragma unroll
				for(int m = 1; m < 4; ++m) {
					const double tm = fma(M[m * 4 + 3], v2, fma(M[m * 4 + 2], v1, fma(M[m * 4 + 1], v0, M[m * 4 + 0])));
					num = fma(tm, u[t][m - 1], num);
				}
				const double den = fma(g3 * u[t][2], v2, fma(g2 * u[t][1], v1, fma(g1 * u[t][0], v0, g0)));
				const double r = fast_div(num, den);
				const bool inr = GS > 0 ? (t < GS ? tid + THREADS * t < nGap : tid + THREADS * (t - GS) < nBase) : tid + THREADS * t < n;
				const bool ok = inr && fabs(r) < HU_RHO_SKIP; /* false for NaN, inf */
				rho[t] = ok ? r : HU_RHO_SKIP;
				nv += __popcll(__ballot(ok));
			}
		}
		stamp(2);
		wnr = em_branch_blk<SPT, NW, EMV, RED>(rho, nv, lenNR, 1.0, red, redc, phase, emIters, DBG ? te : nullptr);
		lenNR = wnr;
		stamp(3);
		if(vt < 4 * Kc) {
			const int k = vt >> 2, m = vt & 3;
			Etab[(2 * HU_MAX_DGK + k) * 4 + m] = hu_exp_call(clam[m] * (lenNR * crate[k]));
		}
		lds_barrier();
		if(vt < 16) { /* G'_mn = mean_k exp(lam_m w_vr r_k) exp(lam_n w_nr r_k) */
			const int m = vt >> 2, nn = vt & 3;
			double g = 0;
			for(int k = 0; k < Kc; ++k) g += Etab[(HU_MAX_DGK + k) * 4 + m] * Etab[(2 * HU_MAX_DGK + k) * 4 + nn];
			Gtab[vt] = g * rKc;
		}
		lds_barrier();
		for(int e = vt; e < 100; e += THREADS) {
			if(e < 80) { /* T^b_mn = c^b_n G'_mn; Z^b_mk = sum_n T^b_mn C_mnk */
				const int b = e >> 4, m = (e >> 2) & 3, kk = e & 3;
				double z = 0;
#pragma unroll
				for(int nn = 0; nn < 4; ++nn) z = fma(ccb[b * 4 + nn] * Gtab[m * 4 + nn], cC[(m * 4 + nn) * 4 + kk], z);
				tabM[b * HU_TP + m * 4 + kk] = z;
			}
			else { const int b = (e - 80) >> 2, m = (e - 80) & 3; tabD[b * 4 + m] = ccb[b * 4 + m] * Gtab[m * 5]; }
		}
		lds_barrier();
		stamp(1);
		/* (ii) message r->u from children v, n against u's own message; EM on the u-r branch */
		nv = 0;
		{
			unsigned long long bq = bop, bq2 = bop2;
			asm volatile("" : "+v"(bq), "+v"(bq2));
			double Mg[GS > 0 ? 16 : 1], Dg[GS > 0 ? 4 : 1];
			if(GS > 0) {
#pragma unroll
				for(int e = 0; e < 16; ++e) Mg[e] = tabM[4 * HU_TP + e];
#pragma unroll
				for(int e = 0; e < 4; ++e) Dg[e] = tabD[4 * 4 + e];
			}
#pragma unroll
			for(int t = 0; t < SPT; ++t) {
				const unsigned bi = (unsigned)((t < 21 ? bq >> (3 * t) : bq2 >> (3 * (t - 21))) & 7u);
				const double* M = (GS > 0 && t < GS) ? Mg : tabM + bi * HU_TP;
				const double* D = (GS > 0 && t < GS) ? Dg : tabD + bi * 4;
				const double v0 = VL == 3 ? vdyn[(0 * SPT + t) * THREADS + tid] : v[t][0], v1 = VL == 3 ? vdyn[(1 * SPT + t) * THREADS + tid] : v[t][1];
				const double v2 = VL == 3 ? vdyn[(2 * SPT + t) * THREADS + tid] : VL == 1 ? vl[t * THREADS + tid] : v[t][2];
				double A = fma(M[3], u[t][2], fma(M[2], u[t][1], fma(M[1], u[t][0], M[0])));
#pragma unroll
				for(int m = 1; m < 4; ++m) {
					const double tm = fma(M[m * 4 + 3], u[t][2], fma(M[m * 4 + 2], u[t][1], fma(M[m * 4 + 1], u[t][0], M[m * 4 + 0])));
					A = fma(tm, m == 3 ? v2 : m == 2 ? v1 : v0, A);
				}
				const double piX = fma(D[3], v2, fma(D[2], v1, fma(D[1], v0, D[0])));
				const double r = fast_div(A, piX);
				const bool inr = GS > 0 ? (t < GS ? tid + THREADS * t < nGap : tid + THREADS * (t - GS) < nBase) : tid + THREADS * t < n;
				const bool ok = inr && fabs(r) < HU_RHO_SKIP;
				rho[t] = ok ? r : HU_RHO_SKIP;
				nv += __popcll(__ballot(ok));
			}
		}
		stamp(2);
		wur = em_branch_blk<SPT, NW, EMV, RED>(rho, nv, lenUR, w0j, red, redc, phase, emIters, DBG ? te : nullptr);
		lenUR = wur;
		lenVR = w0j - wur;
		stamp(3);
		if(fabs(wur - wur0) < HU_BRANCH_EPS && fabs(wnr - wnr0) < HU_BRANCH_EPS) { ++iter; break; }
		wur0 = wur; wnr0 = wnr;
	}
	if(tid == 0) { HuPlaceOut o; o.wnr = lenNR; o.wur = lenUR; o.iters = iter; o.pad = emIters; out[ci] = o; }
	if(DBG && tid == 0) {
		tk[4] = (long long) __builtin_amdgcn_s_memtime() - t0; tk[5] = iter; tk[6] = emIters;
		for(int i = 0; i < 8; ++i) dbg[(size_t) blockIdx.x * 8 + i] = tk[i];
		for(int i = 0; i < 4; ++i) dbg[(size_t) gridDim.x * 8 + (size_t) blockIdx.x * 4 + i] = te[i];
	}
}

/* Measured and not kept (round 2), both aimed at THREE waves per SIMD for the placement (k_place_blk issues VALU 49 % of the time at two):
 * (a) the 12-slot kernel with the third components of both messages in LDS, the model constants through the scalar cache and the
 *     gap-slot tables in scalar registers: the compiler reaches 168 VGPRs only with 288 B / lane of scratch: 12.0 ms against 8.9;
 * (b) three waves x 8 slots per candidate (k_place_blk<8, 3, ...>, 168 VGPRs, 96 B of scratch, four workgroups per CU): 10.4 ms —
 *     the three-wave reductions of every EM step cost more than the third wave per SIMD hides. */
