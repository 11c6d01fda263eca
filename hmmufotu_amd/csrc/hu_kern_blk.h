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
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, const uint32_t* __restrict__ pairs,
		const int32_t* __restrict__ seedCnt, const int32_t* __restrict__ seedId, const uint32_t* __restrict__ seedDN,
		int weighted, HuEstOut* __restrict__ out) {
	__shared__ double red[2 * THREADS / 64];
	__shared__ long long redl[2 * THREADS / 64];
	constexpr int NW = THREADS / 64;
	const int read = blockIdx.x / HU_MAX_SEEDS, s = blockIdx.x % HU_MAX_SEEDS, tid = threadIdx.x;
	if(s >= seedCnt[read]) return;
	const int u = seedId[(size_t) read * HU_MAX_SEEDS + s];
	const int v = db.parent[u];
	const uint32_t dn = seedDN[(size_t) read * HU_MAX_SEEDS + s];
	const uint32_t pv = pairs[(size_t) read * db.nNodesPad + v];
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
 * k_place_blk: joint branch-length optimisation of one candidate placement per workgroup
 * (src/PhyloTreeUnrooted.cpp:800-847 outer loop, :749-798 EM), table-driven.
 *
 * With both messages of a site in the eigenbasis (a_U, a_V) and V orthonormal (U = Pi^-1/2 V, so that
 * sum_i pi_i U_im U_in = delta_mn), every per-site quantity of the two sweeps is a bilinear form:
 *   sweep (i)   rho = (a_U^T N^b a_V) / (sum_m G_mm a_Um a_Vm),   N^b_mn = W^b_mn G_mn
 *   sweep (ii)  rho = (a_V^T Z^b a_U) / (sum_m T^b_mm a_Vm),       Z^b_mk = sum_n T^b_mn C_mnk, T^b_mn = c^b_n G'_mn
 * (a_U pre-scaled so that pi . e_U = 1), b = the read's base at the site or 4 for a gap, G / G' the
 * category-averaged products of exponentials of the two branch lengths involved.  The five 4x4 tables are
 * rebuilt per sweep by 80 threads; a site costs ~30 FMA + one division instead of three 4x4 matvecs per
 * rate category.  The messages (8 doubles per site) and the per-site ratios live in registers for the whole
 * optimisation: every message byte crosses HBM once (the streaming k_place re-reads them per sweep).
 * The EM step is 1 / (1 + rho_j q0/p0) summed over the sites: fma, v_rcp_f64, two Newton steps, add. */
#define HU_RHO_SKIP 1e200        /* sentinel ratio of a site the EM skips (NaN ratio in the reference, padding) */
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

template<int SPT>
__device__ inline double em_branch_blk(const double (&rho)[SPT], double cnt, double w0, double maxL, double* red, int& phase, int& emIters) {
	double q0 = exp(-w0), p0 = 1 - q0, p = p0, q = q0;
	const double rc = 1.0 / cnt;
	for(int it = 0; it < HU_MAX_ITER && p >= 0 && p <= 1; ++it) {
		const double k = fast_div(q0, p0);
		double s = 0;
		if(k >= 1e-100 && k <= 1e100) { /* p0 / (rho q0 + p0) = 1 / (rho k + 1); skipped sites contribute < 1e-100 */
#pragma unroll
			for(int t = 0; t < SPT; ++t) {
				const double x = fma(rho[t], k, 1.0);
				double y = __builtin_amdgcn_rcp(x);
				y = fma(fma(-x, y, 1.0), y, y);
				y = fma(fma(-x, y, 1.0), y, y);
				s += y;
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
		s = block_sum(s, red, phase);
		p = s * rc; q = 1 - p;
		++emIters;
		if(q0 * HU_EXP_MEPS < q && q < q0 * HU_EXP_PEPS) break; /* |log q - log q0| < BRANCH_EPS */
		p0 = p; q0 = q;
	}
	double w = -log(q);
	if(w > maxL) w = maxL;
	return w;
}

template<int SPT>
__global__ __launch_bounds__(HU_BLK_THREADS) void k_place_blk(HuDbDev db, HuModelDev mdl, const int8_t* __restrict__ codes,
		const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend,
		const HuCand* __restrict__ cands, HuPlaceOut* __restrict__ out) {
	__shared__ double red[8];
	__shared__ double cst[HU_PC_COUNT];
	__shared__ double Etab[3 * HU_MAX_DGK * 4];          /* [which][k][m] = exp(lam_m len_which rate_k) */
	__shared__ __attribute__((aligned(16))) double tabM[5 * HU_TP];
	__shared__ __attribute__((aligned(16))) double tabD[6 * 4];             /* sweep (i): row 0 = G_mm; sweep (ii): row b = T^b_mm */
	const int tid = threadIdx.x;
	const HuCand cd = cands[blockIdx.x];
	const int read = cd.read, u = cd.node;
	const int start = rstart[read], end = rend[read], n = end - start + 1;
	const int Kc = mdl.dgK > 0 ? mdl.dgK : 1;
	const double rKc = 1.0 / (double) Kc;
	const int8_t* __restrict__ cdr = codes + (size_t) read * db.csLen + start;
	const int64_t mOff = ((int64_t) u * db.winLen + (start - db.winStart)) * 4;
	const double* __restrict__ Ub = db.up + mOff;
	const double* __restrict__ Vb = db.down + mOff;
	/* the candidate's two messages (already in the eigenbasis), once from HBM */
	double aU[SPT][4], aV[SPT][4]; int bo[SPT];
#pragma unroll
	for(int t = 0; t < SPT; ++t) {
		const int j = tid + HU_BLK_THREADS * t, jj = j < n ? j : 0;
		load4(Ub + (size_t) jj * 4, aU[t]); load4(Vb + (size_t) jj * 4, aV[t]);
		const int b = cdr[jj];
		bo[t] = (b >= 0 ? b : 4);
	}
	for(int i = tid; i < HU_PC_COUNT; i += HU_BLK_THREADS) cst[i] = db.placeConst[i];
	lds_barrier();
	const double* clam = cst + HU_PC_LAM; const double* crate = cst + HU_PC_RATE; const double* cW = cst + HU_PC_W;
	const double* cC = cst + HU_PC_C; const double* ccb = cst + HU_PC_CB;
	{
		const double s0 = cst[HU_PC_S + 0], s1 = cst[HU_PC_S + 1], s2 = cst[HU_PC_S + 2], s3 = cst[HU_PC_S + 3];
#pragma unroll
		for(int t = 0; t < SPT; ++t) { /* pi . e_U = 1 from here on: both ratios are invariant under a scaling of a_U */
			const double piU = (s0 * aU[t][0] + s2 * aU[t][2]) + (s1 * aU[t][1] + s3 * aU[t][3]);
			const double inv = 1.0 / piU;
#pragma unroll
			for(int m = 0; m < 4; ++m) aU[t][m] *= inv;
		}
	}
	const double w0 = db.blen[u];
	double lenUR = w0 * cd.ratio0, lenVR = w0 * (1 - cd.ratio0), lenNR = cd.wnr0;
	double wur0 = lenUR, wnr0 = lenNR;
	const double w0j = lenUR + lenVR;
	double wur = wur0, wnr = wnr0;
	int iter = 0, emIters = 0, phase = 0;
	double rho[SPT];
	for(; iter < HU_MAX_ITER && 0 <= wur && wur <= w0j; ++iter) {
		if(tid < 8 * Kc) {
			const int which = tid / (4 * Kc), k = (tid >> 2) % Kc, m = tid & 3;
			Etab[(which * HU_MAX_DGK + k) * 4 + m] = exp(clam[m] * ((which ? lenVR : lenUR) * crate[k]));
		}
		lds_barrier();
		if(tid < 84) { /* G_mn = mean_k exp(lam_m w_ur r_k) exp(lam_n w_vr r_k) */
			const int b = tid >> 4, m = tid < 80 ? (tid >> 2) & 3 : tid - 80, nn = tid < 80 ? tid & 3 : tid - 80;
			double g = 0;
			for(int k = 0; k < Kc; ++k) g += Etab[k * 4 + m] * Etab[(HU_MAX_DGK + k) * 4 + nn];
			g *= rKc;
			if(tid < 80) tabM[b * HU_TP + m * 4 + nn] = cW[b * 16 + m * 4 + nn] * g;
			else tabD[m] = g;
		}
		lds_barrier();
		/* (i) message r->n from children u, v against the read's leaf message; EM on the n-r branch */
		double cnt = 0;
		{
			const double g0 = tabD[0], g1 = tabD[1], g2 = tabD[2], g3 = tabD[3];
#pragma unroll
			for(int t = 0; t < SPT; ++t) {
				const double* M = tabM + bo[t] * HU_TP;
				double num = 0;
#pragma unroll
				for(int nn = 0; nn < 4; ++nn) {
					const double tn = (aU[t][0] * M[0 * 4 + nn] + aU[t][2] * M[2 * 4 + nn]) + (aU[t][1] * M[1 * 4 + nn] + aU[t][3] * M[3 * 4 + nn]);
					num = fma(tn, aV[t][nn], num);
				}
				const double den = (g0 * aU[t][0] * aV[t][0] + g2 * aU[t][2] * aV[t][2]) + (g1 * aU[t][1] * aV[t][1] + g3 * aU[t][3] * aV[t][3]);
				double r = fast_div(num, den);
				const bool ok = tid + HU_BLK_THREADS * t < n && fabs(r) < HU_RHO_SKIP; /* false for NaN, inf */
				rho[t] = ok ? r : HU_RHO_SKIP;
				cnt += ok ? 1.0 : 0.0;
			}
		}
		cnt = block_sum(cnt, red, phase);
		wnr = em_branch_blk<SPT>(rho, cnt, lenNR, 1.0, red, phase, emIters);
		lenNR = wnr;
		if(tid < 4 * Kc) {
			const int k = tid >> 2, m = tid & 3;
			Etab[(2 * HU_MAX_DGK + k) * 4 + m] = exp(clam[m] * (lenNR * crate[k]));
		}
		lds_barrier();
		if(tid < 100) { /* G'_mn = mean_k exp(lam_m w_vr r_k) exp(lam_n w_nr r_k); T^b_mn = c^b_n G'_mn */
			const int b = tid < 80 ? tid >> 4 : (tid - 80) >> 2, m = tid < 80 ? (tid >> 2) & 3 : (tid - 80) & 3, kk = tid & 3;
			double z = 0, tmm = 0;
#pragma unroll
			for(int nn = 0; nn < 4; ++nn) {
				double g = 0;
				for(int k = 0; k < Kc; ++k) g += Etab[(HU_MAX_DGK + k) * 4 + m] * Etab[(2 * HU_MAX_DGK + k) * 4 + nn];
				const double T = ccb[b * 4 + nn] * (g * rKc);
				z = fma(T, cC[(m * 4 + nn) * 4 + kk], z);
				if(nn == m) tmm = T;
			}
			if(tid < 80) tabM[b * HU_TP + m * 4 + kk] = z;
			else tabD[b * 4 + m] = tmm;
		}
		lds_barrier();
		/* (ii) message r->u from children v, n against u's own message; EM on the u-r branch */
		cnt = 0;
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			const double* M = tabM + bo[t] * HU_TP;
			const double* D = tabD + bo[t] * 4;
			double A = 0;
#pragma unroll
			for(int m = 0; m < 4; ++m) {
				const double tm = (M[m * 4 + 0] * aU[t][0] + M[m * 4 + 2] * aU[t][2]) + (M[m * 4 + 1] * aU[t][1] + M[m * 4 + 3] * aU[t][3]);
				A = fma(tm, aV[t][m], A);
			}
			const double piX = (D[0] * aV[t][0] + D[2] * aV[t][2]) + (D[1] * aV[t][1] + D[3] * aV[t][3]);
			double r = fast_div(A, piX);
			const bool ok = tid + HU_BLK_THREADS * t < n && fabs(r) < HU_RHO_SKIP;
			rho[t] = ok ? r : HU_RHO_SKIP;
			cnt += ok ? 1.0 : 0.0;
		}
		cnt = block_sum(cnt, red, phase);
		wur = em_branch_blk<SPT>(rho, cnt, lenUR, w0j, red, phase, emIters);
		lenUR = wur;
		lenVR = w0j - wur;
		if(fabs(wur - wur0) < HU_BRANCH_EPS && fabs(wnr - wnr0) < HU_BRANCH_EPS) { ++iter; break; }
		wur0 = wur; wnr0 = wnr;
	}
	if(tid == 0) { HuPlaceOut o; o.wnr = lenNR; o.wur = lenUR; o.iters = iter; o.pad = emIters; out[blockIdx.x] = o; }
}
