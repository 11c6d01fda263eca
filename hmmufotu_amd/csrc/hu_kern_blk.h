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
/* e = U a */
__device__ inline void from_eig(const HuModelDev& m, const double* a, double* e) {
#pragma unroll
	for(int i = 0; i < 4; ++i) e[i] = fmax((m.U[i*4+0] * a[0] + m.U[i*4+1] * a[1]) + (m.U[i*4+2] * a[2] + m.U[i*4+3] * a[3]), 0.0);
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
			double a[4], c[4];
			if(wur == 0) { for(int i = 0; i < 4; ++i) z[t][i] = eU[t][i]; } else { to_eig(mdl, eU[t], a); conv_eig(mdl, Eu, a, z[t]); }
			if(wvr == 0) { for(int i = 0; i < 4; ++i) c[i] = eV[t][i]; } else { to_eig(mdl, eV[t], a); conv_eig(mdl, Ev, a, c); }
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

/* EM on register-resident ratios; the count of usable sites is taken once */
template<int SPT>
__device__ inline double em_branch_blk(const double (&rho)[SPT], int n, double cnt, double w0, double maxL, double* red, int& phase, int& emIters) {
	double q0 = exp(-w0), p0 = 1 - q0, p = p0, q = q0;
	for(int it = 0; it < HU_MAX_ITER && p >= 0 && p <= 1; ++it) {
		double s = 0;
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			const double r = rho[t];               /* NaN marks both "unusable site" and "beyond n" */
			const double x = fma(r, q0, p0);
			double tt;
			if(x > 1e-300 && x < 1e300) {
				double y = __builtin_amdgcn_rcp(x);
				y = fma(y, fma(-x, y, 1.0), y);
				y = fma(y, fma(-x, y, 1.0), y);
				tt = p0 * y;
			}
			else tt = p0 / x;
			s += isnan(r) ? 0.0 : tt;
		}
		s = block_sum(s, red, phase);
		p = s / cnt; q = 1 - p;
		++emIters;
		if(fabs(log(q) - log(q0)) < HU_BRANCH_EPS) break;
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
	__shared__ double Etab[3 * HU_MAX_DGK * 4];
	__shared__ double Ltab[HU_MAX_DGK * 5 * 4];
	const int tid = threadIdx.x;
	const HuCand cd = cands[blockIdx.x];
	const int read = cd.read, u = cd.node;
	const int start = rstart[read], end = rend[read], n = end - start + 1;
	const int Kc = mdl.dgK > 0 ? mdl.dgK : 1;
	const int8_t* __restrict__ cdr = codes + (size_t) read * db.csLen + start;
	const int64_t mOff = ((int64_t) u * db.winLen + (start - db.winStart)) * 4;
	const double* __restrict__ Ub = db.up + mOff;
	const double* __restrict__ Vb = db.down + mOff;
	/* the candidate's two messages, once from HBM, kept in the eigenbasis */
	double aU[SPT][4], aV[SPT][4]; int bb[SPT];
	{
		double eU[SPT][4], eV[SPT][4];
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			const int j = tid + HU_BLK_THREADS * t, jj = j < n ? j : 0;
			load4(Ub + (size_t) jj * 4, eU[t]); load4(Vb + (size_t) jj * 4, eV[t]); bb[t] = cdr[jj];
		}
#pragma unroll
		for(int t = 0; t < SPT; ++t) { to_eig(mdl, eU[t], aU[t]); to_eig(mdl, eV[t], aV[t]); }
	}
	const double w0 = db.blen[u];
	double lenUR = w0 * cd.ratio0, lenVR = w0 * (1 - cd.ratio0), lenNR = cd.wnr0;
	double wur0 = lenUR, wnr0 = lenNR;
	const double w0j = lenUR + lenVR;
	double wur = wur0, wnr = wnr0;
	double pi2 = 0;
	for(int i = 0; i < 4; ++i) pi2 += mdl.pi[i] * mdl.pi[i];
	double api[4];
	to_eig(mdl, mdl.pi, api);
	int iter = 0, emIters = 0, phase = 0;
	double rho[SPT];
	for(; iter < HU_MAX_ITER && 0 <= wur && wur <= w0j; ++iter) {
		for(int i = tid; i < Kc * 4; i += HU_BLK_THREADS) {
			const double r = mdl.rate[i >> 2], l = mdl.lam[i & 3];
			Etab[0 * HU_MAX_DGK * 4 + i] = exp(l * (lenUR * r));
			Etab[1 * HU_MAX_DGK * 4 + i] = exp(l * (lenVR * r));
		}
		lds_barrier();
		/* (i) message r->n from children u, v; EM on the n-r branch against the read's leaf message */
		double cnt = 0;
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			double X[4] = {0, 0, 0, 0};
			for(int k = 0; k < Kc; ++k) {
				double cu[4], cv[4];
				if(lenUR == 0) from_eig(mdl, aU[t], cu); else conv_eig(mdl, Etab + k * 4, aU[t], cu);
				if(lenVR == 0) from_eig(mdl, aV[t], cv); else conv_eig(mdl, Etab + HU_MAX_DGK * 4 + k * 4, aV[t], cv);
				for(int i = 0; i < 4; ++i) X[i] += cu[i] * cv[i];
			}
			const double piX = (mdl.pi[0] * X[0] + mdl.pi[2] * X[2]) + (mdl.pi[1] * X[1] + mdl.pi[3] * X[3]);
			const int b = bb[t];
			double r;
			if(b >= 0) r = sel4(X, b) / piX;
			else r = ((mdl.pi[0] * mdl.pi[0] * X[0] + mdl.pi[2] * mdl.pi[2] * X[2]) + (mdl.pi[1] * mdl.pi[1] * X[1] + mdl.pi[3] * mdl.pi[3] * X[3])) / (piX * pi2);
			if(tid + HU_BLK_THREADS * t >= n) r = NAN;
			rho[t] = r;
			cnt += isnan(r) ? 0.0 : 1.0;
		}
		cnt = block_sum(cnt, red, phase);
		wnr = em_branch_blk<SPT>(rho, n, cnt, lenNR, 1.0, red, phase, emIters);
		lenNR = wnr;
		for(int i = tid; i < Kc * 4; i += HU_BLK_THREADS)
			Etab[2 * HU_MAX_DGK * 4 + i] = exp(mdl.lam[i & 3] * (lenNR * mdl.rate[i >> 2]));
		lds_barrier();
		for(int i = tid; i < Kc * 5; i += HU_BLK_THREADS) {
			const int k = i / 5, b = i % 5;
			double c[4];
			if(b < 4) {
				if(lenNR == 0) { for(int x = 0; x < 4; ++x) c[x] = x == b ? 1.0 : 0.0; }
				else { double a[4]; for(int m = 0; m < 4; ++m) a[m] = mdl.U1[m*4+b]; conv_eig(mdl, Etab + 2 * HU_MAX_DGK * 4 + k * 4, a, c); }
			}
			else {
				if(lenNR == 0) { for(int x = 0; x < 4; ++x) c[x] = mdl.pi[x]; }
				else conv_eig(mdl, Etab + 2 * HU_MAX_DGK * 4 + k * 4, api, c);
			}
			for(int x = 0; x < 4; ++x) Ltab[(k * 5 + b) * 4 + x] = c[x];
		}
		lds_barrier();
		/* (ii) message r->u from children v, n; EM on the u-r branch against u's own message */
		cnt = 0;
#pragma unroll
		for(int t = 0; t < SPT; ++t) {
			const int b = bb[t], bi = b >= 0 ? b : 4;
			double X[4] = {0, 0, 0, 0};
			for(int k = 0; k < Kc; ++k) {
				double cv[4];
				if(lenVR == 0) from_eig(mdl, aV[t], cv); else conv_eig(mdl, Etab + HU_MAX_DGK * 4 + k * 4, aV[t], cv);
				const double* cn = Ltab + (k * 5 + bi) * 4;
				for(int i = 0; i < 4; ++i) X[i] += cv[i] * cn[i];
			}
			double eU[4];
			from_eig(mdl, aU[t], eU);
			const double piX = (mdl.pi[0] * X[0] + mdl.pi[2] * X[2]) + (mdl.pi[1] * X[1] + mdl.pi[3] * X[3]);
			const double piU = (mdl.pi[0] * eU[0] + mdl.pi[2] * eU[2]) + (mdl.pi[1] * eU[1] + mdl.pi[3] * eU[3]);
			const double A = (mdl.pi[0] * X[0] * eU[0] + mdl.pi[2] * X[2] * eU[2]) + (mdl.pi[1] * X[1] * eU[1] + mdl.pi[3] * X[3] * eU[3]);
			double r = A / (piX * piU);
			if(tid + HU_BLK_THREADS * t >= n) r = NAN;
			rho[t] = r;
			cnt += isnan(r) ? 0.0 : 1.0;
		}
		cnt = block_sum(cnt, red, phase);
		wur = em_branch_blk<SPT>(rho, n, cnt, lenUR, w0j, red, phase, emIters);
		lenUR = wur;
		lenVR = w0j - wur;
		if(fabs(wur - wur0) < HU_BRANCH_EPS && fabs(wnr - wnr0) < HU_BRANCH_EPS) { ++iter; break; }
		wur0 = wur; wnr0 = wnr;
	}
	if(tid == 0) { HuPlaceOut o; o.wnr = lenNR; o.wur = lenUR; o.iters = iter; o.pad = emIters; out[blockIdx.x] = o; }
}
