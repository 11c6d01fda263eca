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

