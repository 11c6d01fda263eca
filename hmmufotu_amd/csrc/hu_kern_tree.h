// Tree pre-evaluation on the device: every directed-edge message of an unrooted
// tree by a post-order + a pre-order sweep over levels, i.e. what hmmufotu-build's "re-root at
// every node and evaluate" loop leaves in the .ptu (src/hmmufotu-build.cpp:415-466,
// src/PhyloTreeUnrooted.cpp:315-374), plus the ancestral argmax sequences
// (src/PhyloTreeUnrooted.cpp:1085-1093).  Messages are produced in the reference's LOG space
// with its scaling rule (src/PhyloTreeUnrooted.h:1495-1529); hu_db_create packs them afterwards.
#pragma once
#include "hu_common.h"
#include "hu_kern_sep.h"

/* log(P(t) . exp(M + scale)) - scale, scale = -510 - max M when max M < -510 */
__device__ inline void tree_conv(const HuModelDev& mdl, double t, const double* M, double* Y) {
	const double mx = max4d(M);
	const double scale = (mx != -INFINITY && mx < HU_MIN_LOGLIK_EXP) ? HU_MIN_LOGLIK_EXP - mx : 0.0;
	double e[4], c[4];
	for(int i = 0; i < 4; ++i) e[i] = exp(M[i] + scale);
	if(t == 0) { for(int i = 0; i < 4; ++i) c[i] = e[i]; }
	else {
		double E[4], a[4];
		for(int k = 0; k < 4; ++k) E[k] = exp(mdl.lam[k] * t);
		to_eig(mdl, e, a); conv_eig(mdl, E, a, c);
	}
	for(int i = 0; i < 4; ++i) Y[i] = log(c[i]) - scale;
}

/* combine contributions: X[k][i] summed over incoming messages, then row_mean_exp_scaled over the
 * rate categories for inner nodes (src/PhyloTreeUnrooted.cpp:329-343) */
struct TreeAcc {
	double X[HU_MAX_DGK][4];
	__device__ void init(int Kc) { for(int k = 0; k < Kc; ++k) for(int i = 0; i < 4; ++i) X[k][i] = 0; }
	__device__ void add(const HuModelDev& mdl, int Kc, double len, const double* M) {
		for(int k = 0; k < Kc; ++k) { double Y[4]; tree_conv(mdl, len * mdl.rate[k], M, Y); for(int i = 0; i < 4; ++i) X[k][i] += Y[i]; }
	}
	__device__ void finish(const HuModelDev& mdl, int Kc, double* out) const {
		if(mdl.dgK == 0) { for(int i = 0; i < 4; ++i) out[i] = X[0][i]; return; }
		for(int i = 0; i < 4; ++i) {
			double mx = X[0][i];
			for(int k = 1; k < Kc; ++k) mx = fmax(mx, X[k][i]);
			const double sc = (mx != -INFINITY && mx < HU_MIN_LOGLIK_EXP) ? HU_MIN_LOGLIK_EXP - mx : 0.0;
			double s = 0;
			for(int k = 0; k < Kc; ++k) s += exp(X[k][i] + sc);
			out[i] = log(s / Kc) - sc;
		}
	}
};

struct HuTreeDev {
	int32_t n, csLen, root;
	int64_t winStart, winLen;
	const int32_t* parent; const double* blen;
	const int32_t* childOff; const int32_t* childIdx;   /* CSR children lists */
	int8_t* seq;                                         /* [n][csLen]; inner rows are written */
	double* up; double* down;                            /* [n][winLen][4] */
};

/* up messages of the nodes of one level (all their children are done); grid (sites/256, nodes) */
__global__ __launch_bounds__(256) void k_tree_up(HuTreeDev t, HuModelDev mdl, const int32_t* __restrict__ nodes) {
	const int u = nodes[blockIdx.y];
	const int64_t w = (int64_t) blockIdx.x * 256 + threadIdx.x;
	if(w >= t.winLen) return;
	const int64_t j = t.winStart + w;
	const int Kc = mdl.dgK > 0 ? mdl.dgK : 1;
	double out[4];
	const int c0 = t.childOff[u], c1 = t.childOff[u + 1];
	if(c0 == c1) { /* leaf (src/PhyloTreeUnrooted.h:1431-1437) */
		const int b = t.seq[(size_t) u * t.csLen + j];
		for(int i = 0; i < 4; ++i) out[i] = b >= 0 ? (i == b ? 0.0 : -INFINITY) : mdl.logpi[i];
	}
	else {
		TreeAcc acc; acc.init(Kc);
		for(int c = c0; c < c1; ++c) {
			const int v = t.childIdx[c];
			double M[4]; load4(t.up + ((size_t) v * t.winLen + w) * 4, M);
			acc.add(mdl, Kc, t.blen[v], M);
		}
		acc.finish(mdl, Kc, out);
		t.seq[(size_t) u * t.csLen + j] = (int8_t) argmax4_tied(out); /* exact-arithmetic ties -> first index, like maxCoeff */
	}
	double* dst = t.up + ((size_t) u * t.winLen + w) * 4;
	*reinterpret_cast<double2*>(dst) = make_double2(out[0], out[1]);
	*reinterpret_cast<double2*>(dst + 2) = make_double2(out[2], out[3]);
}

/* down message parent(u) -> u of the nodes of one level: parent's own incoming message from above
 * (unless the parent is the root) and the up messages of u's siblings, in neighbour order */
__global__ __launch_bounds__(256) void k_tree_down(HuTreeDev t, HuModelDev mdl, const int32_t* __restrict__ nodes) {
	const int u = nodes[blockIdx.y];
	const int64_t w = (int64_t) blockIdx.x * 256 + threadIdx.x;
	if(w >= t.winLen) return;
	const int Kc = mdl.dgK > 0 ? mdl.dgK : 1;
	const int p = t.parent[u];
	TreeAcc acc; acc.init(Kc);
	if(p != t.root) { double M[4]; load4(t.down + ((size_t) p * t.winLen + w) * 4, M); acc.add(mdl, Kc, t.blen[p], M); }
	for(int c = t.childOff[p]; c < t.childOff[p + 1]; ++c) {
		const int v = t.childIdx[c];
		if(v == u) continue;
		double M[4]; load4(t.up + ((size_t) v * t.winLen + w) * 4, M);
		acc.add(mdl, Kc, t.blen[v], M);
	}
	double out[4];
	acc.finish(mdl, Kc, out);
	double* dst = t.down + ((size_t) u * t.winLen + w) * 4;
	*reinterpret_cast<double2*>(dst) = make_double2(out[0], out[1]);
	*reinterpret_cast<double2*>(dst + 2) = make_double2(out[2], out[3]);
}
