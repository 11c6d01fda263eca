// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_models.h header).  PARITY UNPINNED.
//
// Seed-Estimate-Place phylogenetic placement on a pre-evaluated unrooted tree.  Follows
//   src/SeqUtils.cpp:37-54 (pDist), src/HmmUFOtu_main.cpp:127-216 (getSeed, estimateSeq,
//   filterPlacements, placeSeq, calcQValues), src/hmmufotu.cpp:645-647,720-733,
//   src/PhyloTreeUnrooted.cpp:315-374 (loglikConv/loglik/evaluate), :707-954 (treeLoglik,
//   copySubTree, both optimizeBranchLength, estimateSeq, both placeSeq), :1018-1052,
//   :1166-1177; src/PhyloTreeUnrooted.h:410-510, :815-817, :1431-1445, :1488-1529,
//   :1584-1637; src/math/Stats.h:233-241.
//
// Floating-point association choices where Eigen's evaluation order is version dependent
// (Eigen3 is un-vendored and un-pinned):
//   * 4-term dot of a matrix ROW with a vector (row blocks are strided => scalar path,
//     redux_novec_unroller halves the range):        (e0+e1)+(e2+e3)
//   * 4-term dot of two contiguous Vector4d (SSE2 Packet2d): (e0+e2)+(e1+e3)
//   * rowwise mean over K rate categories: sequential sum / K
//   * exp/log: libm
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <limits>
#include "oracle_models.h"

namespace orc {

static const double NEG_INF = -std::numeric_limits<double>::infinity();
static const double MIN_LOGLIK_EXP = -1021 / 2; /* DBL_MIN_EXP / 2 in INTEGER arithmetic = -510 */
static const double INVALID_LOGLIK = 1;
static const double BRANCH_EPS = 1e-5;
static const int MAX_ITER = 100;
static const int MAX_Q = 250;

struct V4 { double v[4]; };

inline double max4(const double* v) { return std::max(std::max(v[0], v[1]), std::max(v[2], v[3])); }
inline int argmax4(const double* v) { /* Eigen maxCoeff(&idx): first strict maximum */
	int b = 0;
	for(int i = 1; i < 4; ++i) if(v[i] > v[b]) b = i;
	return b;
}
/* NOT the reference: the product's documented rule for components that tie in exact arithmetic — the first index
 * whose value is within a relative `tol` (in linear space) of the maximum.  The reference takes the first STRICT maximum of values whose
 * last bits depend on Eigen's summation order; where two components are mathematically equal (equal base frequencies) that order, not the
 * data, picks the winner.  tests/ use this variant (AssignOpts::tieTol > 0) to tell such sites from real disagreements. */
inline int argmax4_tol(const double* v, double tol) {
	if(!(tol > 0)) return argmax4(v);
	const double thr = max4(v) + std::log1p(-tol);
	for(int i = 0; i < 3; ++i) if(v[i] >= thr) return i;
	return 3;
}
inline double scale_of(double maxV) { return (maxV != NEG_INF && maxV < MIN_LOGLIK_EXP) ? MIN_LOGLIK_EXP - maxV : 0; }

/* dot_product_scaled(Matrix4d, Vector4d) (src/PhyloTreeUnrooted.h:1495-1503) */
inline V4 dps_mat(const double* X, const double* V) {
	V4 Y;
	double scale = scale_of(max4(V));
	double e[4];
	for(int k = 0; k < 4; ++k) e[k] = std::exp(V[k] + scale);
	for(int i = 0; i < 4; ++i) {
		const double* r = X + i * 4;
		Y.v[i] = std::log((r[0] * e[0] + r[1] * e[1]) + (r[2] * e[2] + r[3] * e[3])) - scale;
	}
	return Y;
}
/* dot_product_scaled(Vector4d, Vector4d) (src/PhyloTreeUnrooted.h:1505-1510) */
inline double dps_vec(const double* P, const double* V) {
	double scale = scale_of(max4(V));
	double e[4];
	for(int k = 0; k < 4; ++k) e[k] = std::exp(V[k] + scale);
	return std::log((P[0] * e[0] + P[2] * e[2]) + (P[1] * e[1] + P[3] * e[3])) - scale;
}
/* row_mean_exp_scaled (src/PhyloTreeUnrooted.h:1521-1529); X[k] = column k */
inline V4 row_mean_exp_scaled(const V4* X, int K) {
	V4 r;
	for(int i = 0; i < 4; ++i) {
		double mx = X[0].v[i];
		for(int k = 1; k < K; ++k) mx = std::max(mx, X[k].v[i]);
		double sc = scale_of(mx);
		double s = 0;
		for(int k = 0; k < K; ++k) s += std::exp(X[k].v[i] + sc);
		r.v[i] = std::log(s / K) - sc;
	}
	return r;
}

struct Tree {
	int nNodes = 0, csLen = 0, root = 0;
	const int* parent = nullptr;      // -1 for root
	const double* blen = nullptr;     // length of branch to parent
	const int8_t* seq = nullptr;      // nNodes x csLen
	const double* up = nullptr;       // message node->parent, [node][site][4]
	const double* down = nullptr;     // message parent->node
	const double* height = nullptr;
	const int* annoId = nullptr;      // optional taxon-annotation class per node
	Model model;
	int dgK = 0;                      // 0 = no discrete Gamma
	double dgR[16];
	long winStart = 0;                // messages are stored for columns [winStart, winStart+winLen)
	long winLen = 0;

	const int* rowOf = nullptr;       /* optional: up / down hold the rows of some nodes only (row of a node, -1 = absent: reading it is a test-harness bug) */

	size_t row(int node) const { if(!rowOf) return (size_t) node; if(rowOf[node] < 0) { std::fprintf(stderr, "oracle: messages of node %d are not resident\n", node); std::abort(); } return (size_t) rowOf[node]; }
	const double* U(int node, int j) const { return up + (row(node) * winLen + (j - winStart)) * 4; }
	const double* D(int node, int j) const { return down + (row(node) * winLen + (j - winStart)) * 4; }
	const int8_t* S(int node) const { return seq + (size_t) node * csLen; }
};

/* SeqUtils::pDist (src/SeqUtils.cpp:37-54) */
inline void pdist_counts(const int8_t* a, const int8_t* b, int start, int end, long& d, long& N) {
	d = 0; N = 0;
	for(int i = start; i <= end; ++i) {
		int b1 = a[i], b2 = b[i];
		if(b1 >= 0 && b2 >= 0) { N++; if(b1 != b2) d++; }
	}
}
inline double pdist(const int8_t* a, const int8_t* b, int start, int end) {
	long d, N; pdist_counts(a, b, start, end, d, N);
	return static_cast<double>(d) / N;
}

struct PTLoc { int start, end; long id; double dist; long d, N; };
inline bool operator<(const PTLoc& l, const PTLoc& r) { return l.dist < r.dist; }

enum TieMode { TIE_STABLE = 0, TIE_LIBSTDCXX = 1 };

/* getSeed + truncation (src/HmmUFOtu_main.cpp:127-152, src/hmmufotu.cpp:646-647).
 * TIE_LIBSTDCXX: literal std::sort on dist only (the reference's semantics; falls back to
 * stable when a NaN dist would make std::sort undefined).  TIE_STABLE: (dist, id)
 * lexicographic, NaN last — the deterministic order the GPU top-k implements. */
/* the loop of getSeed: every non-root node within maxHeight with its distance, in node order */
inline std::vector<PTLoc> scanSeeds(const Tree& t, const int8_t* seq, int start, int end, double maxHeight, bool& hasNaN) {
	std::vector<PTLoc> locs;
	hasNaN = false;
	for(int i = 0; i < t.nNodes; ++i) {
		if(i != t.root && t.height[i] <= maxHeight) {
			PTLoc l; l.start = start; l.end = end; l.id = i;
			pdist_counts(t.S(i), seq, start, end, l.d, l.N);
			l.dist = static_cast<double>(l.d) / l.N;
			if(std::isnan(l.dist)) hasNaN = true;
			locs.push_back(l);
		}
	}
	return locs;
}
/* the sort, the maxDiff cut and the caller's truncation, on a copy of the scan */
inline std::vector<PTLoc> orderSeeds(std::vector<PTLoc> locs, bool hasNaN, double maxDiff, int tieMode, size_t maxNSeed) {
	if(locs.empty()) return locs;
	if(tieMode == TIE_LIBSTDCXX && !hasNaN)
		std::sort(locs.begin(), locs.end());
	else
		std::sort(locs.begin(), locs.end(), [](const PTLoc& a, const PTLoc& b) {
			bool an = std::isnan(a.dist), bn = std::isnan(b.dist);
			if(an != bn) return bn;
			if(!an && a.dist != b.dist) return a.dist < b.dist;
			return a.id < b.id; });
	double bestDist = locs[0].dist, worstDist = locs[locs.size() - 1].dist;
	if(worstDist < bestDist + maxDiff) {
		size_t g = 0;
		for(; g < locs.size(); ++g) if(locs[g].dist - bestDist > maxDiff) break;
		locs.erase(locs.begin() + g, locs.end());
	}
	if(locs.size() > maxNSeed) locs.erase(locs.end() - (locs.size() - maxNSeed), locs.end());
	return locs;
}
inline std::vector<PTLoc> getSeed(const Tree& t, const int8_t* seq, int start, int end,
		double maxDiff, double maxHeight, int tieMode, size_t maxNSeed) {
	bool hasNaN;
	std::vector<PTLoc> locs = scanSeeds(t, seq, start, end, maxHeight, hasNaN);
	return orderSeeds(std::move(locs), hasNaN, maxDiff, tieMode, maxNSeed);
}

struct Placement {
	int start = 0, end = 0;
	int cNode = -1, pNode = -1, aNode = -1;
	double wuv = NAN, ratio = NAN, wnr = NAN, loglik = NAN, height = NAN, qPlace = NAN, qTaxon = NAN;
	double estLoglik = NAN; int iters = 0; /* outer iterations of the joint optimisation */
	int emIters = 0;  /* passes through the 2-node EM loop, summed over both branches and all outer iterations (diagnostic: the engine reports the same count) */
	double annoDist() const { return aNode == cNode ? wuv * ratio + wnr : (1 - ratio) * wuv + wnr; }
};

/* getLeafLoglik (src/PhyloTreeUnrooted.h:1431-1437) */
inline V4 leafLoglik(const Tree& t, const int8_t* seq, int j) {
	V4 r; int b = seq[j];
	if(b >= 0) { for(int i = 0; i < 4; ++i) r.v[i] = NEG_INF; r.v[b] = 0; }
	else for(int i = 0; i < 4; ++i) r.v[i] = std::log(t.model.pi[i]);
	return r;
}
inline void inferWeight(const double* ll, double* w) {
	double mx = max4(ll), s = 0;
	for(int i = 0; i < 4; ++i) w[i] = std::exp(ll[i] - mx);
	s = (w[0] + w[2]) + (w[1] + w[3]);
	for(int i = 0; i < 4; ++i) w[i] /= s;
}

/* PTUnrooted::estimateSeq (src/PhyloTreeUnrooted.cpp:849-877) */
inline Placement estimateSeq(const Tree& t, const int8_t* seq, const PTLoc& loc, bool weighted, double tieTol = 0) {
	const int u = (int) loc.id, v = t.parent[u];
	double cDist = loc.dist;
	double pDist = pdist(t.S(v), seq, loc.start, loc.end);
	double ratio = cDist / (cDist + pDist);
	if(std::isnan(ratio)) ratio = 0.5;
	double w0 = t.blen[u];
	double wur = w0 * ratio, wvr = w0 - wur;
	double Pu[16], Pv[16], Pn[16];
	t.model.Pr(wur, Pu); t.model.Pr(wvr, Pv);
	const int n = loc.end - loc.start + 1;
	std::vector<V4> R(n);
	double d = 0, Nw = 0;
	for(int j = loc.start; j <= loc.end; ++j) {
		V4 a = dps_mat(Pu, t.U(u, j)), b = dps_mat(Pv, t.D(u, j));
		V4& r = R[j - loc.start];
		for(int i = 0; i < 4; ++i) r.v[i] = a.v[i] + b.v[i];
		V4 nl = leafLoglik(t, seq, j);
		int b1 = argmax4_tol(r.v, tieTol), b2 = argmax4(nl.v);
		if(!weighted) { if(b1 != b2) d++; }
		else {
			double w1[4], w2[4]; inferWeight(r.v, w1); inferWeight(nl.v, w2);
			if(b1 != b2) d += w1[b1] * w2[b2];
			Nw += w1[b1] * w2[b2];
		}
	}
	double wnr = weighted ? d / Nw : d / (loc.end - loc.start + 1);
	t.model.Pr(wnr, Pn);
	double loglik = 0;
	for(int j = loc.start; j <= loc.end; ++j) {
		V4 nl = leafLoglik(t, seq, j);
		V4 c = dps_mat(Pn, nl.v);
		double X[4];
		const V4& r = R[j - loc.start];
		for(int i = 0; i < 4; ++i) X[i] = r.v[i] + c.v[i];
		loglik += dps_vec(t.model.pi, X);
	}
	Placement p;
	p.start = loc.start; p.end = loc.end; p.cNode = u; p.pNode = v; p.aNode = ratio <= 0.5 ? u : v;
	p.wuv = w0; p.ratio = ratio; p.wnr = wnr; p.loglik = loglik; p.estLoglik = loglik;
	p.height = 0; p.qPlace = 0; p.qTaxon = 0;
	return p;
}

/* filterPlacements (src/HmmUFOtu_main.cpp:162-173) */
inline void filterPlacements(std::vector<Placement>& places, double maxError) {
	if(places.empty()) return;
	std::sort(places.rbegin(), places.rend(), [](const Placement& l, const Placement& r) { return l.loglik < r.loglik; });
	double best = places[0].loglik;
	size_t g = 0;
	for(; g < places.size(); ++g) if(best - places[g].loglik > maxError) break;
	places.erase(places.begin() + g, places.end());
}

/* message of an inner node from two incoming messages (src/PhyloTreeUnrooted.cpp:320-346) */
struct Conv { const double* P; /* dgK (or 1) matrices, 16 doubles each */ };
inline V4 nodeLoglik2(const Tree& t, const double* P1, const double* m1, const double* P2, const double* m2) {
	if(t.dgK == 0) {
		V4 a = dps_mat(P1, m1), b = dps_mat(P2, m2), r;
		for(int i = 0; i < 4; ++i) r.v[i] = (0 + a.v[i]) + b.v[i];
		return r;
	}
	V4 X[16];
	for(int k = 0; k < t.dgK; ++k) {
		V4 a = dps_mat(P1 + 16 * k, m1), b = dps_mat(P2 + 16 * k, m2);
		for(int i = 0; i < 4; ++i) X[k].v[i] = (0 + a.v[i]) + b.v[i];
	}
	return row_mean_exp_scaled(X, t.dgK);
}
inline void catP(const Tree& t, double len, double* P) {
	if(t.dgK == 0) t.model.Pr(len * 1.0, P);
	else for(int k = 0; k < t.dgK; ++k) t.model.Pr(len * t.dgR[k], P + 16 * k);
}

/* 2-node Felsenstein EM (src/PhyloTreeUnrooted.cpp:749-798); U/V are n messages of 4 */
inline double optimizeBranchLength2(const Tree& t, const V4* U, const V4* V, int n, double w0, double maxL, int* steps = nullptr) {
	double q0 = std::exp(-w0), p0 = 1 - q0, p = p0, q = q0;
	const double* pi = t.model.pi;
	std::vector<double> eA(n), eB(n);
	std::vector<char> ok(n);
	for(int j = 0; j < n; ++j) {
		double s[4];
		for(int i = 0; i < 4; ++i) s[i] = U[j].v[i] + V[j].v[i];
		double logA = dps_vec(pi, s);
		double logB = dps_vec(pi, U[j].v) + dps_vec(pi, V[j].v);
		ok[j] = !(std::isnan(logA) || std::isnan(logB));
		double scale = std::max(logA, logB);
		eA[j] = std::exp(logA - scale); eB[j] = std::exp(logB - scale);
	}
	for(int iter = 0; iter < MAX_ITER && p >= 0 && p <= 1; ++iter) {
		if(steps) ++*steps;
		p = 0; int N = 0;
		for(int j = 0; j < n; ++j) {
			if(!ok[j]) continue;
			p += eB[j] * p0 / (eA[j] * q0 + eB[j] * p0);
			N++;
		}
		p /= N; q = 1 - p;
		if(std::fabs(std::log(q) - std::log(q0)) < BRANCH_EPS) break;
		p0 = p; q0 = q;
	}
	double w = -std::log(q);
	if(w > maxL) w = maxL;
	return w;
}

/* const placeSeq -> copySubTree -> mutable placeSeq -> joint optimizeBranchLength
 * (src/PhyloTreeUnrooted.cpp:925-954, 721-747, 879-923, 800-847).  The dead r->v message
 * of each outer iteration (:829-833) is not computed: nothing reads it. */
/* fixRoot: NOT the reference — the value placeSeq evidently meant to return: loglik(r, j) from the three
 * children at the optimised lengths (src/PhyloTreeUnrooted.cpp:320-346 with dGamma), then treeLoglik */
inline V4 nodeLoglik3(const Tree& t, const double* P1, const double* m1, const double* P2, const double* m2, const double* P3, const double* m3) {
	const int Kc = t.dgK > 0 ? t.dgK : 1;
	V4 X[16];
	for(int k = 0; k < Kc; ++k) {
		V4 a = dps_mat(P1 + 16 * k, m1), b = dps_mat(P2 + 16 * k, m2), c = dps_mat(P3 + 16 * k, m3);
		for(int i = 0; i < 4; ++i) X[k].v[i] = ((0 + a.v[i]) + b.v[i]) + c.v[i];
	}
	return t.dgK == 0 ? X[0] : row_mean_exp_scaled(X, t.dgK);
}
inline void placeSeq(const Tree& t, const int8_t* seq, Placement& place, double maxHeight, bool fixRoot = false) {
	const int u = place.cNode, v = place.pNode;
	const int start = place.start, end = place.end, n = end - start + 1;
	const double ratio0 = place.ratio, wnr0in = place.wnr;
	const double w0 = t.blen[u];
	double lenUR = w0 * ratio0, lenVR = w0 * (1 - ratio0), lenNR = wnr0in;
	std::vector<V4> Um(n), Vm(n), Nm(n), RN(n), RU(n);
	for(int j = 0; j < n; ++j) {
		for(int i = 0; i < 4; ++i) { Um[j].v[i] = t.U(u, start + j)[i]; Vm[j].v[i] = t.D(u, start + j)[i]; }
		Nm[j] = leafLoglik(t, seq, start + j);
	}
	double wur0 = lenUR, wvr0 = lenVR, wnr0 = lenNR;
	const double w0j = wur0 + wvr0;
	double wur = wur0, wvr = wvr0, wnr = wnr0;
	double PU[16 * 16], PV[16 * 16], PN[16 * 16];
	int iter = 0, em = 0;
	for(; iter < MAX_ITER && 0 <= wur && wur <= w0j; ++iter) {
		catP(t, lenUR, PU); catP(t, lenVR, PV);
		for(int j = 0; j < n; ++j) RN[j] = nodeLoglik2(t, PU, Um[j].v, PV, Vm[j].v);
		wnr = optimizeBranchLength2(t, RN.data(), Nm.data(), n, lenNR, 1, &em);
		lenNR = wnr;
		catP(t, lenNR, PN);
		for(int j = 0; j < n; ++j) RU[j] = nodeLoglik2(t, PV, Vm[j].v, PN, Nm[j].v);
		wur = optimizeBranchLength2(t, RU.data(), Um.data(), n, lenUR, w0j, &em);
		lenUR = wur;
		wvr = w0j - wur;
		lenVR = wvr;
		if(std::fabs(wur - wur0) < BRANCH_EPS && std::fabs(wnr - wnr0) < BRANCH_EPS) { ++iter; break; }
		wur0 = wur; wvr0 = wvr; wnr0 = wnr;
	}
	/* initRootLoglik() leaves the root message at INVALID_LOGLIK = 1; loglik(r,j) is
	 * discarded; treeLoglik(start,end) sums dot_product_scaled(pi, ones) */
	double ones[4] = { INVALID_LOGLIK, INVALID_LOGLIK, INVALID_LOGLIK, INVALID_LOGLIK };
	double ll = 0;
	for(int j = start; j <= end; ++j) ll += dps_vec(t.model.pi, ones);
	if(fixRoot) {
		catP(t, lenUR, PU); catP(t, w0j - lenUR, PV); catP(t, lenNR, PN);
		ll = 0;
		for(int j = 0; j < n; ++j) { V4 r = nodeLoglik3(t, PU, Um[j].v, PV, Vm[j].v, PN, Nm[j].v); ll += dps_vec(t.model.pi, r.v); }
	}
	place.loglik = ll;
	place.wnr = lenNR;
	double wurF = lenUR;
	place.ratio = wurF / w0;
	place.height = t.height[u] + wurF;
	if(place.ratio <= 0.5 || t.height[v] > maxHeight) place.aNode = u; else place.aNode = v;
	place.iters = iter; place.emIters = em;
}

inline double add_scaled(double a, double b) { double s = std::max(a, b); return std::log(std::exp(a - s) + std::exp(b - s)) + s; }
inline double p2q(double p) { return -10 * std::log(p) / std::log(10.0); }

/* calcQValues (src/HmmUFOtu_main.cpp:182-216); prior: 0 UNIFORM, 1 HEIGHT */
inline void calcQValues(const Tree& t, std::vector<Placement>& places, int prior) {
	if(places.empty()) return;
	const size_t n = places.size();
	std::vector<double> pp(n);
	std::vector<std::pair<int, double>> tax; /* taxon class -> logP */
	double norm = NEG_INF;
	for(size_t i = 0; i < n; ++i) {
		const Placement& pl = places[i];
		double logPrior = prior == 0 ? -0.0 : -(pl.annoDist() - pl.wnr + pl.height);
		double p = pl.loglik + logPrior;
		pp[i] = p;
		int key = t.annoId ? t.annoId[pl.aNode] : pl.aNode;
		bool found = false;
		for(auto& kv : tax) if(kv.first == key) { kv.second = add_scaled(kv.second, p); found = true; break; }
		if(!found) tax.push_back({key, p});
		norm = add_scaled(norm, p);
	}
	double mx = pp[0];
	for(size_t i = 1; i < n; ++i) mx = std::max(mx, pp[i]);
	std::vector<double> pr(n);
	double s = 0;
	for(size_t i = 0; i < n; ++i) { pr[i] = std::exp(pp[i] - mx); s += pr[i]; }
	for(size_t i = 0; i < n; ++i) {
		pr[i] /= s;
		double q = p2q(1 - pr[i]);
		places[i].qPlace = q > MAX_Q ? MAX_Q : q;
	}
	for(size_t i = 0; i < n; ++i) {
		int key = t.annoId ? t.annoId[places[i].aNode] : places[i].aNode;
		double tp = 0;
		for(auto& kv : tax) if(kv.first == key) tp = kv.second;
		double q = p2q(1 - std::exp(tp - norm));
		places[i].qTaxon = q > MAX_Q ? MAX_Q : q;
	}
}

struct AssignOpts {
	double maxDiff = std::numeric_limits<double>::infinity();
	double maxHeight = std::numeric_limits<double>::infinity();
	int maxNSeed = 50;
	double maxError = 20;
	int weighted = 0;
	int onlyML = 0;
	int prior = 0;
	int tieMode = TIE_LIBSTDCXX;       /* the reference's: literal std::sort on dist alone */
	int fixRootLoglik = 0;
	double tieTol = 0;     /* > 0: argmax4_tol in estimateSeq (test aid, see there); 0 = the reference's rule */
};

/* the SEP part of the per-read task (src/hmmufotu.cpp:641-647,720-733); seq = DigitalSeq of
 * the alignment, [start,end] 0-based inclusive */
inline std::vector<Placement> assignSeq(const Tree& t, const int8_t* seq, int start, int end, const AssignOpts& o,
		std::vector<PTLoc>* seedsOut = nullptr, std::vector<Placement>* estOut = nullptr, std::vector<int>* filtOut = nullptr) {
	std::vector<PTLoc> seeds = getSeed(t, seq, start, end, o.maxDiff, o.maxHeight, o.tieMode, (size_t) o.maxNSeed);
	if(seedsOut) *seedsOut = seeds;
	std::vector<Placement> places;
	for(const PTLoc& l : seeds) places.push_back(estimateSeq(t, seq, l, o.weighted != 0, o.tieTol));
	if(estOut) *estOut = places;
	filterPlacements(places, o.maxError);
	if(filtOut) { filtOut->clear(); for(const Placement& p : places) filtOut->push_back(p.cNode); }
	for(Placement& p : places) placeSeq(t, seq, p, o.maxHeight, o.fixRootLoglik != 0);
	if(o.onlyML)
		std::sort(places.rbegin(), places.rend(), [](const Placement& l, const Placement& r) { return l.loglik < r.loglik; });
	else {
		calcQValues(t, places, o.prior);
		std::sort(places.rbegin(), places.rend(), [](const Placement& l, const Placement& r) { return l.qPlace < r.qPlace; });
	}
	return places;
}

/* ---- chimera check (-C) of the per-read task (src/hmmufotu.cpp:653-691) ----
 * The read's region is cut into numSeg equal segments (integer division; trailing columns belong to no
 * segment); every segment is estimated/filtered/placed on the read's COMMON seeds with the distance
 * re-measured inside the segment; the 5' and 3' halves pool their placements, each pool is sorted by the
 * placed loglik with the same std::sort call, and the two winners are re-scored on each other's branch.
 * Because of F4 every placed loglik is segLen * log(sum pi e), so the pools tie completely (order =
 * std::sort's tie permutation of the pooled sequence) and the log-odds come out as exactly 0. */
struct ChimeraResult { bool checked = false, isChimera = false; double lod = NAN; Placement seg5, seg3, alt5, alt3; size_t n5 = 0, n3 = 0; };
inline ChimeraResult chimeraCheck(const Tree& t, const int8_t* seq, int start, int end, const std::vector<PTLoc>& seeds,
		const AssignOpts& o, int numSeg, double maxChimeraError, double minChimeraLod) {
	ChimeraResult res;
	const int segLen = (end - start + 1) / numSeg;
	if(seeds.empty() || segLen < 1) return res; /* the reference indexes an empty vector here (undefined); reported as "not checked" */
	auto byLoglik = [](const Placement& l, const Placement& r) { return l.loglik < r.loglik; };
	std::vector<Placement> seg5, seg3;
	for(int n = 0; n < numSeg; ++n) {
		const int s0 = start + n * segLen, e0 = s0 + segLen - 1;
		std::vector<Placement> segPlaces;
		for(const PTLoc& s : seeds) {
			PTLoc l; l.start = s0; l.end = e0; l.id = s.id;
			pdist_counts(seq, t.S((int) s.id), s0, e0, l.d, l.N);
			l.dist = static_cast<double>(l.d) / l.N;
			segPlaces.push_back(estimateSeq(t, seq, l, o.weighted != 0, o.tieTol));
		}
		filterPlacements(segPlaces, maxChimeraError);
		for(Placement& p : segPlaces) placeSeq(t, seq, p, o.maxHeight, o.fixRootLoglik != 0);
		std::vector<Placement>& pool = n < numSeg / 2 ? seg5 : seg3;
		pool.insert(pool.end(), segPlaces.begin(), segPlaces.end());
	}
	res.n5 = seg5.size(); res.n3 = seg3.size();
	std::sort(seg5.rbegin(), seg5.rend(), byLoglik);
	std::sort(seg3.rbegin(), seg3.rend(), byLoglik);
	res.seg5 = seg5[0]; res.seg3 = seg3[0];
	PTLoc a5; a5.start = res.seg5.start; a5.end = res.seg5.end; a5.id = res.seg3.cNode; /* seg3's branch, distance to seg5's own node */
	pdist_counts(seq, t.S(res.seg5.cNode), a5.start, a5.end, a5.d, a5.N); a5.dist = static_cast<double>(a5.d) / a5.N;
	res.alt5 = estimateSeq(t, seq, a5, o.weighted != 0, o.tieTol);
	placeSeq(t, seq, res.alt5, o.maxHeight, o.fixRootLoglik != 0);
	PTLoc a3; a3.start = res.seg3.start; a3.end = res.seg3.end; a3.id = res.seg5.cNode;
	pdist_counts(seq, t.S(res.seg3.cNode), a3.start, a3.end, a3.d, a3.N); a3.dist = static_cast<double>(a3.d) / a3.N;
	res.alt3 = estimateSeq(t, seq, a3, o.weighted != 0, o.tieTol);
	placeSeq(t, seq, res.alt3, o.maxHeight, o.fixRootLoglik != 0);
	res.lod = res.seg5.loglik - res.alt5.loglik + res.seg3.loglik - res.alt3.loglik;
	res.isChimera = res.seg5.aNode != res.seg3.aNode && res.lod > minChimeraLod; /* getTaxonId() = aNode id (src/PhyloTreeUnrooted.h:430-435) */
	res.checked = true;
	return res;
}

/* ---- tree pre-evaluation (what hmmufotu-build stores in .ptu) ----
 * messages for every directed edge by post-order + pre-order passes, equivalent to the
 * reference's "re-root at every node and evaluate" loop (src/hmmufotu-build.cpp:454-459,
 * src/PhyloTreeUnrooted.cpp:320-374); ancestral sequences by per-site argmax of the
 * node->parent message (src/PhyloTreeUnrooted.cpp:1085-1093); heights = min distance to
 * a descendant leaf (src/PhyloTreeUnrooted.cpp:274-287). */
inline void treeEvaluate(int nNodes, int csLen, const int* parent, const double* blen, int8_t* seq,
		const Model& model, int dgK, const double* dgR, double* up, double* down, double* rootMsg, double* height) {
	std::vector<std::vector<int>> children(nNodes);
	int root = -1;
	for(int i = 0; i < nNodes; ++i) { if(parent[i] < 0) root = i; else children[parent[i]].push_back(i); }
	std::vector<int> order; order.reserve(nNodes); /* pre-order */
	{ std::vector<int> st{root}; while(!st.empty()) { int u = st.back(); st.pop_back(); order.push_back(u); for(int c : children[u]) st.push_back(c); } }
	const int Kc = dgK > 0 ? dgK : 1;
	std::vector<double> P((size_t) nNodes * Kc * 16);
	for(int i = 0; i < nNodes; ++i) if(i != root)
		for(int k = 0; k < Kc; ++k) model.Pr(blen[i] * (dgK > 0 ? dgR[k] : 1.0), &P[((size_t) i * Kc + k) * 16]);
	auto msg = [&](double* base, int node, int j) { return base + ((size_t) node * csLen + j) * 4; };
	double logpi[4]; for(int i = 0; i < 4; ++i) logpi[i] = std::log(model.pi[i]);
	/* combine a list of (P-set, incoming message) contributions at an inner node */
	auto combine = [&](const std::vector<std::pair<const double*, const double*>>& in, bool isLeaf, int leafBase, double* out) {
		V4 X[16];
		for(int k = 0; k < Kc; ++k) for(int i = 0; i < 4; ++i) X[k].v[i] = 0;
		for(auto& pm : in) for(int k = 0; k < Kc; ++k) { V4 c = dps_mat(pm.first + 16 * k, pm.second); for(int i = 0; i < 4; ++i) X[k].v[i] += c.v[i]; }
		V4 r;
		if(!isLeaf && dgK > 0) r = row_mean_exp_scaled(X, Kc);
		else if(dgK > 0) { for(int i = 0; i < 4; ++i) r.v[i] = 0; }
		else r = X[0];
		if(isLeaf) for(int i = 0; i < 4; ++i) r.v[i] += leafBase >= 0 ? (i == leafBase ? 0.0 : NEG_INF) : logpi[i];
		for(int i = 0; i < 4; ++i) out[i] = r.v[i];
	};
	for(int oi = nNodes - 1; oi >= 0; --oi) { /* post-order: up messages */
		int u = order[oi];
		bool leaf = children[u].empty();
		for(int j = 0; j < csLen; ++j) {
			std::vector<std::pair<const double*, const double*>> in;
			for(int c : children[u]) in.push_back({&P[(size_t) c * Kc * 16], msg(up, c, j)});
			double* out = u == root ? rootMsg + (size_t) j * 4 : msg(up, u, j);
			combine(in, leaf, leaf ? seq[(size_t) u * csLen + j] : -1, out);
		}
	}
	for(int u : order) { /* pre-order: down messages parent->u exclude u's own contribution */
		if(u == root) continue;
		int p = parent[u];
		for(int j = 0; j < csLen; ++j) {
			std::vector<std::pair<const double*, const double*>> in;
			if(p != root) in.push_back({&P[(size_t) p * Kc * 16], msg(down, p, j)});
			for(int c : children[p]) if(c != u) in.push_back({&P[(size_t) c * Kc * 16], msg(up, c, j)});
			combine(in, false, -1, msg(down, u, j));
		}
	}
	for(int u = 0; u < nNodes; ++u) { /* ancestral sequences */
		if(children[u].empty()) continue;
		for(int j = 0; j < csLen; ++j)
			seq[(size_t) u * csLen + j] = (int8_t) argmax4(u == root ? rootMsg + (size_t) j * 4 : msg(up, u, j));
	}
	for(int i = 0; i < nNodes; ++i) height[i] = -1;
	for(int l = 0; l < nNodes; ++l) {
		if(!children[l].empty()) continue;
		double h = 0;
		for(int node = l; node >= 0; node = parent[node]) {
			if(height[node] < 0 || h < height[node]) height[node] = h;
			if(parent[node] >= 0) h += blen[node];
		}
	}
}

} // namespace orc
