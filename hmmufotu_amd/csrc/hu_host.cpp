// Host-side preparation: substitution-model spectral forms, profile post-load chain,
// readers of the reference's on-disk formats.  No device code in this file.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <map>
#include <sstream>
#include <memory>
#include <algorithm>
#include <new>
#include <stdexcept>
#include <sched.h>
#include <atomic>
#include <thread>
#include "hu_common.h"

static thread_local char g_err[512] = "";

void hu_set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}
extern "C" const char* hu_last_error(void) { return g_err; }

/* ---- CPU budget (hu_common.h) */
static int cpu_budget_compute() {
	if(const char* e = getenv("HU_CPU_BUDGET")) { const int v = atoi(e); if(v >= 1) return v; }
	long best = (long) std::thread::hardware_concurrency();
	if(best < 1) best = 1;
	{ cpu_set_t set; CPU_ZERO(&set); if(sched_getaffinity(0, sizeof(set), &set) == 0) { const long c = CPU_COUNT(&set); if(c >= 1 && c < best) best = c; } }
	{ /* cgroup v2: "<quota> <period>" or "max <period>" */
		FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
		if(f) { char q[64]; long per = 0; if(fscanf(f, "%63s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) { const long c = (atol(q) + per - 1) / per; if(c >= 1 && c < best) best = c; } fclose(f); }
	}
	{ /* cgroup v1 */
		FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"); FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
		long q = -1, per = 0;
		if(fq && fp && fscanf(fq, "%ld", &q) == 1 && fscanf(fp, "%ld", &per) == 1 && q > 0 && per > 0) { const long c = (q + per - 1) / per; if(c >= 1 && c < best) best = c; }
		if(fq) fclose(fq);
		if(fp) fclose(fp);
	}
	/* one process per GPU under a launcher (torch.distributed.run sets LOCAL_WORLD_SIZE): the ranks of a node share its CPUs */
	if(const char* e = getenv("LOCAL_WORLD_SIZE")) { const int w = atoi(e); if(w > 1) best = std::max<long>(1, best / w); }
	return (int) best;
}
int hu_cpu_budget() { static const int b = cpu_budget_compute(); return b; }
/* helpers of ALL pools together: the budget less one, because every caller of a pool works itself (the other callers — one thread per batch in
 * flight — sleep on their stream while a batch's host stage runs: hipDeviceScheduleBlockingSync, hu_engine.hip) */
static std::atomic<int>& helper_tokens() { static std::atomic<int> t{std::max(0, hu_cpu_budget() - 1)}; return t; }
int hu_helpers_acquire(int want) {
	if(want <= 0) return 0;
	std::atomic<int>& t = helper_tokens();
	int have = t.load();
	for(;;) {
		const int take = std::min(want, std::max(have, 0));
		if(take == 0) return 0;
		if(t.compare_exchange_weak(have, have - take)) return take;
	}
}
void hu_helpers_release(int n) { if(n > 0) helper_tokens().fetch_add(n); }

int hu_catch_all(const char* fn) noexcept {
	try { throw; }
	catch(const std::bad_alloc&) { hu_set_error("%s: out of host memory (std::bad_alloc)", fn); return HU_ERR_NOMEM; }
	catch(const std::length_error& e) { hu_set_error("%s: a size the host cannot hold (%s)", fn, e.what()); return HU_ERR_NOMEM; }
	catch(const std::exception& e) { hu_set_error("%s: %s", fn, e.what()); return HU_ERR_STATE; }
	catch(...) { hu_set_error("%s: unknown exception", fn); return HU_ERR_STATE; }
}

/* ------------------------------------------------------------------------------------------------
 * The first k places of std::sort, for HU_SEED_ORDER_LIBSTDCXX.
 *
 * The reference keeps the first maxNSeed elements of std::sort(locs.begin(), locs.end()) with operator< on dist alone
 * (src/HmmUFOtu_main.cpp:139, src/hmmufotu.cpp:646-647; operator< src/PhyloTreeUnrooted.h:1623-1625): which of the nodes tying at the
 * cut-off distance are kept, and the order of equal distances inside the list, is the permutation libstdc++'s introsort leaves — a
 * deterministic function of the sequence of comparison results, hence of the keys in node order.  Restated here from the published
 * algorithm of libstdc++'s <bits/stl_algo.h> (std::__sort, GCC 4.9 - 13; the reference's distribution builds with the system's g++):
 *   introsort loop on [first, last) while it holds more than 16 elements: depth limit 2 floor(lg n), decremented per partition — at 0 the
 *   range is heap-sorted (make_heap + sort_heap); pivot = median of first + 1, first + (last - first) / 2, last - 1 swapped into *first;
 *   unguarded Hoare partition of [first + 1, last) against *first (both scans stop on elements EQUAL to the pivot, which are swapped);
 *   recursion into [cut, last), loop on [first, cut); then ONE insertion sort over everything (strict comparisons: stable on what the
 *   partitions left).
 * Only what can reach the first k places is executed: a range that starts at or beyond place k is never partitioned (its elements are
 * >= everything left of it and the strict insertion sort cannot carry them across its start), so the cost is ~2n element visits
 * instead of n lg n.  tests/test_seed_order.py checks it against the literal std::sort of the oracle on tie-heavy inputs of every size.
 * Elements are (key << 24 | index) with key an order-isomorphic integer image of dist; comparisons look at the key alone. */
namespace {
struct PrefixSort {
	uint64_t* a; size_t k;
	static bool less(uint64_t x, uint64_t y) { return (x >> 24) < (y >> 24); }
	void median_to_first(uint64_t* result, uint64_t* x, uint64_t* y, uint64_t* z) {
		if(less(*x, *y)) { if(less(*y, *z)) std::swap(*result, *y); else if(less(*x, *z)) std::swap(*result, *z); else std::swap(*result, *x); }
		else if(less(*x, *z)) std::swap(*result, *x);
		else if(less(*y, *z)) std::swap(*result, *z);
		else std::swap(*result, *y);
	}
	uint64_t* partition(uint64_t* first, uint64_t* last, const uint64_t* pivot) {
		for(;;) {
			while(less(*first, *pivot)) ++first;
			--last;
			while(less(*pivot, *last)) --last;
			if(!(first < last)) return first;
			std::swap(*first, *last);
			++first;
		}
	}
	void loop(uint64_t* first, uint64_t* last, long depth) {
		while(last - first > 16) {
			if(depth == 0) { std::make_heap(first, last, less); std::sort_heap(first, last, less); return; }    /* __partial_sort(first, last, last) */
			--depth;
			uint64_t* mid = first + (last - first) / 2;
			median_to_first(first, first + 1, mid, last - 1);
			uint64_t* cut = partition(first + 1, last, first);
			if(cut < a + k) loop(cut, last, depth);       /* a range beyond the first k places cannot change them */
			last = cut;
		}
	}
	void run(size_t n) {
		if(n == 0) return;
		long lg = 0; for(size_t m = n; m > 1; m >>= 1) ++lg;
		loop(a, a + n, 2 * lg);
		/* __final_insertion_sort over the part whose blocks are final: everything up to the end of the block that holds place k - 1.
		 * The first unpartitioned range starts at or beyond k; blocks of <= 16 elements lie before it.  Sorting a few elements more
		 * than needed is harmless, fewer would not be: take [0, min(n, k + 16)) — a block reaching across k ends within 16 of it...
		 * unless it is a heap-sorted or skipped range, whose elements an insertion sort leaves where they are relative to [0, k). */
		const size_t e = std::min(n, k + 16);
		for(size_t i = 1; i < e; ++i) {
			const uint64_t v = a[i]; size_t j = i;
			while(j > 0 && less(v, a[j - 1])) { a[j] = a[j - 1]; --j; }
			a[j] = v;
		}
	}
};
}
/* a [n] packed (key << 24 | index); on return a[0 .. min(k, n)) are the first places of std::sort on the keys */
void hu_sort_prefix_packed(uint64_t* a, size_t n, size_t k) { PrefixSort s{a, k}; s.run(n); }

extern "C" int hu_sort_prefix_libstdcxx(const double* dist, int64_t n, int64_t k, int32_t* out_idx) try {
	if(n < 0 || k < 0 || (n > 0 && (!dist || !out_idx)) || n >= (1ll << 24)) { hu_set_error("hu_sort_prefix_libstdcxx: bad argument"); return HU_ERR_ARG; }
	/* order-isomorphic integer keys: the rank of each distinct value (40 bits are plenty for < 2^24 elements) */
	std::vector<double> vals(dist, dist + n);
	for(int64_t i = 0; i < n; ++i) if(std::isnan(vals[i])) { hu_set_error("hu_sort_prefix_libstdcxx: NaN distance (std::sort is undefined on it)"); return HU_ERR_ARG; }
	std::sort(vals.begin(), vals.end());
	vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
	std::vector<uint64_t> a((size_t) n);
	for(int64_t i = 0; i < n; ++i) a[i] = ((uint64_t)(std::lower_bound(vals.begin(), vals.end(), dist[i]) - vals.begin()) << 24) | (uint64_t) i;
	hu_sort_prefix_packed(a.data(), (size_t) n, (size_t) k);
	for(int64_t i = 0; i < std::min(n, k); ++i) out_idx[i] = (int32_t)(a[i] & 0xffffffu);
	return HU_OK;
} catch(...) { return hu_catch_all("hu_sort_prefix_libstdcxx"); }

extern "C" void hu_default_opts(hu_opts* o) {
	o->align_mode = HU_MODE_GLOBAL;
	o->max_nseed = 50;
	o->max_diff = std::numeric_limits<double>::infinity();
	o->max_height = std::numeric_limits<double>::infinity();
	o->max_error = 20;
	o->weighted = 0;
	o->only_ml = 0;
	o->prior = HU_PRIOR_UNIFORM;
	o->fix_root_loglik = 0;
	o->seed_order = HU_SEED_ORDER_LIBSTDCXX;      /* the reference's own std::sort order (src/HmmUFOtu_main.cpp:139) */
	o->ignore_orient = 0;
}

/* ------------------------------------------------------------------ substitution models */
static void sym_eig4(double A[16], double V[16], double w[4]) {
	for(int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
	for(int sweep = 0; sweep < 64; ++sweep) {
		double off = 0;
		for(int p = 0; p < 4; ++p) for(int q = p + 1; q < 4; ++q) off += A[p*4+q] * A[p*4+q];
		if(off < 1e-300) break;
		for(int p = 0; p < 4; ++p) for(int q = p + 1; q < 4; ++q) {
			double apq = A[p*4+q];
			if(apq == 0) continue;
			double theta = (A[q*4+q] - A[p*4+p]) / (2 * apq);
			double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
			double c = 1 / std::sqrt(t * t + 1), s = t * c;
			for(int k = 0; k < 4; ++k) { double a = A[k*4+p], b = A[k*4+q]; A[k*4+p] = c * a - s * b; A[k*4+q] = s * a + c * b; }
			for(int k = 0; k < 4; ++k) { double a = A[p*4+k], b = A[q*4+k]; A[p*4+k] = c * a - s * b; A[q*4+k] = s * a + c * b; }
			for(int k = 0; k < 4; ++k) { double a = V[k*4+p], b = V[k*4+q]; V[k*4+p] = c * a - s * b; V[k*4+q] = s * a + c * b; }
		}
	}
	for(int i = 0; i < 4; ++i) w[i] = A[i*4+i];
}

/* Rate matrix of each model type as implied by the reference's Pr(t):
 *   GTR   Q = R*diag(pi), diagonal = -rowsum, divided by -trace(Q)   (src/GTR.cpp:124-131; note
 *         DNASubModel::scale is called with its default pi = Ones(), src/DNASubModel.h:154)
 *   TN93  Q(i,j) = beta pi(j) k(i,j), k = kr for A<->G, ky for C<->T, 1 otherwise (src/TN93.h:113-154)
 *   HKY85 kr = ky = kappa (src/HKY85.h:111-153); F81 k = 1 (src/F81.h:110-118)
 *   K80   pi = 1/4, transitions kappa*beta, transversions beta, beta = 1/(2 kappa) (src/K80.h:98-118)
 *   JC69  pi = 1/4, every rate 1/3 (src/JC69.h:97-101) */
int hu_model_prepare(const hu_model_desc* d, HuModelDev* out) {
	memset(out, 0, sizeof(*out));
	out->type = d->type;
	double pi[4], Q[16];
	for(int i = 0; i < 4; ++i) pi[i] = (d->type == HU_K80 || d->type == HU_JC69) ? 0.25 : d->pi[i];
	for(int i = 0; i < 4; ++i) if(!(pi[i] > 0)) { hu_set_error("model: base frequency %d is not positive", i); return HU_ERR_ARG; }
	auto ti = [](int i, int j) { return (i ^ j) == 2; }; /* A(0)<->G(2), C(1)<->T(3) */
	switch(d->type) {
	case HU_GTR: {
		for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j) Q[i*4+j] = i == j ? 0.0 : d->par[i*4+j] * pi[j];
		double tr = 0;
		for(int i = 0; i < 4; ++i) { double s = 0; for(int j = 0; j < 4; ++j) if(j != i) s += Q[i*4+j]; Q[i*4+i] = -s; tr += -s; }
		for(int i = 0; i < 16; ++i) Q[i] = Q[i] / -tr;
		break;
	}
	case HU_TN93: case HU_HKY85: case HU_F81: {
		double kr = 1, ky = 1, beta;
		if(d->type == HU_TN93) { kr = d->par[0]; ky = d->par[1]; beta = d->par[2]; }
		else if(d->type == HU_HKY85) { kr = ky = d->par[0]; beta = d->par[1]; }
		else beta = d->par[0];
		for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j)
			Q[i*4+j] = i == j ? 0.0 : beta * pi[j] * (ti(i, j) ? ((i == 0 || i == 2) ? kr : ky) : 1.0);
		break;
	}
	case HU_K80: {
		double kappa = d->par[0], beta = 1 / (2 * kappa);
		for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j) Q[i*4+j] = i == j ? 0.0 : (ti(i, j) ? kappa * beta : beta);
		break;
	}
	case HU_JC69:
		for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j) Q[i*4+j] = i == j ? 0.0 : 1.0 / 3;
		break;
	default:
		hu_set_error("model: unknown type %d", d->type);
		return HU_ERR_ARG;
	}
	if(d->type != HU_GTR)
		for(int i = 0; i < 4; ++i) { double s = 0; for(int j = 0; j < 4; ++j) if(j != i) s += Q[i*4+j]; Q[i*4+i] = -s; }
	double S[16], V[16], sq[4], isq[4];
	for(int i = 0; i < 4; ++i) { sq[i] = std::sqrt(pi[i]); isq[i] = 1 / sq[i]; }
	for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j) S[i*4+j] = sq[i] * Q[i*4+j] * isq[j];
	for(int i = 0; i < 4; ++i) for(int j = i + 1; j < 4; ++j) { double a = 0.5 * (S[i*4+j] + S[j*4+i]); S[i*4+j] = S[j*4+i] = a; }
	sym_eig4(S, V, out->lam);
	/* the zero eigenvalue and its eigenvector sqrt(pi) are exact in theory: moved to index 0 and snapped, so that
	 * P(t) rows keep summing to 1 for huge t, U(i,0) = 1, U1(0,i) = pi(i): component 0 of a message in the
	 * eigenbasis is pi . e (k_place_blk normalises messages by it and keeps the other three components) */
	int z = 0;
	for(int k = 1; k < 4; ++k) if(std::fabs(out->lam[k]) < std::fabs(out->lam[z])) z = k;
	if(z != 0) { std::swap(out->lam[0], out->lam[z]); for(int i = 0; i < 4; ++i) std::swap(V[i*4+0], V[i*4+z]); }
	out->lam[0] = 0.0;
	for(int i = 0; i < 4; ++i) V[i*4+0] = sq[i];
	for(int i = 0; i < 4; ++i) for(int k = 0; k < 4; ++k) { out->U[i*4+k] = isq[i] * V[i*4+k]; out->U1[k*4+i] = V[i*4+k] * sq[i]; }
	for(int i = 0; i < 4; ++i) { out->U[i*4+0] = 1.0; out->U1[0*4+i] = pi[i]; }
	for(int i = 0; i < 4; ++i) { out->pi[i] = pi[i]; out->logpi[i] = std::log(pi[i]); }
	out->dgK = d->dg_k;
	if(d->dg_k < 0 || d->dg_k > HU_MAX_DGK) { hu_set_error("model: dg_k %d out of range", d->dg_k); return HU_ERR_ARG; }
	out->rate[0] = 1.0;
	for(int k = 0; k < d->dg_k; ++k) out->rate[k] = d->dg_rate[k];
	return HU_OK;
}

/* constants of k_place_blk: with messages in the eigenbasis (a = U^-1 e) every per-site quantity of the joint
 * branch-length optimisation is a bilinear form a_U^T M a_V; the model-dependent factors of those 4x4 tables: */
void hu_place_consts(const HuModelDev& m, double* pc) {
	for(int i = 0; i < 4; ++i) pc[HU_PC_LAM + i] = m.lam[i];
	for(int i = 0; i < HU_MAX_DGK; ++i) pc[HU_PC_RATE + i] = m.rate[i];
	double pi2 = 0;
	for(int i = 0; i < 4; ++i) pi2 += m.pi[i] * m.pi[i];
	for(int a = 0; a < 4; ++a) for(int c = 0; c < 4; ++c) {
		for(int b = 0; b < 4; ++b) pc[HU_PC_W + b * 16 + a * 4 + c] = m.U[b*4+a] * m.U[b*4+c];
		double g = 0;
		for(int i = 0; i < 4; ++i) g += m.pi[i] * m.pi[i] * m.U[i*4+a] * m.U[i*4+c];
		pc[HU_PC_W + 4 * 16 + a * 4 + c] = g / pi2;
		for(int k = 0; k < 4; ++k) {
			double t = 0;
			for(int i = 0; i < 4; ++i) t += m.pi[i] * m.U[i*4+a] * m.U[i*4+c] * m.U[i*4+k];
			pc[HU_PC_C + (a * 4 + c) * 4 + k] = t;
		}
	}
	for(int k = 0; k < 4; ++k) {
		for(int b = 0; b < 4; ++b) pc[HU_PC_CB + b * 4 + k] = m.U1[k*4+b];
		double a = 0, s = 0;
		for(int i = 0; i < 4; ++i) { a += m.U1[k*4+i] * m.pi[i]; s += m.pi[i] * m.U[i*4+k]; }
		pc[HU_PC_CB + 4 * 4 + k] = a;
		pc[HU_PC_S + k] = s;
	}
}

/* host-only export for CPU tests of the spectral forms */
extern "C" int hu_model_spectral(const hu_model_desc* d, double* U, double* lam, double* U1) try {
	HuModelDev m;
	int rc = hu_model_prepare(d, &m);
	if(rc != HU_OK) return rc;
	memcpy(U, m.U, sizeof(m.U)); memcpy(lam, m.lam, sizeof(m.lam)); memcpy(U1, m.U1, sizeof(m.U1));
	return HU_OK;
} catch(...) { return hu_catch_all("hu_model_spectral"); }

/* ------------------------------------------------------------------ profile (BandedHMMP7) */
/* post-load chain of operator>> (src/BandedHMMP7.cpp:104-109): extend_index,
 * adjustProfileLocalMode (:721-733), wingRetract (:1083-1120) */
int HuProfileHost::init(const hu_profile_desc* d) {
	if(d->K < 1 || d->L < d->K || d->L > 65535) { hu_set_error("profile: bad sizes K=%d L=%d", d->K, d->L); return HU_ERR_ARG; }
	K = d->K; L = d->L;
	EM.assign(d->EM, d->EM + 4 * (K + 1));
	EI.assign(d->EI, d->EI + 4 * (K + 1));
	T7.assign(d->T, d->T + 7 * (K + 1));
	p2cs.assign(d->p2cs, d->p2cs + K + 1);
	p2cs[0] = 0;
	for(int k = 1; k <= K; ++k)
		if(p2cs[k] < 1 || p2cs[k] > L || (k > 1 && p2cs[k] <= p2cs[k - 1])) { hu_set_error("profile: MAP of column %d invalid", k); return HU_ERR_ARG; }
	cs2p.assign(L + 2, 0);
	for(int k = 1; k <= K; ++k) cs2p[p2cs[k]] = k;
	for(int i = p2cs[K] + 1; i <= L; ++i) cs2p[i] = K;
	const double* Tc = T7.data();
	std::vector<double> entry(K + 1, 0.0), exitp(K + 1, 0.0);
	double t0 = std::exp(-Tc[0]), tK = std::exp(-Tc[(size_t) K * 7]);
	for(int k = 1; k <= K; ++k) { entry[k] = t0; exitp[k] = tK; }
	{ /* B->D1->...->Dj-1->Mj folded into B->Mj; the partial sums are shared between consecutive j */
		double chain = Tc[2]; /* Tmat_cost[0](M,D) */
		for(int j = 2; j <= K; ++j) {
			if(j > 2) chain += Tc[(size_t)(j - 2) * 7 + 6];
			double cost = chain + Tc[(size_t)(j - 1) * 7 + 5];
			entry[j] += std::exp(-cost);
			if(entry[j] > 1) entry[j] = 1;
		}
	}
	for(int i = 1; i <= K - 1; ++i) { /* Mi->Di+1->...->DK->E folded into Mi->E */
		double cost = Tc[(size_t) i * 7 + 2];
		for(int j = i + 1; j < K; ++j) cost += Tc[(size_t) j * 7 + 6];
		cost += Tc[(size_t) K * 7 + 5];
		exitp[i] += std::exp(-cost);
		if(exitp[i] > 1) exitp[i] = 1;
	}
	entryC.resize(K + 1); exitC.resize(K + 1);
	for(int k = 0; k <= K; ++k) { entryC[k] = -std::log(entry[k]); exitC[k] = -std::log(exitp[k]); }
	return HU_OK;
}

/* setSequenceMode (src/BandedHMMP7.cpp:561-583) with p1 of src/BandedHMMP7Bg.cpp:33-35 */
void hu_mode_costs(int K, int mode, double* tNN, double* tNB, double* tEC, double* tCC) {
	double p1 = K >= 350 ? K / (K + 1.0) : 350 / (350 + 1.0);
	double term = 1 - p1, nn = 0, cc = 0;
	switch(mode) {
	case HU_MODE_LOCAL: nn = cc = term; break;
	case HU_MODE_NGCL: cc = term; break;
	case HU_MODE_CGNL: nn = term; break;
	default: break;
	}
	*tNN = -std::log(nn); *tNB = -std::log(1.0 - nn); *tEC = -std::log(1.0); *tCC = -std::log(cc);
}

/* ------------------------------------------------------------------ .hmm reader */
static double hmm_value(const std::string& s) { return s != "*" ? atof(s.c_str()) : std::numeric_limits<double>::infinity(); }

/* HMMER3/f-style text as BandedHMMP7's operator>> accepts it (src/BandedHMMP7.cpp:100-246) */
int hu_read_hmm(const char* path, HuProfileHost& prof, std::vector<double>& EM, std::vector<double>& EI,
		std::vector<double>& T, std::vector<int32_t>& p2cs, int& K, int& L) {
	std::ifstream in(path);
	if(!in) { hu_set_error("cannot open HMM file '%s'", path); return HU_ERR_IO; }
	return hu_read_hmm_stream(in, path, prof, EM, EI, T, p2cs, K, L);
}
int hu_read_hmm_stream(std::istream& in, const char* path, HuProfileHost& prof, std::vector<double>& EM, std::vector<double>& EI,
		std::vector<double>& T, std::vector<int32_t>& p2cs, int& K, int& L) {
	std::string line;
	K = 0; L = 0;
	int k = 0;
	bool mapYes = false, done = false;
	std::map<std::string, std::string> tags;
	while(std::getline(in, line)) {
		if(line == "//") { done = true; break; }
		if(line.empty()) continue;
		std::istringstream iss(line);
		std::string tag;
		if(!isspace((unsigned char) line[0])) {
			iss >> tag;
			if(tag.substr(0, 6) == "HMMER3") {
				if(tag.length() < 8 || tag[7] < 'f') { hu_set_error("obsolete HMM file version %s", tag.c_str()); return HU_ERR_IO; }
			}
			else if(tag == "LENG") {
				iss >> K;
				if(K < 1 || K > 65535) { hu_set_error("HMM LENG %d out of range", K); return HU_ERR_IO; }
				EM.assign((size_t) 4 * (K + 1), std::numeric_limits<double>::infinity());
				EI = EM;
				T.assign((size_t) 7 * (K + 1), std::numeric_limits<double>::infinity());
				p2cs.assign(K + 1, 0);
			}
			else if(tag == "ALPH") { std::string a; iss >> a; if(a != "DNA") { hu_set_error("HMM alphabet must be DNA"); return HU_ERR_IO; } }
			else if(tag == "MAXL") iss >> L;
			else if(tag == "HMM") { std::string skip; std::getline(in, skip); }
			else { std::string val; iss >> val; tags[tag] = val; if(tag == "MAP") mapYes = (val == "yes"); }
		}
		else {
			if(K == 0) { hu_set_error("HMM body before LENG"); return HU_ERR_IO; }
			iss >> tag;
			std::string tmp;
			bool compo = tag == "COMPO";
			int idx = 0;
			bool isInt = sscanf(tag.c_str(), "%d", &idx) == 1;
			if(k > K) { hu_set_error("HMM has more than LENG states"); return HU_ERR_IO; }
			if(compo || isInt) {
				if((compo && k != 0) || (!compo && idx != k)) { hu_set_error("HMM state line %s out of order", tag.c_str()); return HU_ERR_IO; }
				for(int i = 0; i < 4; ++i) { iss >> tmp; EM[(size_t) k * 4 + i] = hmm_value(tmp); }
				if(!compo) {
					if(!mapYes) { hu_set_error("HMM file must have the MAP flag set to 'yes'"); return HU_ERR_IO; }
					iss >> tmp; p2cs[k] = atoi(tmp.c_str());
				}
				for(int i = 0; i < 4; ++i) { in >> tmp; EI[(size_t) k * 4 + i] = hmm_value(tmp); }
				for(int i = 0; i < 7; ++i) { in >> tmp; T[(size_t) k * 7 + i] = hmm_value(tmp); }
			}
			else { /* non-COMPO begin-state line: tag is the first insert emission */
				if(k != 0) { hu_set_error("unexpected HMM line"); return HU_ERR_IO; }
				EI[0] = hmm_value(tag);
				for(int i = 1; i < 4; ++i) { iss >> tmp; EI[i] = hmm_value(tmp); }
				for(int i = 0; i < 7; ++i) { in >> tmp; T[i] = hmm_value(tmp); }
			}
			std::getline(in, tmp); /* rest of the transition line */
			k++;
		}
	}
	if(!done || k != K + 1) { hu_set_error("HMM file '%s' truncated (read %d of %d states)", path, k, K + 1); return HU_ERR_IO; }
	if(L == 0) L = p2cs[K];
	hu_profile_desc d{K, L, EM.data(), EI.data(), T.data(), p2cs.data()};
	return prof.init(&d);
}

/* ------------------------------------------------------------------ .ptu reader */
namespace {
struct Rd {
	std::ifstream in;
	bool ok = true;
	uint64_t size = 0;          /* of the file: no length field read from it may ask for more memory than that */
	void measure() { in.seekg(0, std::ios::end); const std::streamoff e = in.tellg(); size = e > 0 ? (uint64_t) e : 0; in.seekg(0, std::ios::beg); }
	template<class X> X get() { X v{}; in.read((char*) &v, sizeof(X)); if(!in) ok = false; return v; }
	std::string str() {
		uint64_t n = get<uint64_t>();
		if(!ok || n > size) { ok = false; return std::string(); }
		std::string s(n, '\0');
		if(n) in.read(&s[0], n);
		if(!in) ok = false;
		return s;
	}
};
}

/* model text block embedded in the .ptu (readers: src/GTR.cpp:43-81, TN93.cpp:40-77,
 * HKY85.cpp:40-75, F81.cpp:40-73, K80.cpp:41-71, JC69.cpp:39-65); consumes exactly the bytes
 * the reference consumes so that the binary dG block that follows stays aligned */
int hu_read_model_text(std::istream& in, hu_model_desc& m) {
	memset(&m, 0, sizeof(m));
	std::string type, tag, line, value;
	in >> type;
	in.ignore();
	static const char* names[] = {"GTR", "TN93", "HKY85", "F81", "K80", "JC69"};
	m.type = -1;
	for(int i = 0; i < 6; ++i) if(type == names[i]) m.type = i;
	if(m.type < 0) { hu_set_error("ptu: unknown model type '%s'", type.c_str()); return HU_ERR_IO; }
	for(int i = 0; i < 4; ++i) m.pi[i] = 0.25;
	while(in >> tag) {
		if(tag[0] == '#') { std::getline(in, line); continue; }
		if(tag == "Type:") {
			in >> value;
			if(value != type) { hu_set_error("ptu: model block type mismatch"); return HU_ERR_IO; }
			if(m.type == HU_JC69) { std::getline(in, line); break; }
		}
		else if(tag == "pi:") { for(int i = 0; i < 4; ++i) in >> m.pi[i]; }
		else if(tag == "R:" && m.type == HU_GTR) { for(int i = 0; i < 16; ++i) in >> m.par[i]; }
		else if(tag == "Q:" && m.type == HU_GTR) { for(int i = 0; i <= 4; ++i) std::getline(in, line); break; }
		else if(tag == "kr:" && m.type == HU_TN93) in >> m.par[0];
		else if(tag == "ky:" && m.type == HU_TN93) in >> m.par[1];
		else if(tag == "kappa:" && m.type == HU_HKY85) in >> m.par[0];
		else if(tag == "kappa:" && m.type == HU_K80) { in >> m.par[0]; std::getline(in, line); break; }
		else if(tag == "beta:" && (m.type == HU_TN93 || m.type == HU_HKY85 || m.type == HU_F81)) {
			in >> m.par[m.type == HU_TN93 ? 2 : m.type == HU_HKY85 ? 1 : 0];
			std::getline(in, line);
			break;
		}
		else { hu_set_error("ptu: unrecognised model tag '%s'", tag.c_str()); return HU_ERR_IO; }
	}
	return in ? HU_OK : HU_ERR_IO;
}

/* PTUnrooted::load (src/PhyloTreeUnrooted.cpp:496-535 and :99-114, :605-621, :632-697,
 * :1068-1083; src/DigitalSeq.cpp:105-121; src/util/ProgEnv.cpp:24-64) */
int hu_read_ptu(const char* path, HuTreeHost& t) { return hu_read_ptu_sink(path, t, nullptr); }

/* sink != NULL: the 4 x csLen messages are handed to it one directed edge at a time (is_down, node, data, root flag) instead of
 * being kept in t.up / t.down: a gg_97-scale file holds 98 GB of them, which hu_db_load sends straight on to the device */
int hu_read_ptu_sink(const char* path, HuTreeHost& t, const std::function<int(bool, int64_t, const double*)>* sink) {
	Rd r;
	r.in.open(path, std::ios::binary);
	if(!r.in) { hu_set_error("cannot open PTU file '%s'", path); return HU_ERR_IO; }
	r.measure();
	char magic[8];
	r.in.read(magic, 8);
	if(!r.in || memcmp(magic, "HmmUFOtu", 8) != 0) { hu_set_error("'%s' is not a HmmUFOtu database file", path); return HU_ERR_IO; }
	for(int i = 0; i < 3; ++i) r.get<int32_t>();
	uint64_t n = r.get<uint64_t>();
	int32_t csLen = r.get<int32_t>();
	if(!r.ok || n == 0 || n > (1u << 24) || csLen < 1 || csLen > 65535) { hu_set_error("ptu: bad header"); return HU_ERR_IO; }
	/* the file carries 2 (n - 1) messages of 4 x csLen doubles: a header that promises more than the file holds is refused before anything is sized by it */
	if((n - 1) * 2 * 32 * (uint64_t) csLen > r.size) { hu_set_error("ptu: the header names %llu nodes x %d columns, the file has %llu bytes", (unsigned long long) n, (int) csLen, (unsigned long long) r.size); return HU_ERR_IO; }
	t.n = (int32_t) n; t.csLen = csLen;
	t.parent.assign(n, -1); t.blen.assign(n, 0.0); t.height.assign(n, 0.0); t.annoDist.assign(n, 0.0);
	t.seq.assign((size_t) n * csLen, (int8_t) -2);
	t.names.resize(n); t.annos.resize(n);
	for(uint64_t i = 0; i < n; ++i) {
		int64_t id = r.get<int64_t>();
		if(!r.ok || id != (int64_t) i) { hu_set_error("ptu: node %llu has id %lld", (unsigned long long) i, (long long) id); return HU_ERR_IO; }
		t.names[i] = r.str();
		bool withAbc = r.get<uint8_t>() != 0;
		if(withAbc) r.str();
		r.str(); /* seq name */
		std::string s = r.str();
		if(!r.ok) break;
		if(!s.empty()) {
			if((int) s.size() != csLen) { hu_set_error("ptu: node %llu sequence length %zu != csLen", (unsigned long long) i, s.size()); return HU_ERR_IO; }
			memcpy(&t.seq[(size_t) i * csLen], s.data(), csLen);
		}
		t.annos[i] = r.str();
		t.annoDist[i] = r.get<double>();
	}
	uint64_t nEdges = r.get<uint64_t>();
	if(!r.ok || nEdges != 2 * (n - 1)) { hu_set_error("ptu: edge count %llu does not match %llu nodes", (unsigned long long) nEdges, (unsigned long long) n); return HU_ERR_IO; }
	std::vector<double> one;
	if(sink) one.resize((size_t) csLen * 4);
	else { t.up.assign((size_t) n * csLen * 4, 0.0); t.down.assign((size_t) n * csLen * 4, 0.0); }
	/* first pass cannot know parents before all isParent flags are seen, so keep the edge list */
	struct Edge { int64_t a, b; bool aParent; double len; std::streampos pos; };
	std::vector<Edge> edges(nEdges);
	for(uint64_t e = 0; e < nEdges; ++e) {
		Edge& E = edges[e];
		E.a = r.get<int64_t>(); E.b = r.get<int64_t>(); E.aParent = r.get<uint8_t>() != 0;
		E.len = r.get<double>();
		uint64_t N = r.get<uint64_t>();
		if(!r.ok || N != (uint64_t) 4 * csLen || E.a < 0 || E.b < 0 || E.a >= (int64_t) n || E.b >= (int64_t) n) { hu_set_error("ptu: bad edge %llu", (unsigned long long) e); return HU_ERR_IO; }
		E.pos = r.in.tellg();
		/* message of a->b: if a is the parent it is down[b], else up[a] */
		double* dst = sink ? one.data() : E.aParent ? &t.down[(size_t) E.b * csLen * 4] : &t.up[(size_t) E.a * csLen * 4];
		r.in.read((char*) dst, sizeof(double) * N);
		if(!r.in) { r.ok = false; break; }
		if(sink) { int rc = (*sink)(E.aParent, E.aParent ? E.b : E.a, dst); if(rc != HU_OK) return rc; }
		if(E.aParent) { t.parent[E.b] = (int32_t) E.a; t.blen[E.b] = E.len; }
	}
	int64_t rootId = r.get<int64_t>();
	if(!r.ok || rootId < 0 || rootId >= (int64_t) n) { hu_set_error("ptu: bad root"); return HU_ERR_IO; }
	t.root = (int32_t) rootId;
	r.in.read((char*) (sink ? one.data() : &t.up[(size_t) rootId * csLen * 4]), sizeof(double) * 4 * csLen);
	if(sink && r.in) { int rc = (*sink)(false, rootId, one.data()); if(rc != HU_OK) return rc; }
	for(uint64_t i = 0; i < n; ++i) {
		int64_t id = r.get<int64_t>(); double h = r.get<double>();
		if(!r.ok || id < 0 || id >= (int64_t) n) { hu_set_error("ptu: bad height record"); return HU_ERR_IO; }
		t.height[id] = h;
	}
	uint32_t nIdx = r.get<uint32_t>();
	for(uint32_t i = 0; i < nIdx && r.ok; ++i) { r.get<uint32_t>(); r.get<int64_t>(); }
	if(!r.ok) { hu_set_error("ptu: truncated file"); return HU_ERR_IO; }
	int rc = hu_read_model_text(r.in, t.model);
	if(rc != HU_OK) return rc;
	bool hasDG = r.get<uint8_t>() != 0;
	if(hasDG) {
		int32_t K = r.get<int32_t>();
		r.get<double>(); /* alpha */
		if(!r.ok || K < 1 || K > HU_MAX_DGK) { hu_set_error("ptu: bad discrete-Gamma block"); return HU_ERR_IO; }
		for(int i = 0; i <= K; ++i) r.get<double>();
		for(int i = 0; i < K; ++i) t.model.dg_rate[i] = r.get<double>();
		t.model.dg_k = K;
	}
	if(!r.ok) { hu_set_error("ptu: truncated file"); return HU_ERR_IO; }
	int roots = 0;
	for(uint64_t i = 0; i < n; ++i) if(t.parent[i] < 0) roots++;
	if(roots != 1 || t.parent[t.root] >= 0) { hu_set_error("ptu: tree is not rooted at a single node"); return HU_ERR_IO; }
	/* annotation classes: equal strings share an id (calcQValues groups by getTaxonName()) */
	std::map<std::string, int32_t> cls;
	t.annoId.resize(n);
	for(uint64_t i = 0; i < n; ++i) {
		auto it = cls.find(t.annos[i]);
		if(it == cls.end()) it = cls.insert({t.annos[i], (int32_t) cls.size()}).first;
		t.annoId[i] = it->second;
	}
	return HU_OK;
}

/* ---- hu_tree_info: the tree of a .ptu without its messages (include/hmmufotu_amd.h) ---- */
struct hu_tree_info { HuTreeHost t; std::vector<std::vector<int32_t>> children; };
extern "C" int hu_tree_info_load(const char* ptu_path, hu_tree_info** out) try {
	if(!ptu_path || !out) { hu_set_error("hu_tree_info_load: null argument"); return HU_ERR_ARG; }
	*out = nullptr;
	std::unique_ptr<hu_tree_info> ti(new hu_tree_info);
	std::vector<int32_t> order;           /* children in the order their parent -> child edges stand in the file */
	const std::function<int(bool, int64_t, const double*)> sink = [&](bool isDown, int64_t node, const double*) -> int { if(isDown) order.push_back((int32_t) node); return HU_OK; };
	int rc = hu_read_ptu_sink(ptu_path, ti->t, &sink);
	if(rc != HU_OK) return rc;
	ti->children.assign((size_t) ti->t.n, std::vector<int32_t>());
	for(int32_t c : order) { const int32_t p = c >= 0 && c < ti->t.n ? ti->t.parent[c] : -1; if(p >= 0 && p < ti->t.n) ti->children[p].push_back(c); }
	*out = ti.release();
	return HU_OK;
} catch(...) { return hu_catch_all("hu_tree_info_load"); }
extern "C" void hu_tree_info_free(hu_tree_info* t) { delete t; }
extern "C" int hu_tree_info_get(const hu_tree_info* t, int32_t* n_nodes, int32_t* cs_len, int32_t* root, hu_model_desc* model) {
	if(!t) return HU_ERR_ARG;
	if(n_nodes) *n_nodes = t->t.n; if(cs_len) *cs_len = t->t.csLen; if(root) *root = t->t.root; if(model) *model = t->t.model;
	return HU_OK;
}
extern "C" int hu_tree_info_node(const hu_tree_info* t, int32_t i, int32_t* parent, double* blen, double* anno_dist, int32_t* is_leaf,
		const char** name, const char** anno) {
	if(!t || i < 0 || i >= t->t.n) return HU_ERR_ARG;
	if(parent) *parent = t->t.parent[i]; if(blen) *blen = t->t.blen[i]; if(anno_dist) *anno_dist = t->t.annoDist[i];
	if(is_leaf) *is_leaf = t->children[i].empty();
	if(name) *name = t->t.names[i].c_str(); if(anno) *anno = t->t.annos[i].c_str();
	return HU_OK;
}
extern "C" int hu_tree_info_children(const hu_tree_info* t, int32_t i, const int32_t** children) {
	if(!t || i < 0 || i >= t->t.n) return 0;
	if(children) *children = t->children[i].data();
	return (int) t->children[i].size();
}
