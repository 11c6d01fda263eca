// C ABI implementation: database packing into HBM, batch workspace, stage launches and the
// host-side pieces of the per-read task (filterPlacements, calcQValues, final sort), which call
// literally the same std::sort as the reference so that tie permutations agree.
// gfx950 only; there is NO CPU fallback: every compute entry point fails with HU_ERR_DEVICE
// when no device is present.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <atomic>
#include <thread>
#include <memory>
#include <mutex>
#include <chrono>
#include <fstream>
#include <sstream>
#include <new>
#include <stdexcept>
#include "hu_common.h"
#include "hu_kern_sep.h"
#include "hu_kern_align.h"
#include "hu_kern_tree.h"
#include "hu_kern_blk.h"
#include "hu_kern_refsort.h"
#include "hu_kern_rank.h"

#define HIPCHK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) { \
	hu_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return HU_ERR_DEVICE; } } while(0)

static const double kInf = std::numeric_limits<double>::infinity();

/* the <= 50-record host stages (filterPlacements, calcQValues) are independent per read: a small persistent pool
 * per calling thread (creating and joining 16 threads per stage cost more than the stage's work) */
#include <condition_variable>
#include <functional>
struct HuPool {
	std::vector<std::thread> th;
	std::mutex m;
	std::condition_variable cvGo, cvDone;
	std::function<void(size_t)> fn;
	size_t n = 0, chunk = 64;
	std::atomic<size_t> next{0};
	uint64_t gen = 0;
	int active = 0, permit = 0;  /* permit: helpers of the current run that may work (hu_helpers_acquire); the others only report back */
	bool quit = false;
	std::exception_ptr err;      /* first exception of a work item of the current run: rethrown on the calling thread (hu_catch_all turns it into a status) */
	explicit HuPool(unsigned nt) {
		try { th.reserve(nt); } catch(...) { return; }
		for(unsigned t = 0; t < nt; ++t) try { th.emplace_back([this] {
			uint64_t seen = 0;
			for(;;) {
				{
					std::unique_lock<std::mutex> lk(m);
					cvGo.wait(lk, [&] { return quit || gen != seen; });
					if(quit) return;
					seen = gen;
					if(permit > 0) --permit; else { if(--active == 0) cvDone.notify_all(); continue; }
				}
				work();
				std::lock_guard<std::mutex> lk(m);
				if(--active == 0) cvDone.notify_all();
			}
		}); } catch(...) { break; }       /* a helper that cannot be started (std::system_error) is done without */
	}
	void work() noexcept {
		for(;;) {
			size_t a = next.fetch_add(chunk); if(a >= n) break; size_t e = std::min(n, a + chunk);
			try { for(size_t i = a; i < e; ++i) fn(i); }
			catch(...) { std::lock_guard<std::mutex> lk(m); if(!err) err = std::current_exception(); next = n; }   /* nothing leaves a thread body; the rest of the run is dropped */
		}
	}
	template<class F> void run(size_t count, F f) {
		const int got = hu_helpers_acquire((int) th.size());
		{
			std::lock_guard<std::mutex> lk(m);
			fn = f; n = count; next = 0; active = (int) th.size(); permit = got; err = nullptr; ++gen;
		}
		cvGo.notify_all();
		work();                                   /* the caller works too */
		std::unique_lock<std::mutex> lk(m);
		cvDone.wait(lk, [&] { return active == 0; });
		hu_helpers_release(got);
		if(err) { std::exception_ptr e = err; err = nullptr; lk.unlock(); std::rethrow_exception(e); }
	}
	~HuPool() { { std::lock_guard<std::mutex> lk(m); quit = true; } cvGo.notify_all(); for(auto& t : th) t.join(); }
};
template<class F> static void parallel_for(size_t n, F f) {
	/* HU_HOST_THREADS (default 8) per driving thread: several batches are in flight per GPU and eight GPUs share a host */
	static const unsigned cap = [] { const char* e = getenv("HU_HOST_THREADS"); int v = e ? atoi(e) : 8; return (unsigned)(v < 1 ? 1 : v > 64 ? 64 : v); }();
	unsigned nt = std::thread::hardware_concurrency();
	if(nt > cap) nt = cap;
	if(n < 512 || nt <= 1) { for(size_t i = 0; i < n; ++i) f(i); return; }
	static thread_local std::unique_ptr<HuPool> pool;    /* one pool per driving thread (one per batch in flight) */
	if(!pool) pool.reset(new HuPool(nt - 1));
	pool->run(n, f);
}

/* page-locked host memory for the buffers that cross PCIe every batch (pageable memory is staged by the runtime:
 * a third of the rate, and the copy blocks the calling thread) */
template<class T> struct PinnedAlloc {
	typedef T value_type;
	PinnedAlloc() = default;
	template<class U> PinnedAlloc(const PinnedAlloc<U>&) {}
	T* allocate(size_t n) {
		void* p = nullptr;
		if(hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); p = malloc(n * sizeof(T)); std::lock_guard<std::mutex> lk(mtx()); pageable().push_back(p); }
		if(!p) throw std::bad_alloc();
		return (T*) p;
	}
	void deallocate(T* p, size_t) {
		{
			std::lock_guard<std::mutex> lk(mtx());
			auto& v = pageable();
			for(size_t i = 0; i < v.size(); ++i) if(v[i] == (void*) p) { v.erase(v.begin() + i); free(p); return; }
		}
		(void) hipHostFree(p);
	}
	static std::vector<void*>& pageable() { static std::vector<void*> v; return v; }
	static std::mutex& mtx() { static std::mutex m; return m; }
	template<class U> bool operator==(const PinnedAlloc<U>&) const { return true; }
	template<class U> bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template<class T> using PinnedVec = std::vector<T, PinnedAlloc<T>>;

/* growable device buffer */
template<class X> struct DBuf {
	X* p = nullptr; size_t cap = 0;
	int ensure(size_t n) {
		if(n <= cap) return HU_OK;
		if(p) (void) hipFree(p);
		p = nullptr; cap = 0;
		size_t want = n + n / 8 + 16;
		hipError_t e = hipMalloc((void**) &p, want * sizeof(X));
		if(e != hipSuccess) { hu_set_error("hipMalloc(%zu bytes) failed: %s", want * sizeof(X), hipGetErrorString(e)); return HU_ERR_NOMEM; }
		cap = want;
		return HU_OK;
	}
	void free_() { if(p) (void) hipFree(p); p = nullptr; cap = 0; }
	DBuf() = default;
	DBuf(const DBuf&) = delete;
	DBuf& operator=(const DBuf&) = delete;
	~DBuf() { free_(); }      /* temporaries on an early-return path and the members of a deleted hu_batch free themselves */
};

struct hu_db {
	int device = 0;
	HuDbDev dev;
	double partialFrac = 0;      /* nodes whose bases cover less than 90 % of the profile block's positions (partial reference sequences) */
	HuModelDev mdl;
	hu_model_desc mdesc;
	HuProfileHost prof;
	std::vector<double> hT7, hEM, hEI;
	std::vector<int32_t> parent, annoId;
	std::vector<double> blen, height, annoDist;
	std::vector<int8_t> seq;
	std::vector<std::string> annos, names;   /* only when loaded from a .ptu */
	std::vector<void*> allocs;
	int64_t hbmBytes = 0;
	int32_t* dAnnoId = nullptr;    /* [nNodes] class of the node's taxon annotation (calcQValues sums by taxon name) */
	double* dLlTab = nullptr;      /* [csLen + 1] the final loglik placeSeq returns for a region of k columns: k sequential additions of log(sum_i pi_i e) */
};

template<class X> static int dev_alloc(hu_db* db, X** p, size_t n) {
	size_t bytes = std::max<size_t>(n, 1) * sizeof(X);
	hipError_t e = hipMalloc((void**) p, bytes);
	if(e != hipSuccess) { hu_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); return HU_ERR_NOMEM; }
	db->allocs.push_back(*p);
	db->hbmBytes += (int64_t) bytes;
	return HU_OK;
}
template<class X> static int dev_upload(hu_db* db, X** p, const X* src, size_t n) {
	int rc = dev_alloc(db, p, n);
	if(rc != HU_OK) return rc;
	if(n) HIPCHK(hipMemcpy(*p, src, n * sizeof(X), hipMemcpyHostToDevice));
	return HU_OK;
}

extern "C" int hu_device_count(void) try {
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess) return 0;
	int ok = 0;
	for(int i = 0; i < n; ++i) {
		hipDeviceProp_t p;
		if(hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
	}
	return ok;
} catch(...) { return hu_catch_all("hu_device_count"); }

static int init_sym_map() {
	int8_t m[128];
	for(int i = 0; i < 128; ++i) m[i] = -1;
	/* IUPACNucl: symbols ACGT, degenerate codes -> first expansion, gaps "-._" (src/IUPACNucl.cpp:33-50,
	 * src/DegenAlphabet.cpp:43-64) */
	m['A'] = 0; m['C'] = 1; m['G'] = 2; m['T'] = 3; m['U'] = 3;
	m['M'] = 0; m['R'] = 0; m['W'] = 0; m['S'] = 1; m['Y'] = 1; m['K'] = 2;
	m['V'] = 0; m['H'] = 0; m['D'] = 0; m['B'] = 1; m['N'] = 0;
	m['-'] = -2; m['.'] = -2; m['_'] = -2;
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_sym_map), m, sizeof(m)));
	return HU_OK;
}
static int8_t host_sym(char c) {
	switch(c) {
	case 'A': case 'M': case 'R': case 'W': case 'V': case 'H': case 'D': case 'N': return 0;
	case 'C': case 'S': case 'Y': case 'B': return 1;
	case 'G': case 'K': return 2;
	case 'T': case 'U': return 3;
	case '-': case '.': case '_': return -2;
	default: return -1;
	}
}

/* How a host thread waits for its stream.  hipStreamSynchronize under HIP's default scheduling (hipDeviceScheduleAuto) SPINS: with six batches in flight
 * per GPU that is six host cores at 100 % doing nothing (BENCH_r03: host_cores_busy_per_rank 5.8 of a 16-core quota — eight ranks would ask for ~46).
 * hipDeviceScheduleBlockingSync is not an option for a library: set after the runtime has created its queues (PyTorch initialises the device first) it
 * leaves this ROCm build with completion signals it cannot attach its interrupt handler to ("hsa_amd_signal_async_handler() failed to set the handler",
 * the next hipHostFree never returns: gpurun_out/cli_probe/trace.err, round 4).  So the engine waits by itself: it polls hipStreamQuery — a read of the
 * stream's completion signal, no system call — for the first ~30 us (a short kernel ends inside them), then sleeps between polls, 20 us growing to 100:
 * a thread waiting out a 3 ms kernel makes ~35 polls and uses ~2 % of a core; a synchronisation returns at most one sleep (<= 100 us + the wake-up) after
 * the stream has drained, which the other batches in flight hide.  HU_SYNC=spin keeps the runtime's wait. */
static hipError_t hu_wait(hipStream_t st) {
	static const bool spin = [] { const char* e = getenv("HU_SYNC"); return e && !strcmp(e, "spin"); }();
	if(spin) return hipStreamSynchronize(st);
	const auto t0 = std::chrono::steady_clock::now();
	long ns = 20000;
	for(;;) {
		const hipError_t e = hipStreamQuery(st);
		if(e != hipErrorNotReady) return e;
		if(std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(30)) continue;
		timespec ts{0, ns};
		nanosleep(&ts, nullptr);
		if(ns < 100000) ns += 20000;
	}
}

/* ------------------------------------------------------------------------------ database */
extern "C" int hu_db_create(const hu_profile_desc* prof, const hu_tree_desc* tree, const hu_model_desc* model,
		int device, hu_db** out) try {
	if(!prof || !tree || !model || !out) { hu_set_error("hu_db_create: null argument"); return HU_ERR_ARG; }
	*out = nullptr;
	if(hu_device_count() <= 0) { hu_set_error("no gfx950 device visible: the engine has no CPU path"); return HU_ERR_DEVICE; }
	HIPCHK(hipSetDevice(device));
	hu_db* db = new hu_db;
	bool built = false;
	HuScope guard([&] { if(!built) hu_db_destroy(db); });      /* a half-built database frees itself on every way out, an exception included */
	db->device = device;
	int rc = db->prof.init(prof);
	if(rc == HU_OK) rc = hu_model_prepare(model, &db->mdl);
	if(rc != HU_OK) return rc;
	db->mdesc = *model;
	const int n = tree->n_nodes, L = tree->cs_len;
	if(n < 2 || n >= (1 << 24) || L != prof->L) { hu_set_error("tree: n_nodes %d / cs_len %d inconsistent with profile L %d", n, L, prof->L); return HU_ERR_ARG; }
	const int64_t winStart = tree->win_len > 0 ? tree->win_start : 0, winLen = tree->win_len > 0 ? tree->win_len : L;
	if(winStart < 0 || winStart + winLen > L) { hu_set_error("tree: message window out of range"); return HU_ERR_ARG; }
	db->parent.assign(tree->parent, tree->parent + n);
	db->blen.assign(tree->blen, tree->blen + n);
	db->height.assign(tree->height, tree->height + n);
	db->seq.assign(tree->seq, tree->seq + (size_t) n * L);
	if(tree->anno_id) db->annoId.assign(tree->anno_id, tree->anno_id + n);
	else { db->annoId.resize(n); for(int i = 0; i < n; ++i) db->annoId[i] = i; }
	if(tree->anno_dist) db->annoDist.assign(tree->anno_dist, tree->anno_dist + n); else db->annoDist.assign(n, 0.0);
	int root = -1, nroot = 0;
	for(int i = 0; i < n; ++i) {
		if(db->parent[i] < 0) { root = i; nroot++; }
		else if(db->parent[i] >= n) { hu_set_error("tree: parent of node %d out of range", i); return HU_ERR_ARG; }
	}
	if(nroot != 1) { hu_set_error("tree: %d roots", nroot); return HU_ERR_ARG; }
	HuDbDev& d = db->dev;
	memset(&d, 0, sizeof(d));
	d.nNodes = n; d.csLen = L; d.root = root;
	d.nNodesPad = (n + HU_NODE_PAD - 1) / HU_NODE_PAD * HU_NODE_PAD;
	d.W = (L + 31) / 32; d.WQ = (L + 127) / 128;
	d.winStart = winStart; d.winLen = winLen;
	d.K = prof->K; d.L = L;
	auto fail = [&](int code) { return code; };
	/* scan order of the CS columns: the profile (match) columns first, then the rest, both ascending */
	std::vector<int32_t> posCol((size_t) d.WQ * 128, -1), colPos(L, -1);
	{
		int p = 0;
		for(int k = 1; k <= prof->K; ++k) { const int c = db->prof.p2cs[k] - 1; colPos[c] = p; posCol[p++] = c; }
		for(int c = 0; c < L; ++c) if(colPos[c] < 0) { colPos[c] = p; posCol[p++] = c; }
		int32_t* dp = nullptr;
		if((rc = dev_upload(db, &dp, posCol.data(), posCol.size())) != HU_OK) return fail(rc);
		d.posCol = dp;
	}
	/* bit-planes of the node sequences in scan order, [WQ][3][nNodesPad] x uint4 */
	{
		const size_t np = d.nNodesPad, cnt = (size_t) d.WQ * 3 * np;
		std::vector<uint4> pl(cnt, make_uint4(0, 0, 0, 0));
		for(int i = 0; i < n; ++i) {
			const int8_t* s = &db->seq[(size_t) i * L];
			for(int cc = 0; cc < L; ++cc) {
				const int code = s[cc];
				if(code < 0) continue;
				const int c = colPos[cc];
				const int q = c >> 7, w = (c >> 5) & 3; const uint32_t bit = 1u << (c & 31);
				uint32_t* p0 = &pl[((size_t) q * 3 + 0) * np + i].x + w;
				uint32_t* p1 = &pl[((size_t) q * 3 + 1) * np + i].x + w;
				uint32_t* pv = &pl[((size_t) q * 3 + 2) * np + i].x + w;
				if(code & 1) *p0 |= bit;
				if(code & 2) *p1 |= bit;
				*pv |= bit;
			}
		}
		uint4* dp = nullptr;
		if((rc = dev_upload(db, &dp, pl.data(), cnt)) != HU_OK) return fail(rc);
		d.planes = dp;
		d.QM = (prof->K + 127) / 128;      /* scan positions >= QM * 128 hold non-profile columns only */
		if(d.WQ > d.QM) {
			unsigned long long* cp = nullptr;
			if((rc = dev_alloc(db, &cp, (size_t)(d.WQ - d.QM) * 128 * 3 * (np / 64))) != HU_OK) return fail(rc);
			k_col_planes<<<dim3((unsigned)(np / 64), (unsigned)(d.WQ - d.QM)), 64>>>(d, cp);
			hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
			if(e1 != hipSuccess || e2 != hipSuccess) { hu_set_error("transposing the non-profile planes failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); return fail(HU_ERR_DEVICE); }
			d.colPlanes = cp;
		}
		{
			uint2* cv = nullptr;
			if((rc = dev_alloc(db, &cv, np)) != HU_OK) return fail(rc);
			k_node_cover<<<(unsigned)(np / 256), 256>>>(d, cv);
			hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
			if(e1 != hipSuccess || e2 != hipSuccess) { hu_set_error("node coverage intervals failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); return fail(HU_ERR_DEVICE); }
			d.nodeCover = cv;
			/* how many sequences are partial: with many of them the distance of a read to a barely overlapping node (d / N over a handful
			 * of columns) ranks among the best, d_scan orders the candidates badly, and the seed stage takes the pair-matrix path */
			std::vector<uint2> hc((size_t) d.nNodes);
			if(hipMemcpy(hc.data(), cv, hc.size() * sizeof(uint2), hipMemcpyDeviceToHost) != hipSuccess) { hu_set_error("node coverage intervals: copy failed"); return fail(HU_ERR_DEVICE); }
			const uint32_t span = (uint32_t) std::min<int64_t>((int64_t) d.QM * 128, (int64_t) prof->K);
			size_t partial = 0;
			for(const uint2& c : hc) { const uint32_t f = c.x & 0xffffu, l = c.x >> 16; if(f > l || (l - f + 1) * 10u < span * 9u) ++partial; }
			db->partialFrac = d.nNodes ? (double) partial / d.nNodes : 0.0;
		}
	}
	{
		int32_t* p; double* q;
		if((rc = dev_upload(db, &p, db->parent.data(), (size_t) n)) != HU_OK) return fail(rc); d.parent = p;
		if((rc = dev_upload(db, &q, db->blen.data(), (size_t) n)) != HU_OK) return fail(rc); d.blen = q;
		if((rc = dev_upload(db, &q, db->height.data(), (size_t) n)) != HU_OK) return fail(rc); d.height = q;
	}
	{
		double *qu, *qd;
		const size_t cnt = (size_t) n * winLen * 4;
		if(tree->msgs_on_device) { qu = const_cast<double*>(tree->up); qd = const_cast<double*>(tree->down); db->hbmBytes += 2 * (int64_t) cnt * 8; }
		else {
			if((rc = dev_upload(db, &qu, tree->up, cnt)) != HU_OK) return fail(rc);
			if((rc = dev_upload(db, &qd, tree->down, cnt)) != HU_OK) return fail(rc);
		}
		/* pack once: log-space .ptu messages -> linear 4-vectors + binary exponent, in place */
		int32_t *ku, *kd;
		const size_t ns = (size_t) n * winLen;
		if((rc = dev_alloc(db, &ku, ns)) != HU_OK) return fail(rc);
		if((rc = dev_alloc(db, &kd, ns)) != HU_OK) return fail(rc);
		(void) hipGetLastError();
		k_pack_msgs<<<(unsigned)((ns + 255) / 256), 256>>>(db->mdl, qu, ku, ns);
		k_pack_msgs<<<(unsigned)((ns + 255) / 256), 256>>>(db->mdl, qd, kd, ns);
		hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
		if(e1 != hipSuccess || e2 != hipSuccess) { hu_set_error("message packing failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); return fail(HU_ERR_DEVICE); }
		d.up = qu; d.down = qd; d.upK = ku; d.downK = kd;
	}
	{ /* profile */
		const int K = d.K;
		std::vector<double> T8((size_t)(K + 1) * 8, kInf);
		for(int k = 0; k <= K; ++k) for(int t = 0; t < 7; ++t) T8[(size_t) k * 8 + t] = db->prof.T7[(size_t) k * 7 + t];
		std::vector<int32_t> p2(K + 2, 0);
		for(int k = 0; k <= K; ++k) p2[k] = db->prof.p2cs[k];
		double* q; int32_t* p;
		if((rc = dev_upload(db, &q, db->prof.EM.data(), (size_t) 4 * (K + 1))) != HU_OK) return fail(rc); d.EM = q;
		if((rc = dev_upload(db, &q, db->prof.EI.data(), (size_t) 4 * (K + 1))) != HU_OK) return fail(rc); d.EI = q;
		if((rc = dev_upload(db, &q, T8.data(), T8.size())) != HU_OK) return fail(rc); d.T = q;
		if((rc = dev_upload(db, &q, db->prof.entryC.data(), (size_t) K + 1)) != HU_OK) return fail(rc); d.entryC = q;
		if((rc = dev_upload(db, &q, db->prof.exitC.data(), (size_t) K + 1)) != HU_OK) return fail(rc); d.exitC = q;
		if((rc = dev_upload(db, &p, p2.data(), p2.size())) != HU_OK) return fail(rc); d.p2cs = p;
		{
			const size_t K1 = (size_t) K + 1;
			std::vector<double> Tt(7 * K1), EMt(4 * K1), EIt(4 * K1);
			for(int k = 0; k <= K; ++k) {
				for(int t = 0; t < 7; ++t) Tt[t * K1 + k] = db->prof.T7[(size_t) k * 7 + t];
				for(int c = 0; c < 4; ++c) { EMt[c * K1 + k] = db->prof.EM[(size_t) k * 4 + c]; EIt[c * K1 + k] = db->prof.EI[(size_t) k * 4 + c]; }
			}
			if((rc = dev_upload(db, &q, Tt.data(), Tt.size())) != HU_OK) return fail(rc); d.Tt = q;
			if((rc = dev_upload(db, &q, EMt.data(), EMt.size())) != HU_OK) return fail(rc); d.EMt = q;
			if((rc = dev_upload(db, &q, EIt.data(), EIt.size())) != HU_OK) return fail(rc); d.EIt = q;
		}
		double pc[HU_PC_COUNT];
		hu_place_consts(db->mdl, pc);
		if((rc = dev_upload(db, &q, pc, (size_t) HU_PC_COUNT)) != HU_OK) return fail(rc); d.placeConst = q;
	}
	{ /* what the finish stage reads on the device */
		if((rc = dev_upload(db, &db->dAnnoId, db->annoId.data(), (size_t) n)) != HU_OK) return fail(rc);
		const double e1 = std::exp(1.0);
		const double siteLL = std::log((db->mdl.pi[0] * e1 + db->mdl.pi[2] * e1) + (db->mdl.pi[1] * e1 + db->mdl.pi[3] * e1));   /* dot_product_scaled(pi, ones): SSE2 association (e0 + e2) + (e1 + e3) */
		std::vector<double> tab((size_t) L + 1, 0.0);
		for(int k = 1; k <= L; ++k) tab[k] = tab[k - 1] + siteLL;
		if((rc = dev_upload(db, &db->dLlTab, tab.data(), tab.size())) != HU_OK) return fail(rc);
	}
	if((rc = init_sym_map()) != HU_OK) return fail(rc);
	built = true;
	*out = db;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_db_create"); }

extern "C" int hu_db_load(const char* hmm_path, const char* ptu_path, int device, hu_db** out) { return hu_db_load_window(hmm_path, ptu_path, device, 0, 0, out); }

/* the same for a COLUMN WINDOW of the messages (win_len > 0): the device keeps the 4 x win_len doubles of every directed edge that lie in
 * [win_start, win_start + win_len) — the unit of column-window sharding (hu_windows_plan): a database beyond one GPU's HBM is held as several
 * such windows, one per device.  Profile, node sequences and tree are whole in every window. */
extern "C" int hu_db_load_window(const char* hmm_path, const char* ptu_path, int device, int64_t win_start, int64_t win_len, hu_db** out) try {
	if(!hmm_path || !ptu_path || !out) { hu_set_error("hu_db_load: null argument"); return HU_ERR_ARG; }
	if(win_len < 0 || win_start < 0) { hu_set_error("hu_db_load_window: negative window"); return HU_ERR_ARG; }
	*out = nullptr;
	if(hu_device_count() <= 0) { hu_set_error("no gfx950 device visible: the engine has no CPU path"); return HU_ERR_DEVICE; }
	HuProfileHost prof; std::vector<double> EM, EI, T; std::vector<int32_t> p2cs; int K, L;
	int rc = hu_read_hmm(hmm_path, prof, EM, EI, T, p2cs, K, L);
	if(rc != HU_OK) return rc;
	HIPCHK(hipSetDevice(device));
	/* the messages (4 x csLen doubles per directed edge: 98 GB at gg_97 scale) go from the file to the device edge by edge,
	 * through two page-locked staging rows, never through a host copy of the whole set */
	HuTreeHost t;
	double *dUp = nullptr, *dDown = nullptr;
	double* stage[2] = {nullptr, nullptr};
	hipStream_t st = nullptr;
	hipEvent_t ev[2] = {nullptr, nullptr};
	bool keepMsgs = false;
	HuScope guard([&] { /* on every way out, an exception of the reader included */
		for(int i = 0; i < 2; ++i) { if(stage[i]) (void) hipHostFree(stage[i]); if(ev[i]) (void) hipEventDestroy(ev[i]); }
		if(st) (void) hipStreamDestroy(st);
		if(!keepMsgs) { if(dUp) (void) hipFree(dUp); if(dDown) (void) hipFree(dDown); }
	});
	size_t row = 0; int turn = 0; hipError_t herr = hipSuccess;
	const std::function<int(bool, int64_t, const double*)> sink = [&](bool isDown, int64_t node, const double* data) -> int {
		if(!dUp) { /* first message: n and csLen are known */
			if(win_len > 0 && win_start + win_len > t.csLen) { hu_set_error("hu_db_load_window: columns %lld .. %lld of a database of %d", (long long) win_start, (long long)(win_start + win_len - 1), t.csLen); return HU_ERR_ARG; }
			row = (size_t)(win_len > 0 ? win_len : t.csLen) * 4;
			const size_t bytes = (size_t) t.n * row * sizeof(double);
			if((herr = hipMalloc((void**) &dUp, bytes)) != hipSuccess || (herr = hipMalloc((void**) &dDown, bytes)) != hipSuccess ||
					(herr = hipMemset(dDown, 0, bytes)) != hipSuccess || (herr = hipStreamCreate(&st)) != hipSuccess) return HU_ERR_NOMEM;
			for(int i = 0; i < 2; ++i) if((herr = hipHostMalloc((void**) &stage[i], row * sizeof(double), hipHostMallocDefault)) != hipSuccess ||
					(herr = hipEventCreate(&ev[i])) != hipSuccess) return HU_ERR_NOMEM;
		}
		if((herr = hipEventSynchronize(ev[turn])) != hipSuccess) return HU_ERR_DEVICE;     /* the copy that last used this staging row */
		memcpy(stage[turn], data + (win_len > 0 ? (size_t) win_start * 4 : 0), row * sizeof(double));
		if((herr = hipMemcpyAsync((isDown ? dDown : dUp) + (size_t) node * row, stage[turn], row * sizeof(double), hipMemcpyHostToDevice, st)) != hipSuccess ||
				(herr = hipEventRecord(ev[turn], st)) != hipSuccess) return HU_ERR_DEVICE;
		turn ^= 1;
		return HU_OK;
	};
	rc = hu_read_ptu_sink(ptu_path, t, &sink);
	if(rc == HU_OK && st && (herr = hipStreamSynchronize(st)) != hipSuccess) rc = HU_ERR_DEVICE;
	if(rc != HU_OK) {
		if(herr != hipSuccess) hu_set_error("hu_db_load: moving the messages to the device failed: %s", hipGetErrorString(herr));
		return rc;
	}
	if(K > t.csLen) { hu_set_error("HMM profile size is greater than the tree's CS length"); return HU_ERR_ARG; }
	hu_profile_desc pd{K, t.csLen, EM.data(), EI.data(), T.data(), p2cs.data()};
	hu_tree_desc td;
	memset(&td, 0, sizeof(td));
	td.n_nodes = t.n; td.cs_len = t.csLen; td.parent = t.parent.data(); td.blen = t.blen.data(); td.seq = t.seq.data();
	td.up = dUp; td.down = dDown; td.msgs_on_device = 1;
	if(win_len > 0) { td.win_start = win_start; td.win_len = win_len; }
	td.height = t.height.data(); td.anno_id = t.annoId.data(); td.anno_dist = t.annoDist.data();
	rc = hu_db_create(&pd, &td, &t.model, device, out);
	keepMsgs = rc == HU_OK;
	if(rc == HU_OK) { (*out)->annos = t.annos; (*out)->names = t.names; (*out)->allocs.push_back(dUp); (*out)->allocs.push_back(dDown); }   /* the database owns them */
	return rc;
} catch(...) { return hu_catch_all("hu_db_load_window"); }

/* PTUnrooted::save (src/PhyloTreeUnrooted.cpp:537-567 and :116-129, :595-603, :632-670, :672-697; src/DigitalSeq.cpp:96-104;
 * src/util/ProgEnv.cpp:24-28): the database file hmmufotu / hu_db_load read.  The messages may live on the device (98 GB at
 * gg_97 scale, straight out of hu_tree_evaluate): they go to the file edge by edge through a page-locked row. */
static std::string model_text_of(const hu_model_desc& m) {
	static const char* names[] = {"GTR", "TN93", "HKY85", "F81", "K80", "JC69"};
	char t[64];
	auto num = [&](double v) { snprintf(t, sizeof(t), "%.17g", v); return std::string(t); };
	std::string o = std::string("# DNA Substitution Model\nType: ") + names[m.type] + "\n";
	if(m.type != HU_K80 && m.type != HU_JC69) o += "pi: " + num(m.pi[0]) + " " + num(m.pi[1]) + " " + num(m.pi[2]) + " " + num(m.pi[3]) + "\n";
	if(m.type == HU_GTR) {
		o += "R:\n";
		for(int i = 0; i < 4; ++i) o += num(m.par[4 * i]) + " " + num(m.par[4 * i + 1]) + " " + num(m.par[4 * i + 2]) + " " + num(m.par[4 * i + 3]) + "\n";
		/* Q = R diag(pi), rows summing to 0, scaled to unit rate (GTR::setQfromParams, src/GTR.cpp:124-131); the reference's reader skips
		 * these lines ("for human read only", :68-72) */
		double Q[16], mu = 0;
		for(int i = 0; i < 4; ++i) { double rs = 0; for(int j = 0; j < 4; ++j) { Q[4 * i + j] = i == j ? 0 : m.par[4 * i + j] * m.pi[j]; rs += Q[4 * i + j]; } Q[5 * i] = -rs; mu += m.pi[i] * rs; }
		o += "Q:\n";
		for(int i = 0; i < 4; ++i) o += num(Q[4 * i] / mu) + " " + num(Q[4 * i + 1] / mu) + " " + num(Q[4 * i + 2] / mu) + " " + num(Q[4 * i + 3] / mu) + "\n";
	}
	else if(m.type == HU_TN93) o += "kr: " + num(m.par[0]) + " ky: " + num(m.par[1]) + " beta: " + num(m.par[2]) + "\n";
	else if(m.type == HU_HKY85) o += "kappa: " + num(m.par[0]) + " beta: " + num(m.par[1]) + "\n";
	else if(m.type == HU_F81) o += "beta: " + num(m.par[0]) + "\n";
	else if(m.type == HU_K80) o += "kappa: " + num(m.par[0]) + "\n";
	return o;
}
extern "C" int hu_ptu_write(const char* path, const hu_tree_desc* t, const char* const* names, const char* const* annos, const hu_model_desc* model,
		const char* model_text, double dg_alpha, const double* dg_breaks) try {
	if(!path || !t || !model || t->n_nodes < 2 || t->cs_len < 1 || !t->parent || !t->blen || !t->seq || !t->up || !t->down || !t->height) { hu_set_error("hu_ptu_write: bad argument"); return HU_ERR_ARG; }
	if(model->type < 0 || model->type > HU_JC69 || model->dg_k < 0 || model->dg_k > HU_MAX_DGK) { hu_set_error("hu_ptu_write: bad model"); return HU_ERR_ARG; }
	if(t->win_len > 0 && t->win_len != t->cs_len) { hu_set_error("hu_ptu_write: the file format holds whole messages, not a column window"); return HU_ERR_ARG; }
	const int n = t->n_nodes, L = t->cs_len;
	std::vector<std::vector<int32_t>> children(n);
	int root = -1;
	for(int i = 0; i < n; ++i) { if(t->parent[i] < 0) root = i; else if(t->parent[i] < n) children[t->parent[i]].push_back(i); else { hu_set_error("hu_ptu_write: parent out of range"); return HU_ERR_ARG; } }
	if(root < 0) { hu_set_error("hu_ptu_write: tree has no root"); return HU_ERR_ARG; }
	std::ofstream f(path, std::ios::binary);
	if(!f) { hu_set_error("cannot write PTU file '%s'", path); return HU_ERR_IO; }
	auto put = [&](const void* p, size_t k) { f.write((const char*) p, (std::streamsize) k); };
	auto str = [&](const char* s_, size_t k) { const uint64_t len = k; put(&len, 8); if(k) put(s_, k); };
	put("HmmUFOtu", 8); { const int32_t ver[3] = {1, 5, 1}; put(ver, 12); }
	{ const uint64_t nn = (uint64_t) n; put(&nn, 8); const int32_t l = L; put(&l, 4); }
	char tmp[32];
	for(int i = 0; i < n; ++i) {
		const int64_t id = i; put(&id, 8);
		const char* nm = names && names[i] ? names[i] : (snprintf(tmp, sizeof(tmp), "n%d", i), tmp);
		str(nm, strlen(nm));
		const uint8_t withAbc = 0; put(&withAbc, 1);
		str(nm, strlen(nm));
		str((const char*)(t->seq + (size_t) i * L), (size_t) L);
		const char* an = annos && annos[i] ? annos[i] : "";
		str(an, strlen(an));
		const double ad = t->anno_dist ? t->anno_dist[i] : 0.0; put(&ad, 8);
	}
	const uint64_t nEdges = 2ull * (n - 1); put(&nEdges, 8);
	const size_t row = (size_t) L * 4;
	double* stage = nullptr;
	std::vector<double> hostRow;
	if(t->msgs_on_device) { if(hipHostMalloc((void**) &stage, row * 8, hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); hostRow.resize(row); stage = hostRow.data(); } }
	auto msg = [&](const double* base, int node) -> const double* {
		const double* p = base + (size_t) node * row;
		if(!t->msgs_on_device) return p;
		if(hipMemcpy(stage, p, row * 8, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
		return stage;
	};
	bool ok = true;
	for(int u = 0; u < n && ok; ++u) { /* both directions of every edge, grouped by their first node: parent first, then the children */
		std::vector<int32_t> nb;
		if(t->parent[u] >= 0) nb.push_back(t->parent[u]);
		nb.insert(nb.end(), children[u].begin(), children[u].end());
		for(int32_t v : nb) {
			const bool uIsParent = t->parent[u] < 0 || v != t->parent[u];
			const int child = uIsParent ? v : u;
			const int64_t a = u, b2 = v; put(&a, 8); put(&b2, 8);
			const uint8_t fl = uIsParent ? 1 : 0; put(&fl, 1);
			const double len = t->blen[child]; put(&len, 8);
			const uint64_t N = row; put(&N, 8);
			const double* m = msg(uIsParent ? t->down : t->up, child);
			if(!m) { ok = false; break; }
			put(m, row * 8);
		}
	}
	if(ok) {
		const int64_t rid = root; put(&rid, 8);
		const double* m = msg(t->up, root);
		if(m) put(m, row * 8); else ok = false;
	}
	if(stage && hostRow.empty()) (void) hipHostFree(stage);
	if(!ok) { hu_set_error("hu_ptu_write: copying a message from the device failed"); return HU_ERR_DEVICE; }
	for(int i = 0; i < n; ++i) { const int64_t id = i; put(&id, 8); put(&t->height[i], 8); }
	{ /* MSA index: the leaves in node order */
		uint32_t nl = 0;
		for(int i = 0; i < n; ++i) nl += children[i].empty();
		put(&nl, 4);
		uint32_t k = 0;
		for(int i = 0; i < n; ++i) if(children[i].empty()) { put(&k, 4); const int64_t id = i; put(&id, 8); ++k; }
	}
	{
		static const char* mnames[] = {"GTR", "TN93", "HKY85", "F81", "K80", "JC69"};
		std::string mt = std::string(mnames[model->type]) + "\n" + (model_text ? std::string(model_text) : model_text_of(*model));
		if(mt.empty() || mt.back() != '\n') mt += '\n';
		put(mt.data(), mt.size());
	}
	const uint8_t hasDG = model->dg_k > 0; put(&hasDG, 1);
	if(hasDG) {
		const int32_t K = model->dg_k; put(&K, 4); put(&dg_alpha, 8);
		for(int i = 0; i <= K; ++i) { const double b0 = dg_breaks ? dg_breaks[i] : 0.0; put(&b0, 8); }
		put(model->dg_rate, (size_t) K * 8);
	}
	f.flush();
	if(!f) { hu_set_error("writing PTU file '%s' failed", path); return HU_ERR_IO; }
	return HU_OK;
} catch(...) { return hu_catch_all("hu_ptu_write"); }

/* the reference's own text forms, parsed from memory: what operator<<(ostream&, const BandedHMMP7&) and DNASubModel::write emit */
extern "C" int hu_profile_parse_text(const char* text, int64_t len, int32_t* K, int32_t* L, double* EM, double* EI, double* T, int32_t* p2cs) try {
	if(!text || len < 0 || !K || !L) { hu_set_error("hu_profile_parse_text: bad argument"); return HU_ERR_ARG; }
	std::istringstream in(std::string(text, (size_t) len));
	HuProfileHost prof; std::vector<double> vEM, vEI, vT; std::vector<int32_t> vp; int k = 0, l = 0;
	int rc = hu_read_hmm_stream(in, "<memory>", prof, vEM, vEI, vT, vp, k, l);
	if(rc != HU_OK) return rc;
	*K = k; *L = l;
	if(EM) memcpy(EM, vEM.data(), vEM.size() * 8); if(EI) memcpy(EI, vEI.data(), vEI.size() * 8);
	if(T) memcpy(T, vT.data(), vT.size() * 8); if(p2cs) memcpy(p2cs, vp.data(), vp.size() * 4);
	return HU_OK;
} catch(...) { return hu_catch_all("hu_profile_parse_text"); }
extern "C" int hu_model_parse_text(const char* text, int64_t len, hu_model_desc* out) try {
	if(!text || len < 0 || !out) { hu_set_error("hu_model_parse_text: bad argument"); return HU_ERR_ARG; }
	std::istringstream in(std::string(text, (size_t) len));
	return hu_read_model_text(in, *out);
} catch(...) { return hu_catch_all("hu_model_parse_text"); }

/* host-only parse of the two files (no device needed): used by the format tests */
extern "C" int hu_files_parse(const char* hmm_path, const char* ptu_path, int32_t* K, int32_t* L, int32_t* n_nodes, int32_t* root,
		double* EM, double* EI, double* T, int32_t* p2cs, double* entryC, double* exitC,
		int32_t* parent, double* blen, int8_t* seq, double* height, double* up, double* down, hu_model_desc* model, int fill) try {
	int k = 0, l = 0;
	HuProfileHost prof; std::vector<double> vEM, vEI, vT; std::vector<int32_t> vp;
	if(hmm_path) {
		int rc = hu_read_hmm(hmm_path, prof, vEM, vEI, vT, vp, k, l);
		if(rc != HU_OK) return rc;
		if(K) *K = k; if(L) *L = l;
		if(fill) {
			if(EM) memcpy(EM, vEM.data(), vEM.size() * 8); if(EI) memcpy(EI, vEI.data(), vEI.size() * 8);
			if(T) memcpy(T, vT.data(), vT.size() * 8); if(p2cs) memcpy(p2cs, vp.data(), vp.size() * 4);
			if(entryC) memcpy(entryC, prof.entryC.data(), prof.entryC.size() * 8);
			if(exitC) memcpy(exitC, prof.exitC.data(), prof.exitC.size() * 8);
		}
	}
	if(ptu_path) {
		HuTreeHost t;
		int rc = hu_read_ptu(ptu_path, t);
		if(rc != HU_OK) return rc;
		if(n_nodes) *n_nodes = t.n; if(root) *root = t.root; if(L && !hmm_path) *L = t.csLen;
		if(model) *model = t.model;
		if(fill) {
			if(parent) memcpy(parent, t.parent.data(), t.parent.size() * 4); if(blen) memcpy(blen, t.blen.data(), t.blen.size() * 8);
			if(seq) memcpy(seq, t.seq.data(), t.seq.size()); if(height) memcpy(height, t.height.data(), t.height.size() * 8);
			if(up) memcpy(up, t.up.data(), t.up.size() * 8); if(down) memcpy(down, t.down.data(), t.down.size() * 8);
		}
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_files_parse"); }

extern "C" void hu_db_destroy(hu_db* db) try {
	if(!db) return;
	for(void* p : db->allocs) (void) hipFree(p);
	delete db;
} catch(...) { (void) hu_catch_all("hu_db_destroy"); }
extern "C" int hu_db_info(const hu_db* db, int32_t* K, int32_t* cs_len, int32_t* n_nodes, int32_t* root, int64_t* hbm_bytes) try {
	if(!db) return HU_ERR_ARG;
	if(K) *K = db->dev.K; if(cs_len) *cs_len = db->dev.csLen; if(n_nodes) *n_nodes = db->dev.nNodes;
	if(root) *root = db->dev.root; if(hbm_bytes) *hbm_bytes = db->hbmBytes;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_db_info"); }
extern "C" int hu_db_get_profile(const hu_db* db, double* EM, double* EI, double* T, int32_t* p2cs, double* entry_cost, double* exit_cost) try {
	if(!db) return HU_ERR_ARG;
	const HuProfileHost& p = db->prof;
	if(EM) memcpy(EM, p.EM.data(), p.EM.size() * 8); if(EI) memcpy(EI, p.EI.data(), p.EI.size() * 8);
	if(T) memcpy(T, p.T7.data(), p.T7.size() * 8); if(p2cs) memcpy(p2cs, p.p2cs.data(), p.p2cs.size() * 4);
	if(entry_cost) memcpy(entry_cost, p.entryC.data(), p.entryC.size() * 8);
	if(exit_cost) memcpy(exit_cost, p.exitC.data(), p.exitC.size() * 8);
	return HU_OK;
} catch(...) { return hu_catch_all("hu_db_get_profile"); }
extern "C" int hu_db_get_tree(const hu_db* db, int32_t* parent, double* blen, int8_t* seq, double* height) try {
	if(!db) return HU_ERR_ARG;
	if(parent) memcpy(parent, db->parent.data(), db->parent.size() * 4); if(blen) memcpy(blen, db->blen.data(), db->blen.size() * 8);
	if(seq) memcpy(seq, db->seq.data(), db->seq.size()); if(height) memcpy(height, db->height.data(), db->height.size() * 8);
	return HU_OK;
} catch(...) { return hu_catch_all("hu_db_get_tree"); }
extern "C" const char* hu_db_get_annotation(const hu_db* db, int32_t node) try {
	if(!db || node < 0 || node >= (int32_t) db->annos.size()) return "";
	return db->annos[node].c_str();
} catch(...) { (void) hu_catch_all("hu_db_get_annotation"); return ""; }
extern "C" int hu_db_get_model(const hu_db* db, hu_model_desc* out) try { if(!db || !out) return HU_ERR_ARG; *out = db->mdesc; return HU_OK; } catch(...) { return hu_catch_all("hu_db_get_model"); }

__global__ void k_model_pr(HuModelDev mdl, int n, const double* __restrict__ t, double* __restrict__ P) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	double E[4];
	for(int k = 0; k < 4; ++k) E[k] = exp(mdl.lam[k] * t[i]);
	for(int col = 0; col < 4; ++col) {
		double a[4], c[4];
		for(int m = 0; m < 4; ++m) a[m] = mdl.U1[m*4+col];
		if(t[i] == 0) { for(int r = 0; r < 4; ++r) c[r] = r == col ? 1.0 : 0.0; } else conv_eig(mdl, E, a, c);
		for(int r = 0; r < 4; ++r) P[(size_t) i * 16 + r * 4 + col] = c[r];
	}
}
extern "C" int hu_db_model_pr(const hu_db* db, int n, const double* t, double* P) try {
	if(!db || n < 0) return HU_ERR_ARG;
	HIPCHK(hipSetDevice(db->device));
	DBuf<double> dt, dP;
	int rc;
	if((rc = dt.ensure((size_t) std::max(n, 1))) != HU_OK || (rc = dP.ensure((size_t) std::max(n, 1) * 16)) != HU_OK) return rc;
	HIPCHK(hipMemcpy(dt.p, t, (size_t) n * 8, hipMemcpyHostToDevice));
	(void) hipGetLastError();
	if(n) k_model_pr<<<(n + 63) / 64, 64>>>(db->mdl, n, dt.p, dP.p);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpy(P, dP.p, (size_t) n * 128, hipMemcpyDeviceToHost));
	return HU_OK;
} catch(...) { return hu_catch_all("hu_db_model_pr"); }

/* ------------------------------------------------------------------------------ tree pre-evaluation */
extern "C" int hu_tree_evaluate(int32_t n, int32_t cs_len, const int32_t* parent, const double* blen, int8_t* seq,
		const hu_model_desc* model, int device, int64_t win_start, int64_t win_len, double* up_dev, double* down_dev, double* height) try {
	if(n < 2 || cs_len < 1 || !parent || !blen || !seq || !model || !up_dev || !down_dev) { hu_set_error("hu_tree_evaluate: bad argument"); return HU_ERR_ARG; }
	if(hu_device_count() <= 0) { hu_set_error("no gfx950 device visible: the engine has no CPU path"); return HU_ERR_DEVICE; }
	if(win_len <= 0) { win_start = 0; win_len = cs_len; }
	if(win_start < 0 || win_start + win_len > cs_len) { hu_set_error("hu_tree_evaluate: window out of range"); return HU_ERR_ARG; }
	HIPCHK(hipSetDevice(device));
	HuModelDev mdl;
	int rc = hu_model_prepare(model, &mdl);
	if(rc != HU_OK) return rc;
	int root = -1;
	std::vector<int32_t> cnt(n + 1, 0), depth(n, -1);
	for(int i = 0; i < n; ++i) {
		if(parent[i] < 0) { if(root >= 0) { hu_set_error("tree has more than one root"); return HU_ERR_ARG; } root = i; }
		else if(parent[i] >= n) { hu_set_error("parent of node %d out of range", i); return HU_ERR_ARG; }
		else cnt[parent[i] + 1]++;
	}
	if(root < 0) { hu_set_error("tree has no root"); return HU_ERR_ARG; }
	for(int i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
	std::vector<int32_t> childIdx(std::max(n - 1, 1)), fill(cnt.begin(), cnt.end() - 1);
	for(int i = 0; i < n; ++i) if(parent[i] >= 0) childIdx[fill[parent[i]]++] = i;   /* children in node-id order */
	/* depths by BFS from the root (no assumption on node numbering) */
	std::vector<int32_t> order; order.reserve(n); order.push_back(root); depth[root] = 0;
	for(size_t h = 0; h < order.size(); ++h) { const int u = order[h]; for(int c = cnt[u]; c < cnt[u + 1]; ++c) { depth[childIdx[c]] = depth[u] + 1; order.push_back(childIdx[c]); } }
	if((int) order.size() != n) { hu_set_error("tree is not connected"); return HU_ERR_ARG; }
	int maxD = 0;
	for(int i = 0; i < n; ++i) maxD = std::max(maxD, depth[i]);
	std::vector<int32_t> lvOff(maxD + 2, 0);
	for(int i = 0; i < n; ++i) lvOff[depth[i] + 1]++;
	for(int d = 0; d <= maxD; ++d) lvOff[d + 1] += lvOff[d];
	/* BFS order is already sorted by depth */
	HuTreeDev t;
	t.n = n; t.csLen = cs_len; t.root = root; t.winStart = win_start; t.winLen = win_len; t.up = up_dev; t.down = down_dev;
	int32_t *dPar = nullptr, *dOff = nullptr, *dIdx = nullptr, *dOrd = nullptr; double* dLen = nullptr; int8_t* dSeq = nullptr;
	HuScope guard([&] { (void) hipFree(dPar); (void) hipFree(dOff); (void) hipFree(dIdx); (void) hipFree(dOrd); (void) hipFree(dLen); (void) hipFree(dSeq); });
	#define TCHK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) { hu_set_error("%s failed: %s", #call, hipGetErrorString(e_)); return HU_ERR_DEVICE; } } while(0)
	TCHK(hipMalloc((void**) &dPar, (size_t) n * 4)); TCHK(hipMalloc((void**) &dOff, (size_t)(n + 1) * 4)); TCHK(hipMalloc((void**) &dIdx, childIdx.size() * 4));
	TCHK(hipMalloc((void**) &dOrd, (size_t) n * 4)); TCHK(hipMalloc((void**) &dLen, (size_t) n * 8)); TCHK(hipMalloc((void**) &dSeq, (size_t) n * cs_len));
	TCHK(hipMemcpy(dPar, parent, (size_t) n * 4, hipMemcpyHostToDevice)); TCHK(hipMemcpy(dOff, cnt.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
	TCHK(hipMemcpy(dIdx, childIdx.data(), childIdx.size() * 4, hipMemcpyHostToDevice)); TCHK(hipMemcpy(dOrd, order.data(), (size_t) n * 4, hipMemcpyHostToDevice));
	TCHK(hipMemcpy(dLen, blen, (size_t) n * 8, hipMemcpyHostToDevice)); TCHK(hipMemcpy(dSeq, seq, (size_t) n * cs_len, hipMemcpyHostToDevice));
	t.parent = dPar; t.blen = dLen; t.childOff = dOff; t.childIdx = dIdx; t.seq = dSeq;
	(void) hipGetLastError();
	const unsigned gx = (unsigned)((win_len + 255) / 256);
	for(int d = maxD; d >= 0; --d) { /* post-order by levels */
		const int m = lvOff[d + 1] - lvOff[d];
		for(int a = 0; a < m; a += 65535) k_tree_up<<<dim3(gx, std::min(65535, m - a)), 256>>>(t, mdl, dOrd + lvOff[d] + a);
	}
	for(int d = 1; d <= maxD; ++d) { /* pre-order by levels */
		const int m = lvOff[d + 1] - lvOff[d];
		for(int a = 0; a < m; a += 65535) k_tree_down<<<dim3(gx, std::min(65535, m - a)), 256>>>(t, mdl, dOrd + lvOff[d] + a);
	}
	TCHK(hipGetLastError());
	TCHK(hipDeviceSynchronize());
	TCHK(hipMemcpy(seq, dSeq, (size_t) n * cs_len, hipMemcpyDeviceToHost));
	#undef TCHK
	if(height) { /* calcNodeHeight: distance to the nearest descendant leaf (src/PhyloTreeUnrooted.cpp:274-287) */
		for(int i = 0; i < n; ++i) height[i] = cnt[i] == cnt[i + 1] ? 0.0 : kInf;
		for(int h = n - 1; h > 0; --h) { const int u = order[h], p = parent[u]; height[p] = std::min(height[p], height[u] + blen[u]); }
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_tree_evaluate"); }

/* ------------------------------------------------------------------------------ batch */

typedef HuPlaceRec HostPlace;     /* a candidate's PTPlacement: the same record on the device (hu_kern_rank.h) and in the host mirror */

/* Kernel-selection and diagnostic knobs of a batch.  Defaults come from the environment ONCE, when the batch is created
 * (HU_<NAME>, e.g. HU_XCD_MAP=0; a variable that is set but not a number counts as 1); hu_batch_set_knob changes one
 * afterwards.  Nothing on the per-step path reads the environment. */
struct HuKnobs {
	int viterbi_hbm = 0;         /* HBM-staged value-filing Viterbi for every read                                   */
	int viterbi_values = 0;      /* LDS wavefront that files (M, I, D) of every cell instead of decision bytes       */
	int viterbi_mode = 0;        /* 1 generic workgroup decision-byte kernel, 2 row-per-thread workgroup kernel       */
	int viterbi_dec1 = 0;        /* alias of viterbi_mode = 1                                                        */
	int vw_diag = 0;             /* k_viterbi_wave diagnostics variant                                               */
	int viterbi_force_redo = 0;  /* flag every traceback "needs values": the redo pass runs for all sequences        */
	int tile_unsorted = 0;       /* scan tiles in read order instead of sorted by region start                       */
	int pairs32 = 0;             /* 32-bit (d, N) pairs even when every read has <= 255 bases                        */
	int topk_fast_min = 16384;   /* trees smaller than this take the exact two-pass histogram in k_seed_topk         */
	int topk_general = 0;        /* every read through the general top-k launch (k_seed_topk_d<DT, true>) instead of the straight kernel */
	int scan_pairs = 0;          /* 1: the full (d, N) pair matrix + k_seed_topk on large trees too; -1: the distance-only scan even with many partial sequences */
	int streaming_sep = 0;       /* one-wave streaming estimate / place kernels                                      */
	int est_unsorted = 0, place_unsorted = 0;   /* launch in read order instead of node order                        */
	int xcd_map = 1;             /* an eighth of the node-sorted list per XCD                                        */
	int est_var = 0, place_var = 0;             /* alternative kernel variants (comparison / diagnostics)            */
	int place_nosplit = 0;       /* column-order placement kernel even when the gap / base split applies             */
	int place_lds_pad = 0;       /* KB of unused dynamic LDS per placement workgroup: fewer of them per CU (experiment) */
	int est_lds_pad = 0;         /* the same for the estimate kernel                                                  */
	int vit_lds_pad = 0;         /* the same for the one-wave Viterbi kernel                                          */
	int scan_lds_pad = 0;        /* the same for the distance-only scan                                               */
	int trace = 0;               /* one line per stage decision to stderr                                            */
	int width_split = 1;         /* 0: one launch of the estimate / placement kernels for the whole batch, shaped by its widest region (rounds 1-2) */
	int sort_seq = 0;            /* filterPlacements / the final sort by the restated std::sort for every read (else only where keys tie) */
	int refsort_wgs = HU_RS_WGS_PER_CU * 256;       /* resident workgroups of k_seed_refsort (three per CU fill its LDS and wave slots; fewer leave room for the other batches' kernels beside it) */
	int ref_nofuse = 0;          /* HU_SEED_ORDER_LIBSTDCXX: k_seed_refsort counts its level 0 itself (pass A over the pair row) instead of starting from the stopper masks the scan leaves */
	int refsort_host = 0;        /* HU_SEED_ORDER_LIBSTDCXX: the host restatement of libstdc++'s sort for every read instead of the device kernel (k_seed_refsort) */
	int inject_fault = 0;        /* fault injection for the tests of the exception barrier: 1 = std::bad_alloc inside a worker of the filter stage's
	                              * host pool, 2 = std::length_error on the calling thread of the finish stage, 3 = std::runtime_error in a pool worker
	                              * of the TSV formatter.  Never set by the product */
};
struct HuKnobEntry { const char* name; int HuKnobs::* field; };
static const HuKnobEntry kKnobs[] = {
	{"viterbi_hbm", &HuKnobs::viterbi_hbm}, {"viterbi_values", &HuKnobs::viterbi_values}, {"viterbi_mode", &HuKnobs::viterbi_mode},
	{"viterbi_dec1", &HuKnobs::viterbi_dec1}, {"vw_diag", &HuKnobs::vw_diag}, {"viterbi_force_redo", &HuKnobs::viterbi_force_redo},
	{"pairs32", &HuKnobs::pairs32}, {"tile_unsorted", &HuKnobs::tile_unsorted}, {"topk_fast_min", &HuKnobs::topk_fast_min}, {"scan_pairs", &HuKnobs::scan_pairs}, {"topk_general", &HuKnobs::topk_general}, {"streaming_sep", &HuKnobs::streaming_sep},
	{"est_unsorted", &HuKnobs::est_unsorted}, {"place_unsorted", &HuKnobs::place_unsorted}, {"xcd_map", &HuKnobs::xcd_map},
	{"est_var", &HuKnobs::est_var}, {"place_var", &HuKnobs::place_var}, {"place_nosplit", &HuKnobs::place_nosplit}, {"width_split", &HuKnobs::width_split},
	{"place_lds_pad", &HuKnobs::place_lds_pad}, {"est_lds_pad", &HuKnobs::est_lds_pad}, {"vit_lds_pad", &HuKnobs::vit_lds_pad}, {"scan_lds_pad", &HuKnobs::scan_lds_pad}, {"trace", &HuKnobs::trace}, {"inject_fault", &HuKnobs::inject_fault}, {"refsort_host", &HuKnobs::refsort_host}, {"ref_nofuse", &HuKnobs::ref_nofuse}, {"refsort_wgs", &HuKnobs::refsort_wgs}, {"sort_seq", &HuKnobs::sort_seq},
};
static void knobs_from_env(HuKnobs& k) {
	for(const HuKnobEntry& e : kKnobs) {
		std::string env = "HU_";
		for(const char* c = e.name; *c; ++c) env += (char) toupper(*c);
		const char* v = getenv(env.c_str());
		if(!v) continue;
		char* end = nullptr;
		const long x = strtol(v, &end, 10);
		k.*(e.field) = (end == v) ? 1 : (int) x;
	}
}

enum { ST_NONE = 0, ST_READS = 1, ST_ALIGNED = 2, ST_SEEDED = 3, ST_ESTIMATED = 4, ST_FILTERED = 5, ST_PLACED = 6, ST_FINISHED = 7 };

struct hu_batch {
	hu_db* db = nullptr;
	int maxReads = 0, n = 0, nSeq = 0, state = ST_NONE;
	bool paired = false, fromCodes = false, profile = false;
	HuKnobs knob;
	int maxBases = 1 << 30;      /* most bases any read of the batch can have inside its region (set with the reads) */
	bool fixedRoot = false;      /* the last place call computed the intended root logliks (hu_opts.fix_root_loglik) */
	bool placesGiven = false;    /* the placed candidates came from the caller (hu_batch_set_candidates, placed): hu_finish_batch takes them as they are */
	bool pair16 = false;         /* the pair matrix of the last seed scan holds 16-bit pairs (maxBases <= 255) */
	int scanWidth = 0;           /* bytes per distance of the distance-only scan's matrix in dPairs (0 = none) */
	int pairsKind = 0;           /* what dPairs holds after the last seed stage: 16 / 32-bit (d, N) pairs, or 0 = no pair matrix (distance-only scan, given seeds) */
	int seedCap = HU_MAX_SEEDS;   /* most seeds any read of the batch can have (seed stage) */
	hipStream_t stream = nullptr;
	hipEvent_t ev[2 * HU_T_COUNT];
	bool evSet[HU_T_COUNT] = {false};
	float ms[HU_T_COUNT];
	double wall[4] = {0, 0, 0, 0};
	/* device */
	DBuf<char> dBases, dTraces, dRows;
	DBuf<HuReadDesc> dDescs;
	DBuf<double> dScratch;
	DBuf<uint8_t> dDec;
	DBuf<HuVitOut> dVit;
	DBuf<HuAlnDev> dAlns;
	DBuf<int8_t> dCodes;
	DBuf<int32_t> dStart, dEnd, dSeedCnt, dSeedId, dGiven, dPermCnt;
	DBuf<uint16_t> dPerm;
	PinnedVec<int32_t> hPermCnt;
	DBuf<uint32_t> dRp, dPairs, dSeedDN, dParDN, dBmin, dRSpan, dTileSpan;     /* dRSpan / dTileSpan: uint2 per read / tile */
	DBuf<int32_t> dTileQ, dSlotRead, dReadSlot;
	DBuf<uint32_t> dRq;
	DBuf<unsigned long long> dRefScratch;      /* k_seed_refsort: two key arrays + the level tables per resident workgroup */
	DBuf<int32_t> dHv, dZeroPar; DBuf<unsigned long long> dPairsC; int refHvCount = 0; double refHvHeight = NAN;      /* the reference-order mode under a height filter: the nodes that pass, the compacted pair rows */
	DBuf<int32_t> dBail, dNanCnt, dNanId; DBuf<uint32_t> dNanDN;      /* dNan*: the (dist, node id) selection of a batch whose reads met NaN distances in the reference-order mode */
	int nRefBail = 0;                            /* reads of the last seed stage that the device sort left to the host */
	DBuf<uint32_t> dRefPiv; DBuf<unsigned long long> dRefL0; bool refFused = false;      /* level 0 of the device sort prepared by the scan: pivots [n][4], stopper masks [n][nNodesPad / 64][2] */
	int refHostAll = 0;                          /* the last seed stage in the reference's order ran entirely on the host path (tree beyond the device sort) */
	DBuf<int32_t> dIns, dTileIns, dRetry;     /* dRetry: [0] = count, then the reads the straight top-k launch left to the general one */
	DBuf<uint32_t> dSortK, dSortV;
	DBuf<uint32_t> dCsCnt;      /* counting sort of the launch orders (hu_kern_rank.h): counters, run starts, cursors */
	DBuf<HuEstOut> dEst;
	DBuf<HuCand> dCands;
	DBuf<HuPlaceOut> dPlaceOut;
	/* the <= 64-record stages on the device (hu_kern_rank.h): candidates per read, their offsets, the seed slots in filter order, every
	 * candidate's PTPlacement, the best one per read; meta = {candidates, largest gap-site count, largest base-site count} */
	DBuf<int32_t> dCandCnt, dCandOff, dMeta;
	DBuf<uint8_t> dFiltSlot;
	DBuf<HuPlaceRec> dPlaces;
	DBuf<hu_place_rec> dBest;
	size_t nc = 0;                 /* candidates of the batch (after the filter stage / hu_batch_set_candidates) */
	int maxGapSites = 0, maxBaseSites = 0;
	bool hostCands = false;        /* candOffs / places below mirror the device's (filled on demand: getters, the chimera check) */
	DBuf<double> dRootLL;
	PinnedVec<double> hRootLL;
	/* host */
	std::vector<HuReadDesc> hDescs;
	std::vector<char> hBases;
	PinnedVec<HuVitOut> hVit;
	int64_t cellsTotal = 0, cornerTotal = 0;   /* DP cells of all phases of all sequences / of their corner blocks (set with the reads) */
	int nVitRedo = 0;           /* sequences of the last align call redone by the value-filing Viterbi */
	int rMain = 0;              /* width split of the batch (plan_width_split): regions of at most rMain columns take the main launch; 0 = one launch for all */
	std::vector<int32_t> wideReads;     /* the reads beyond it */
	std::vector<uint32_t> hWideOrd, hWideCand; DBuf<uint32_t> dWideOrd, dWideCand;      /* their (read, seed) slots and their candidates: the lists of the second launches */
	PinnedVec<int32_t> hCandOffAll;   /* candidate offsets on the host when there are wide reads (scan_cands) */
	int nFullRedo = 0;          /* sequences of the last align call whose banded DP found no path: full DP, one launch */
	DBuf<double> dRedoScr; DBuf<HuReadDesc> dRedoDesc; DBuf<HuVitOut> dRedoVit;    /* the redo launches' own scratch: kept (an allocation or a release stalls every stream of the device) */
	PinnedVec<HuAlnDev> hAlns;
	PinnedVec<int32_t> hStart, hEnd, hSeedCnt, hSeedId;
	PinnedVec<uint32_t> hSeedDN;
	PinnedVec<HuEstOut> hEst;
	PinnedVec<HuCand> hCands;
	PinnedVec<HuPlaceOut> hPlaceOut;
	std::vector<int64_t> candOffs;
	std::vector<HostPlace> places;    /* candidates in filterPlacements order, all reads */
	std::vector<HostPlace> tmpPlaces;
	PinnedVec<hu_place_rec> best;      /* page-locked: a device-to-host copy into pageable memory makes the runtime wait (spinning) for the stream inside the call */
	PinnedVec<int32_t> hMeta, hBail;   /* the same for the few words the host reads between stages */
	PinnedVec<char> hRows;            /* alignment rows of the last format call */
	std::vector<char> tsvBuf; std::vector<size_t> tsvOff, tsvLen; size_t tsvSize = 0;
	int maxRegion = 0;
};

struct Timer {
	hu_batch* b; int id;
	Timer(hu_batch* b, int id) : b(b), id(id) { if(b->profile) (void) hipEventRecord(b->ev[2 * id], b->stream); }
	~Timer() { if(b->profile) { (void) hipEventRecord(b->ev[2 * id + 1], b->stream); b->evSet[id] = true; } }
};

extern "C" int hu_batch_create(hu_db* db, int max_reads, hu_batch** out) try {
	if(!db || !out || max_reads < 1) { hu_set_error("hu_batch_create: bad argument"); return HU_ERR_ARG; }
	HIPCHK(hipSetDevice(db->device));
	hu_batch* b = new hu_batch;
	b->db = db; b->maxReads = max_reads;
	knobs_from_env(b->knob);
	for(int i = 0; i < 2 * HU_T_COUNT; ++i) b->ev[i] = nullptr;
	for(int i = 0; i < HU_T_COUNT; ++i) b->ms[i] = 0;
	hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
	for(int i = 0; i < 2 * HU_T_COUNT && e == hipSuccess; ++i) e = hipEventCreate(&b->ev[i]);
	if(e != hipSuccess) { /* nothing of a half-made batch stays behind */
		hu_set_error("hu_batch_create: stream / event creation failed: %s", hipGetErrorString(e));
		for(int i = 0; i < 2 * HU_T_COUNT; ++i) if(b->ev[i]) (void) hipEventDestroy(b->ev[i]);
		if(b->stream) (void) hipStreamDestroy(b->stream);
		delete b;
		return HU_ERR_DEVICE;
	}
	*out = b;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_create"); }
extern "C" int hu_batch_set_knob(hu_batch* b, const char* name, int value) try {
	if(!b || !name) return HU_ERR_ARG;
	for(const HuKnobEntry& e : kKnobs) if(strcmp(e.name, name) == 0) { b->knob.*(e.field) = value; return HU_OK; }
	hu_set_error("hu_batch_set_knob: no knob named '%s'", name);
	return HU_ERR_ARG;
} catch(...) { return hu_catch_all("hu_batch_set_knob"); }
extern "C" void hu_batch_destroy(hu_batch* b) try {
	if(!b) return;
	(void) hipSetDevice(b->db->device);
	(void) hu_wait(b->stream);
	for(int i = 0; i < 2 * HU_T_COUNT; ++i) (void) hipEventDestroy(b->ev[i]);
	(void) hipStreamDestroy(b->stream);
	delete b;      /* the DBuf members free their device memory */
} catch(...) { (void) hu_catch_all("hu_batch_destroy"); }
extern "C" int hu_batch_sync(hu_batch* b) try { if(!b) return HU_ERR_ARG; HIPCHK(hu_wait(b->stream)); return HU_OK; } catch(...) { return hu_catch_all("hu_batch_sync"); }
extern "C" int hu_batch_profile(hu_batch* b, int enable) try { if(!b) return HU_ERR_ARG; b->profile = enable != 0; return HU_OK; } catch(...) { return hu_catch_all("hu_batch_profile"); }
extern "C" int hu_batch_timings(hu_batch* b, float* ms) try {
	if(!b || !ms) return HU_ERR_ARG;
	HIPCHK(hu_wait(b->stream));
	for(int i = 0; i < HU_T_COUNT; ++i) {
		float t = 0;
		if(b->profile && b->evSet[i] && hipEventElapsedTime(&t, b->ev[2 * i], b->ev[2 * i + 1]) == hipSuccess) b->ms[i] = t;
		ms[i] = b->ms[i];
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_timings"); }

/* banded phases of calcViterbiScores as region descriptors (src/BandedHMMP7.cpp:794-881) */
static void build_regions(const hu_db* db, int L, const int32_t* vp /* [2][6] */, HuReadDesc& rd) {
	const int K = db->dev.K;
	const double kMinGapFrac = 0.2;
	struct VP { int start, end, from, to, nIns, nDel; } v[2];
	int nv = 0;
	for(int p = 0; p < 2 && vp; ++p) {
		VP x{vp[p*6], vp[p*6+1], vp[p*6+2], vp[p*6+3], vp[p*6+4], vp[p*6+5]};
		bool valid = x.start > 0 && x.start <= x.end && x.from > 0 && x.from <= x.to && x.nIns >= 0 && x.nDel >= 0;
		if(valid && x.end <= K && x.to <= L) v[nv++] = x;
	}
	rd.nRegions = 0;
	int64_t off = 0, doff = 0;
	auto add = [&](int j0, int j1, int i0, int i1, int withB, int band, const VP* bp) {
		HuRegion& g = rd.reg[rd.nRegions++];
		g.j0 = j0; g.j1 = j1; g.i0 = i0; g.i1 = i1; g.withB = withB; g.band = band;
		g.from = bp ? bp->from : 0; g.start = bp ? bp->start : 0; g.nIns = bp ? bp->nIns : 0; g.nDel = bp ? bp->nDel : 0;
		g.off = off; g.doff = doff;
		if(j1 >= j0 && i1 >= i0) {
			off += (int64_t)(j1 - j0 + 1) * (i1 - i0 + 1);
			/* room for the layout of whichever decision-byte kernel runs: anti-diagonals with a pitch of ni rounded
			 * up to 16 (k_viterbi_dec, _dec2), steps of 64 lanes x 4 or 8 rows or x 1 row (k_viterbi_wave) */
			const int64_t ni = i1 - i0 + 1, nj = j1 - j0 + 1;
			int64_t sz = (ni + nj - 1) * ((ni + 15) & ~15);
			sz = std::max(sz, (ni + nj - 1) * 64);
			sz = std::max(sz, (nj + (ni + 3) / 4 - 1) * 256);
			sz = std::max(sz, (nj + (ni + 7) / 8 - 1) * 512);
			doff += (sz + 15) & ~(int64_t) 15;
		}
	};
	if(nv == 0) add(1, K, 1, L, 1, 0, nullptr); /* full Viterbi (src/BandedHMMP7.cpp:748-771) */
	else {
		for(int p = 0; p < nv; ++p) {
			int upQLen = p == 0 ? v[p].from - 1 : v[p].from - v[p - 1].to;
			if(upQLen < 0) upQLen = 0;
			int up_start = p == 0 ? v[p].start - upQLen * (1 + kMinGapFrac) : v[p - 1].end;
			if(up_start < 1) up_start = 1;
			int up_from = p == 0 ? v[p].from - upQLen * (1 + kMinGapFrac) : v[p - 1].to;
			if(up_from < 1) up_from = 1;
			add(up_start, v[p].start, up_from, v[p].from, 1, 0, nullptr);
			add(v[p].start, v[p].end, v[p].from, v[p].to, 1, 1, &v[p]);
		}
		const VP& last = v[nv - 1];
		int downQLen = L - last.to;
		int down_end = last.end + downQLen * (1 + kMinGapFrac);
		int down_to = last.to + downQLen * (1 + kMinGapFrac);
		if(down_end > K) down_end = K;
		if(down_to > L) down_to = L;
		add(last.end, down_end, last.to, down_to, 0, 0, nullptr);
	}
	/* corner blocks (HuRegion::ci0 / cj0 / coff): what a later phase can look up of an earlier one */
	int64_t coff = 0;
	for(int r = 0; r < rd.nRegions; ++r) {
		HuRegion& g = rd.reg[r];
		int nearI = INT32_MAX, nearJ = INT32_MAX;
		for(int r2 = r + 1; r2 < rd.nRegions; ++r2) {
			const HuRegion& g2 = rd.reg[r2];
			if(g2.i1 >= g2.i0 && g2.j1 >= g2.j0) { nearI = std::min(nearI, g2.i0 - 1); nearJ = std::min(nearJ, g2.j0 - 1); }
		}
		g.ci0 = std::max(g.i0, nearI); g.cj0 = std::max(g.j0, nearJ); g.coff = coff;
		if(nearI == INT32_MAX || g.ci0 > g.i1 || g.cj0 > g.j1 || g.i1 < g.i0 || g.j1 < g.j0) { g.ci0 = g.i1 + 1; g.cj0 = g.j1 + 1; }   /* empty block */
		else coff += (int64_t)(g.i1 - g.ci0 + 1) * (g.j1 - g.cj0 + 1);
	}
	rd.scratchOff = off; /* total cells for now; turned into an offset by the caller */
	rd.decOff = doff;    /* likewise: total decision bytes */
	rd.cornerOff = coff; /* likewise: total corner cells */
}

static int upload_descs(hu_batch* b) {
	int rc;
	if((rc = b->dDescs.ensure(b->hDescs.size())) != HU_OK) return rc;
	HIPCHK(hipMemcpyAsync(b->dDescs.p, b->hDescs.data(), b->hDescs.size() * sizeof(HuReadDesc), hipMemcpyHostToDevice, b->stream));
	return HU_OK;
}

extern "C" int hu_batch_set_reads(hu_batch* b, int n, const char* bases, const int64_t* offs, const int32_t* vpaths,
		const char* mates, const int64_t* moffs, const int32_t* mvpaths) try {
	if(!b || n < 0 || n > b->maxReads || (n > 0 && (!bases || !offs))) { hu_set_error("hu_batch_set_reads: bad argument"); return HU_ERR_ARG; }
	if(mates && !moffs) { hu_set_error("hu_batch_set_reads: mates without offsets"); return HU_ERR_ARG; }
	HIPCHK(hipSetDevice(b->db->device));
	b->n = n; b->paired = mates != nullptr; b->nSeq = b->paired ? 2 * n : n; b->fromCodes = false;
	b->hDescs.assign(b->nSeq, HuReadDesc());
	b->hBases.clear();
	b->maxBases = 0;
	for(int r = 0; r < n; ++r) { /* a merged pair holds at most the bases of both mates */
		const int64_t l = (offs[r + 1] - offs[r]) + (mates ? moffs[r + 1] - moffs[r] : 0);
		if(l > b->maxBases) b->maxBases = (int) std::min<int64_t>(l, 1 << 30);
	}
	int64_t cells = 0, tr = 0, decs = 0, corners = 0;
	for(int s = 0; s < b->nSeq; ++s) {
		const bool isMate = s >= n;
		const int r = isMate ? s - n : s;
		const char* src = isMate ? mates + moffs[r] : bases + offs[r];
		const int64_t len64 = isMate ? moffs[r + 1] - moffs[r] : offs[r + 1] - offs[r];
		HuReadDesc& rd = b->hDescs[s];
		memset(&rd, 0, sizeof(rd));
		rd.baseOff = (int64_t) b->hBases.size();
		rd.len = (int32_t) len64;
		bool bad = len64 < 1 || len64 > 65535;
		for(int64_t i = 0; i < len64 && !bad; ++i) if(host_sym(src[i]) < 0) bad = true; /* PrimarySeq::encodeAt < 0 */
		if(!bad) {
			b->hBases.insert(b->hBases.end(), src, src + len64);
			const int32_t* vp = isMate ? (mvpaths ? mvpaths + (size_t) r * 12 : nullptr) : (vpaths ? vpaths + (size_t) r * 12 : nullptr);
			build_regions(b->db, rd.len, vp, rd);
			const int64_t c = rd.scratchOff, dc = rd.decOff, cc = rd.cornerOff;
			rd.scratchOff = cells; cells += c;
			rd.decOff = decs; decs += dc;
			rd.cornerOff = corners; corners += cc;
		}
		else { rd.len = 0; rd.nRegions = 0; }
		rd.traceOff = tr;
		tr += rd.len + b->db->dev.K + 8;
	}
	int rc;
	if((rc = b->dBases.ensure(b->hBases.size() + 1)) != HU_OK) return rc;
	b->cellsTotal = cells; b->cornerTotal = corners;    /* the DP scratch is sized by hu_align_batch: which kernel runs decides how much it needs */
	if((rc = b->dDec.ensure((size_t) decs + 1)) != HU_OK) return rc;
	if((rc = b->dTraces.ensure((size_t) tr + 1)) != HU_OK) return rc;
	if((rc = b->dVit.ensure(b->nSeq)) != HU_OK) return rc;
	if(!b->hBases.empty()) HIPCHK(hipMemcpyAsync(b->dBases.p, b->hBases.data(), b->hBases.size(), hipMemcpyHostToDevice, b->stream));
	if((rc = upload_descs(b)) != HU_OK) return rc;
	b->state = ST_READS;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_set_reads"); }

static int ensure_read_buffers(hu_batch* b) {
	const HuDbDev& d = b->db->dev;
	const size_t n = (size_t) b->n;
	const size_t tiles = (n + HU_READ_TILE - 1) / HU_READ_TILE;
	int rc;
	if((rc = b->dCodes.ensure(n * d.csLen)) != HU_OK) return rc;
	if((rc = b->dStart.ensure(n)) != HU_OK) return rc;
	if((rc = b->dEnd.ensure(n)) != HU_OK) return rc;
	if((rc = b->dRp.ensure(tiles * d.WQ * HU_READ_TILE * 16)) != HU_OK) return rc;
	if((rc = b->dTileQ.ensure(tiles * (d.WQ + 1))) != HU_OK) return rc;
	if((rc = b->dSlotRead.ensure(std::max<size_t>(tiles, 1) * HU_READ_TILE)) != HU_OK) return rc;
	if((rc = b->dReadSlot.ensure(std::max<size_t>(n, 1))) != HU_OK) return rc;
	if((rc = b->dRq.ensure(std::max<size_t>(n, 1) * ((d.WQ + 31) / 32))) != HU_OK) return rc;
	if((rc = b->dIns.ensure(std::max<size_t>(n, 1) * (HU_MAX_INS + 1))) != HU_OK) return rc;
	if((rc = b->dTileIns.ensure(std::max<size_t>(tiles, 1) * (HU_READ_TILE * HU_MAX_INS + 1))) != HU_OK) return rc;
	if((rc = b->dRSpan.ensure(std::max<size_t>(n, 1) * 2)) != HU_OK) return rc;
	if((rc = b->dTileSpan.ensure(std::max<size_t>(tiles, 1) * 2)) != HU_OK) return rc;
	return HU_OK;
}

/* out[0 .. n) = the values in the order of their keys; keys beyond `bound` count as bound.  runsInOrder: the values of one key in ascending order (the scan's tiling:
 * the same tiles run after run); else in the order the scatter's atomics gave them (the launch orders: which workgroup of a node's run comes first changes nothing).
 * A counting sort on the batch's stream */
static int order_by_key(hu_batch* b, const uint32_t* key, const uint32_t* val, size_t n, uint32_t bound, uint32_t* out, bool runsInOrder) {
	if(!n) return HU_OK;
	const size_t m = (size_t) bound + 1;
	int rc;
	const size_t nTiles = (m + 1023) / 1024;
	if((rc = b->dCsCnt.ensure(3 * m + nTiles + 8)) != HU_OK) return rc;
	uint32_t* cnt = b->dCsCnt.p; uint32_t* start = cnt + m; uint32_t* cur = start + m + 1; uint32_t* tiles = cur + m;
	HIPCHK(hipMemsetAsync(cnt, 0, m * 4, b->stream));
	k_cs_hist<<<(unsigned)((n + 255) / 256), 256, 0, b->stream>>>((int) n, key, bound, cnt);
	k_cs_tile_sums<<<(unsigned) nTiles, 1024, 0, b->stream>>>((int) m, cnt, tiles);
	k_cs_scan_tiles<<<1, 1024, 0, b->stream>>>((int) nTiles, tiles);
	k_cs_scan<<<(unsigned) nTiles, 1024, 0, b->stream>>>((int) m, cnt, tiles, start, cur);
	k_cs_scatter<<<(unsigned)((n + 255) / 256), 256, 0, b->stream>>>((int) n, key, val, bound, cur, out);
	if(runsInOrder) k_cs_runs<<<(unsigned)((m + 3) / 4), 256, 0, b->stream>>>((int) bound, start, out);      /* (the overflow bucket — invalid slots — is left as scattered) */
	HIPCHK(hipGetLastError());
	return HU_OK;
}

/* the scan's tiling: reads sorted by the first column of their region (from the alignments, or from dStart / dEnd when alns is NULL) */
static int tile_reads(hu_batch* b, const HuAlnDev* alns) {
	const int n = b->n;
	if(!n) return HU_OK;
	const int tiles = (n + HU_READ_TILE - 1) / HU_READ_TILE;
	int rc;
	if((rc = b->dSortK.ensure((size_t) n * 2)) != HU_OK || (rc = b->dSortV.ensure((size_t) n * 2)) != HU_OK) return rc;
	k_tile_keys<<<(n + 255) / 256, 256, 0, b->stream>>>(n, alns, b->dStart.p, b->dEnd.p, b->dSortK.p, b->dSortV.p);
	if(b->knob.tile_unsorted) HIPCHK(hipMemcpyAsync(b->dSortV.p + n, b->dSortV.p, (size_t) n * 4, hipMemcpyDeviceToDevice, b->stream));   /* tiles in read order */
	else if((rc = order_by_key(b, b->dSortK.p, b->dSortV.p, (size_t) n, (uint32_t) b->db->dev.csLen + 1u, b->dSortV.p + n, true)) != HU_OK) return rc;      /* keys: 1-based first column, unplaced reads last */
	k_tile_slots<<<(tiles * HU_READ_TILE + 255) / 256, 256, 0, b->stream>>>(n, tiles * HU_READ_TILE, b->dSortV.p + n, b->dSlotRead.p, b->dReadSlot.p);
	HIPCHK(hipGetLastError());
	return HU_OK;
}

static int set_aligned_impl(hu_batch* b, int n, const int8_t* codes, hipMemcpyKind kind, const int32_t* start, const int32_t* end, int maxBases = -1) {
	if(!b || n < 0 || n > b->maxReads || (n > 0 && (!codes || !start || !end))) { hu_set_error("hu_batch_set_aligned: bad argument"); return HU_ERR_ARG; }
	HIPCHK(hipSetDevice(b->db->device));
	const HuDbDev& d = b->db->dev;
	b->n = n; b->nSeq = n; b->paired = false; b->fromCodes = true;
	if(maxBases >= 0) b->maxBases = maxBases;         /* device-to-device: the caller knows (segments of reads it holds) */
	else { /* host codes: count */
		int mb = 0;
		for(int r = 0; r < n; ++r) {
			if(!(start[r] >= 0 && start[r] <= end[r] && end[r] < d.csLen)) continue;
			const int8_t* c = codes + (size_t) r * d.csLen; int k = 0;
			for(int j = start[r]; j <= end[r]; ++j) k += c[j] >= 0;
			mb = std::max(mb, k);
		}
		b->maxBases = mb;
	}
	int rc;
	if((rc = ensure_read_buffers(b)) != HU_OK) return rc;
	b->hAlns.assign(n, HuAlnDev());
	b->hStart.assign(start, start + n); b->hEnd.assign(end, end + n);
	for(int r = 0; r < n; ++r) {
		HuAlnDev& a = b->hAlns[r];
		memset(&a, 0, sizeof(a));
		const bool ok = start[r] >= 0 && start[r] <= end[r] && end[r] < d.csLen && start[r] >= d.winStart && end[r] < d.winStart + d.winLen;
		a.status = ok ? HU_READ_OK : HU_READ_INVALID;
		a.csStart = start[r] + 1; a.csEnd = end[r] + 1;
		if(!ok) { b->hStart[r] = (int32_t) d.winStart; b->hEnd[r] = (int32_t) d.winStart - 1; }   /* empty, inside the resident window */
	}
	(void) hipGetLastError();
	if(n) {
		if((const void*) codes != (const void*) b->dCodes.p) HIPCHK(hipMemcpyAsync(b->dCodes.p, codes, (size_t) n * d.csLen, kind, b->stream));
		HIPCHK(hipMemcpyAsync(b->dStart.p, b->hStart.data(), (size_t) n * 4, hipMemcpyHostToDevice, b->stream));
		HIPCHK(hipMemcpyAsync(b->dEnd.p, b->hEnd.data(), (size_t) n * 4, hipMemcpyHostToDevice, b->stream));
		const int tiles = (n + HU_READ_TILE - 1) / HU_READ_TILE;
		HIPCHK(hipMemsetAsync(b->dRp.p, 0, (size_t) tiles * d.WQ * HU_READ_TILE * 16 * 4, b->stream));
		if((rc = tile_reads(b, nullptr)) != HU_OK) return rc;
		k_planes_from_codes<<<n, 64, 0, b->stream>>>(d, b->dCodes.p, b->dStart.p, b->dEnd.p, b->dRp.p, b->dRq.p, b->dIns.p, b->dReadSlot.p, (uint2*) b->dRSpan.p);
		k_tile_lists<<<tiles, 64, 0, b->stream>>>(d, n, b->dRq.p, b->dIns.p, b->dTileQ.p, b->dTileIns.p, b->dSlotRead.p, (const uint2*) b->dRSpan.p, (uint2*) b->dTileSpan.p);
		HIPCHK(hipGetLastError());
	}
	b->state = ST_ALIGNED;
	return HU_OK;
}
extern "C" int hu_batch_set_aligned(hu_batch* b, int n, const int8_t* codes, const int32_t* start, const int32_t* end) try {
	return set_aligned_impl(b, n, codes, hipMemcpyHostToDevice, start, end);
} catch(...) { return hu_catch_all("hu_batch_set_aligned"); }

extern "C" int hu_align_batch(hu_batch* b, const hu_opts* o) try {
	if(!b || !o) return HU_ERR_ARG;
	if(b->state < ST_READS || b->fromCodes) { hu_set_error("hu_align_batch: no reads set"); return HU_ERR_STATE; }
	HIPCHK(hipSetDevice(b->db->device));
	const HuDbDev& d = b->db->dev;
	const HuKnobs& kb = b->knob;
	double tNN, tNB, tEC, tCC;
	hu_mode_costs(d.K, o->align_mode, &tNN, &tNB, &tEC, &tCC);
	int rc;
	if((rc = ensure_read_buffers(b)) != HU_OK) return rc;
	if((rc = b->dRows.ensure((size_t) b->nSeq * d.csLen)) != HU_OK) return rc;
	if((rc = b->dAlns.ensure(b->nSeq)) != HU_OK) return rc;
	b->hVit.resize(b->nSeq);
	(void) hipGetLastError(); /* the HIP runtime is shared with torch: drop stale sticky errors that are not ours */
	if(b->nSeq) {
		int maxLen = 1;
		for(int s = 0; s < b->nSeq; ++s) maxLen = std::max(maxLen, (int) b->hDescs[s].len);
		const int ldsRows = maxLen + 1;
		bool usedDec = false;
		int decRpl = 0;
		const size_t vlds = (size_t) 9 * ldsRows * sizeof(double);
		/* The one-wave kernel files (M, I, D) only of the corner blocks (a few cells per read); every other kernel keeps all three values of
		 * every cell of every phase: 2-3 MB per 250-base read, 16 GB per batch of 8,192 — allocated only when such a kernel is going to run. */
		const bool wavePath = vlds <= 96 * 1024 && !kb.viterbi_hbm && !kb.viterbi_values && !(kb.viterbi_mode ? kb.viterbi_mode : (kb.viterbi_dec1 ? 1 : 0)) && maxLen <= 512;
		if((rc = b->dScratch.ensure((size_t)(wavePath ? b->cornerTotal : b->cellsTotal) * 3 + 1)) != HU_OK) return rc;
		{
			Timer t(b, HU_T_VITERBI);
			if(vlds <= 96 * 1024 && !kb.viterbi_hbm) { /* LDS-staged wavefront; longer reads take the HBM-staged kernel */
				if(!kb.viterbi_values) { /* one decision byte per cell; (M, I, D) only where a later phase looks */
					if(vlds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_viterbi_dec, hipFuncAttributeMaxDynamicSharedMemorySize, (int) vlds));
					#define VD_ARGS d, b->dDescs.p, b->dBases.p, b->dScratch.p, b->dDec.p, tNN, tNB, tEC, tCC, b->dVit.p, ldsRows
					int haloW = 2;
					for(int s = 0; s < b->nSeq; ++s) for(int r = 0; r < b->hDescs[s].nRegions; ++r)
						haloW = std::max(haloW, b->hDescs[s].reg[r].j1 - b->hDescs[s].reg[r].j0 + 3);
					const size_t vlds2 = vlds + (size_t) 3 * haloW * sizeof(double) + 32 * (maxLen <= 256 ? 256 : 512);
					const int mode = kb.viterbi_mode ? kb.viterbi_mode : (kb.viterbi_dec1 ? 1 : 0);   /* 1 = generic workgroup kernel, 2 = row-per-thread workgroup kernel */
					const int haloWw = std::min(haloW, 512);
					if(mode == 0 && maxLen <= 512) { /* one wave per sequence, no barrier */
						const size_t wl = (size_t) 3 * haloWw * sizeof(double) + (size_t)(kb.vit_lds_pad > 0 && kb.vit_lds_pad <= 40 ? kb.vit_lds_pad : 0) * 1024;
						const int dgv = kb.vw_diag;
						if(maxLen <= 256 && dgv == 1) { k_viterbi_wave<4, 1><<<b->nSeq, 64, wl, b->stream>>>(d, b->dDescs.p, b->dBases.p, b->dScratch.p, b->dDec.p, tNN, tNB, tEC, tCC, b->dVit.p, haloWw); decRpl = 4; }
						else if(maxLen <= 256 && dgv == 2) { k_viterbi_wave<4, 2><<<b->nSeq, 64, wl, b->stream>>>(d, b->dDescs.p, b->dBases.p, b->dScratch.p, b->dDec.p, tNN, tNB, tEC, tCC, b->dVit.p, haloWw); decRpl = 4; }
						else if(maxLen <= 256) { k_viterbi_wave<4><<<b->nSeq, 64, wl, b->stream>>>(d, b->dDescs.p, b->dBases.p, b->dScratch.p, b->dDec.p, tNN, tNB, tEC, tCC, b->dVit.p, haloWw); decRpl = 4; }
						else { k_viterbi_wave<8><<<b->nSeq, 64, wl, b->stream>>>(d, b->dDescs.p, b->dBases.p, b->dScratch.p, b->dDec.p, tNN, tNB, tEC, tCC, b->dVit.p, haloWw); decRpl = 8; }
					}
					else if(mode != 1 && maxLen <= 256 && vlds2 <= 96 * 1024) { /* one DP row per thread, nothing global inside the wavefront */
						if(vlds2 > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_viterbi_dec2<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) vlds2));
						k_viterbi_dec2<256><<<b->nSeq, 256, vlds2, b->stream>>>(VD_ARGS, haloW);
					}
					else if(mode != 1 && maxLen <= 512 && vlds2 <= 96 * 1024) {
						if(vlds2 > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_viterbi_dec2<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) vlds2));
						k_viterbi_dec2<512><<<b->nSeq, 512, vlds2, b->stream>>>(VD_ARGS, haloW);
					}
					else k_viterbi_dec<<<b->nSeq, HU_VIT_THREADS, vlds, b->stream>>>(VD_ARGS);
					#undef VD_ARGS
					k_viterbi_trace_dec<<<(b->nSeq + 63) / 64, 64, 0, b->stream>>>(d, b->dDescs.p, b->dDec.p, b->dTraces.p, b->dVit.p, b->nSeq, kb.viterbi_force_redo != 0, decRpl);
					usedDec = true;
				}
				else {
					if(vlds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_viterbi_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int) vlds));
					k_viterbi_lds<<<b->nSeq, HU_VIT_THREADS, vlds, b->stream>>>(d, b->dDescs.p, b->dBases.p, b->dScratch.p, b->dTraces.p, tNN, tNB, tEC, tCC, b->dVit.p, ldsRows, 0);
					k_viterbi_trace<<<(b->nSeq + 63) / 64, 64, 0, b->stream>>>(d, b->dDescs.p, b->dScratch.p, b->dTraces.p, tNN, tNB, b->dVit.p, b->nSeq);
				}
			}
			else k_viterbi<<<b->nSeq, 64, 0, b->stream>>>(d, b->dDescs.p, b->dBases.p, b->dScratch.p, b->dTraces.p, tNN, tNB, tEC, tCC, b->dVit.p);
		}
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(b->hVit.data(), b->dVit.p, (size_t) b->nSeq * sizeof(HuVitOut), hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hu_wait(b->stream));
		if(usedDec) { /* sequences whose traceback cannot trust the fill-time decisions: redo with every value filed */
			int nRedo = 0;
			for(int s = 0; s < b->nSeq; ++s) if(b->hVit[s].status == HU_READ_NEEDS_VALUES) nRedo++;
			b->nVitRedo = nRedo;
			if(nRedo && wavePath) { /* the batch holds corner scratch only: the few sequences to redo get a full-size scratch of their own, in launches of <= 48 M cells */
				if(vlds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_viterbi_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int) vlds));
				std::vector<int> rs;
				for(int s = 0; s < b->nSeq; ++s) if(b->hVit[s].status == HU_READ_NEEDS_VALUES) rs.push_back(s);
				const int64_t cap = 48ll << 20;
				for(size_t at = 0; at < rs.size();) {
					std::vector<HuReadDesc> rdv; std::vector<HuVitOut> hv;
					int64_t cells = 0; size_t e = at;
					for(; e < rs.size(); ++e) {
						HuReadDesc rd = b->hDescs[rs[e]];
						int64_t c = 0;                         /* cells of all its phases, as build_regions laid them out */
						for(int g = 0; g < rd.nRegions; ++g) if(rd.reg[g].j1 >= rd.reg[g].j0 && rd.reg[g].i1 >= rd.reg[g].i0) c += (int64_t)(rd.reg[g].j1 - rd.reg[g].j0 + 1) * (rd.reg[g].i1 - rd.reg[g].i0 + 1);
						if(!rdv.empty() && cells + c > cap) break;
						rd.scratchOff = cells; cells += c;
						rdv.push_back(rd); hv.push_back(b->hVit[rs[e]]);
					}
					DBuf<double>& scr = b->dRedoScr; DBuf<HuReadDesc>& dd = b->dRedoDesc; DBuf<HuVitOut>& vo = b->dRedoVit;
					if((rc = scr.ensure((size_t) cells * 3 + 1)) != HU_OK || (rc = dd.ensure(rdv.size())) != HU_OK || (rc = vo.ensure(rdv.size())) != HU_OK) return rc;
					HIPCHK(hipMemcpyAsync(dd.p, rdv.data(), rdv.size() * sizeof(HuReadDesc), hipMemcpyHostToDevice, b->stream));
					HIPCHK(hipMemcpyAsync(vo.p, hv.data(), hv.size() * sizeof(HuVitOut), hipMemcpyHostToDevice, b->stream));
					k_viterbi_lds<<<(unsigned) rdv.size(), HU_VIT_THREADS, vlds, b->stream>>>(d, dd.p, b->dBases.p, scr.p, b->dTraces.p, tNN, tNB, tEC, tCC, vo.p, ldsRows, HU_READ_NEEDS_VALUES);
					k_viterbi_trace<<<((unsigned) rdv.size() + 63) / 64, 64, 0, b->stream>>>(d, dd.p, scr.p, b->dTraces.p, tNN, tNB, vo.p, (int) rdv.size());
					HIPCHK(hipGetLastError());
					HIPCHK(hipMemcpyAsync(hv.data(), vo.p, hv.size() * sizeof(HuVitOut), hipMemcpyDeviceToHost, b->stream));
					HIPCHK(hu_wait(b->stream));
					for(size_t k = 0; k < rdv.size(); ++k) {
						b->hVit[rs[at + k]] = hv[k];               /* the trace is written at the sequence's own traceOff */
						HIPCHK(hipMemcpyAsync(b->dVit.p + rs[at + k], &b->hVit[rs[at + k]], sizeof(HuVitOut), hipMemcpyHostToDevice, b->stream));
					}
					HIPCHK(hu_wait(b->stream));
					at = e;
				}
			}
			else if(nRedo) {
				if(vlds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_viterbi_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int) vlds));
				k_viterbi_lds<<<b->nSeq, HU_VIT_THREADS, vlds, b->stream>>>(d, b->dDescs.p, b->dBases.p, b->dScratch.p, b->dTraces.p, tNN, tNB, tEC, tCC, b->dVit.p, ldsRows, HU_READ_NEEDS_VALUES);
				k_viterbi_trace<<<(b->nSeq + 63) / 64, 64, 0, b->stream>>>(d, b->dDescs.p, b->dScratch.p, b->dTraces.p, tNN, tNB, b->dVit.p, b->nSeq);
				HIPCHK(hipGetLastError());
				HIPCHK(hipMemcpyAsync(b->hVit.data(), b->dVit.p, (size_t) b->nSeq * sizeof(HuVitOut), hipMemcpyDeviceToHost, b->stream));
				HIPCHK(hu_wait(b->stream));
			}
		}
		/* banded version failed -> regular HMM (src/HmmUFOtu_main.cpp:89-93): all such sequences in ONE launch, each with a
		 * full-DP descriptor of its own (same bases, same trace slot) and its share of one temporary scratch */
		std::vector<int> redo;
		for(int s = 0; s < b->nSeq; ++s) if(b->hVit[s].status == HU_READ_NEEDS_FULL && !(b->hDescs[s].nRegions == 1 && !b->hDescs[s].reg[0].band)) redo.push_back(s);
		b->nFullRedo = (int) redo.size();
		/* launches of at most 48 M cells (1.2 GB of (M, I, D) scratch): ~130 reads of 250 bp against K = 1,400 each */
		const int64_t cellCap = 48ll << 20;
		for(size_t at = 0; at < redo.size();) {
			std::vector<HuReadDesc> rdv;
			int64_t cells = 0;
			size_t e = at;
			for(; e < redo.size(); ++e) {
				HuReadDesc rd = b->hDescs[redo[e]];
				build_regions(b->db, rd.len, nullptr, rd);
				const int64_t c = rd.scratchOff;
				if(!rdv.empty() && cells + c > cellCap) break;
				rd.scratchOff = cells; cells += c;
				rdv.push_back(rd);
			}
			DBuf<double>& scr = b->dRedoScr; DBuf<HuReadDesc>& dd = b->dRedoDesc; DBuf<HuVitOut>& vo = b->dRedoVit;
			if((rc = scr.ensure((size_t) cells * 3 + 1)) != HU_OK || (rc = dd.ensure(rdv.size())) != HU_OK || (rc = vo.ensure(rdv.size())) != HU_OK) return rc;
			std::vector<HuVitOut> hv(rdv.size());
			HIPCHK(hipMemcpyAsync(dd.p, rdv.data(), rdv.size() * sizeof(HuReadDesc), hipMemcpyHostToDevice, b->stream));
			if(vlds <= 96 * 1024 && !kb.viterbi_hbm) { /* a workgroup per sequence, the wavefront staged in LDS, every value filed for the traceback: K + len steps of a barrier
			                                           * each instead of as many round trips to HBM by one wave (30 - 100 ms for a handful of 150-base reads) */
				if(vlds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_viterbi_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int) vlds));
				k_viterbi_lds<<<(unsigned) rdv.size(), HU_VIT_THREADS, vlds, b->stream>>>(d, dd.p, b->dBases.p, scr.p, b->dTraces.p, tNN, tNB, tEC, tCC, vo.p, ldsRows, 0);
				k_viterbi_trace<<<((unsigned) rdv.size() + 63) / 64, 64, 0, b->stream>>>(d, dd.p, scr.p, b->dTraces.p, tNN, tNB, vo.p, (int) rdv.size());
			}
			else k_viterbi<<<(unsigned) rdv.size(), 64, 0, b->stream>>>(d, dd.p, b->dBases.p, scr.p, b->dTraces.p, tNN, tNB, tEC, tCC, vo.p);
			HIPCHK(hipGetLastError());
			HIPCHK(hipMemcpyAsync(hv.data(), vo.p, hv.size() * sizeof(HuVitOut), hipMemcpyDeviceToHost, b->stream));
			HIPCHK(hu_wait(b->stream));
			for(size_t k = 0; k < rdv.size(); ++k) {
				const int s = redo[at + k];
				b->hVit[s] = hv[k];                     /* the trace is already written at the sequence's traceOff */
				if(b->hVit[s].status == HU_READ_NEEDS_FULL) b->hVit[s].status = HU_READ_INVALID;
				HIPCHK(hipMemcpyAsync(b->dVit.p + s, &b->hVit[s], sizeof(HuVitOut), hipMemcpyHostToDevice, b->stream));
				b->hDescs[s].nRegions = -1; /* mark: full DP was used */
			}
			HIPCHK(hu_wait(b->stream));
			at = e;
		}
		for(int s = 0; s < b->nSeq; ++s) if(b->hVit[s].status == HU_READ_NEEDS_FULL) {
			b->hVit[s].status = HU_READ_INVALID;
			HIPCHK(hipMemcpyAsync(b->dVit.p + s, &b->hVit[s], sizeof(HuVitOut), hipMemcpyHostToDevice, b->stream));
		}
		{
			Timer t(b, HU_T_ALIGN_BUILD);
			k_align_rows<<<b->nSeq, 64, 0, b->stream>>>(d, b->dDescs.p, b->dBases.p, b->dTraces.p, b->dVit.p, b->dRows.p, b->dAlns.p);
			if(b->paired) k_merge_rows<<<b->n, 256, 0, b->stream>>>(d, b->n, o->ignore_orient, b->dRows.p, b->dAlns.p);
			const int tiles = (b->n + HU_READ_TILE - 1) / HU_READ_TILE;
			HIPCHK(hipMemsetAsync(b->dRp.p, 0, (size_t) tiles * d.WQ * HU_READ_TILE * 16 * 4, b->stream));
			if((rc = tile_reads(b, b->dAlns.p)) != HU_OK) return rc;
			k_encode_rows<<<b->n, 64, 0, b->stream>>>(d, b->dRows.p, b->dAlns.p, b->dCodes.p, b->dStart.p, b->dEnd.p, b->dRp.p, b->dRq.p, b->dIns.p, b->dReadSlot.p, (uint2*) b->dRSpan.p);
			k_tile_lists<<<tiles, 64, 0, b->stream>>>(d, b->n, b->dRq.p, b->dIns.p, b->dTileQ.p, b->dTileIns.p, b->dSlotRead.p, (const uint2*) b->dRSpan.p, (uint2*) b->dTileSpan.p);
		}
		HIPCHK(hipGetLastError());
		b->hAlns.resize(b->nSeq);
		b->hStart.resize(b->n); b->hEnd.resize(b->n);
		HIPCHK(hipMemcpyAsync(b->hAlns.data(), b->dAlns.p, (size_t) b->nSeq * sizeof(HuAlnDev), hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hipMemcpyAsync(b->hStart.data(), b->dStart.p, (size_t) b->n * 4, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hipMemcpyAsync(b->hEnd.data(), b->dEnd.p, (size_t) b->n * 4, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hu_wait(b->stream));
		for(int s = 0; s < b->nSeq; ++s) b->hAlns[s].usedFull |= (b->hDescs[s].nRegions == -1 || (b->hDescs[s].nRegions == 1 && !b->hDescs[s].reg[0].band)) ? 1 : 0;
		/* reads whose region leaves the resident message window carry HU_READ_OUT_OF_WINDOW and an empty region (k_encode_rows) */
	}
	b->state = ST_ALIGNED;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_align_batch"); }

__global__ void k_seed_drop_empty(int n, const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, int32_t* __restrict__ seedCnt);
static inline HuReadPlanes read_planes(const hu_batch* b) { return HuReadPlanes{b->dRp.p, b->dRq.p, b->dIns.p, b->dReadSlot.p, (const uint2*) b->dRSpan.p}; }

/* HU_SEED_ORDER_LIBSTDCXX: getSeed's std::sort + the caller's truncation as the reference runs them (src/HmmUFOtu_main.cpp:139,
 * src/hmmufotu.cpp:646-647), on the host, from the pair matrix the scan has just left on the device.  The rows come over in chunks through
 * two page-locked buffers (the copy of one chunk runs under the host work on the previous one); per read the (d, N) of every eligible node
 * become order-isomorphic integer keys in node order, and hu_sort_prefix_packed leaves in the first max_nseed places what libstdc++'s
 * introsort would.  A read with a NaN distance (a node sharing no base with it: 0 / 0) takes the (dist, node id) order with NaN last —
 * std::sort is undefined there and the oracle falls back the same way.  Seeds, their (d, N) and their parents' go back to the device. */

extern "C" int hu_sort_desc_device(int device, const double* keys, int rows, int n, int32_t* order) try {
	if(!keys || !order || rows < 1 || n < 0 || n > HU_MAX_SEEDS) { hu_set_error("hu_sort_desc_device: bad argument"); return HU_ERR_ARG; }
	if(hu_device_count() <= 0) { hu_set_error("no gfx950 device visible: the engine has no CPU path"); return HU_ERR_DEVICE; }
	HIPCHK(hipSetDevice(device));
	if(n == 0) return HU_OK;
	DBuf<double> dk; DBuf<int32_t> dor;
	int rc;
	if((rc = dk.ensure((size_t) rows * n)) != HU_OK || (rc = dor.ensure((size_t) rows * n)) != HU_OK) return rc;
	HIPCHK(hipMemcpy(dk.p, keys, (size_t) rows * n * 8, hipMemcpyHostToDevice));
	(void) hipGetLastError();
	k_sort_desc_test<<<(rows + 63) / 64, 64>>>(rows, n, dk.p, dor.p);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpy(order, dor.p, (size_t) rows * n * 4, hipMemcpyDeviceToHost));
	return HU_OK;
} catch(...) { return hu_catch_all("hu_sort_desc_device"); }

extern "C" int hu_sort_prefix_device_at(int device, const uint32_t* pairs, int rows, int64_t n, int k, int pair16, int64_t root_place, int32_t* out_idx, int32_t* out_cnt);
extern "C" int hu_sort_prefix_device(int device, const uint32_t* pairs, int rows, int64_t n, int k, int pair16, int32_t* out_idx, int32_t* out_cnt) {
	return hu_sort_prefix_device_at(device, pairs, rows, n, k, pair16, n, out_idx, out_cnt);
}
extern "C" int hu_sort_prefix_device_at(int device, const uint32_t* pairs, int rows, int64_t n, int k, int pair16, int64_t root_place, int32_t* out_idx, int32_t* out_cnt) try {
	if(root_place < 0 || root_place > n) { hu_set_error("hu_sort_prefix_device_at: the root's place must be in 0 .. n"); return HU_ERR_ARG; }
	if(!pairs || rows < 1 || n < 1 || n >= (1 << 24) - 1 || k < 1 || k > HU_MAX_SEEDS || !out_idx || !out_cnt) { hu_set_error("hu_sort_prefix_device: bad argument"); return HU_ERR_ARG; }
	if(hu_device_count() <= 0) { hu_set_error("no gfx950 device visible: the engine has no CPU path"); return HU_ERR_DEVICE; }
	HIPCHK(hipSetDevice(device));
	/* a tree of n + 1 nodes with node `root_place` as the root: position p of the sort is node p before it, node p + 1 from it on */
	HuDbDev d; memset(&d, 0, sizeof(d));
	d.nNodes = (int32_t) n + 1; d.nNodesPad = (d.nNodes + HU_NODE_PAD - 1) / HU_NODE_PAD * HU_NODE_PAD; d.root = (int32_t) root_place;
	auto nodeOf = [&](int64_t i) -> size_t { return (size_t)(i < root_place ? i : i + 1); };
	if(hu_refsort_lds(d.nNodes) > 150 * 1024) { hu_set_error("hu_sort_prefix_device: %lld elements are more than the kernel takes", (long long) n); return HU_ERR_ARG; }
	for(size_t i = 0, e = (size_t) rows * (size_t) n; i < e; ++i) /* a p-distance: d differing sites of N compared ones (N = 0: NaN, left to the host path) */
		if((pairs[i] >> 16) > (pairs[i] & 0xffffu) && (pairs[i] & 0xffffu)) { hu_set_error("hu_sort_prefix_device: pair %zu has d > N", i); return HU_ERR_ARG; }
	if(pair16) for(size_t i = 0, e = (size_t) rows * (size_t) n; i < e; ++i)      /* the 16-bit form holds d and N in 8 bits each: a larger value would be cut, not sorted */
		if((pairs[i] >> 16) > 255u || (pairs[i] & 0xffffu) > 255u) { hu_set_error("hu_sort_prefix_device: pair %zu (d = %u, N = %u) does not fit the 16-bit form (d, N <= 255)", i, pairs[i] >> 16, pairs[i] & 0xffffu); return HU_ERR_ARG; }
	const size_t np = (size_t) d.nNodesPad;
	DBuf<int32_t> dPar, dSt, dEn, dCnt, dId, dBail; DBuf<uint32_t> dDN, dPN, dP32; DBuf<uint16_t> dP16; DBuf<unsigned long long> scr;
	int rc;
	if((rc = dPar.ensure(np)) != HU_OK || (rc = dSt.ensure(rows)) != HU_OK || (rc = dEn.ensure(rows)) != HU_OK || (rc = dCnt.ensure(rows)) != HU_OK ||
			(rc = dId.ensure((size_t) rows * HU_MAX_SEEDS)) != HU_OK || (rc = dDN.ensure((size_t) rows * HU_MAX_SEEDS)) != HU_OK || (rc = dPN.ensure((size_t) rows * HU_MAX_SEEDS)) != HU_OK ||
			(rc = dBail.ensure((size_t) rows + 2)) != HU_OK) return rc;
	HIPCHK(hipMemset(dPar.p, 0, np * 4)); HIPCHK(hipMemset(dSt.p, 0, (size_t) rows * 4)); HIPCHK(hipMemset(dEn.p, 0, (size_t) rows * 4)); HIPCHK(hipMemset(dBail.p, 0, 8));
	d.parent = dPar.p;
	std::vector<uint32_t> h32; std::vector<uint16_t> h16;
	if(pair16) {
		h16.assign((size_t) rows * np, 0x0001);
		for(int r = 0; r < rows; ++r) for(int64_t i = 0; i < n; ++i) { const uint32_t v = pairs[(size_t) r * n + i]; h16[(size_t) r * np + nodeOf(i)] = (uint16_t)(((v >> 16) << 8) | (v & 0xffu)); }
		if((rc = dP16.ensure(h16.size())) != HU_OK) return rc;
		HIPCHK(hipMemcpy(dP16.p, h16.data(), h16.size() * 2, hipMemcpyHostToDevice));
	}
	else {
		h32.assign((size_t) rows * np, 1);
		for(int r = 0; r < rows; ++r) for(int64_t i = 0; i < n; ++i) h32[(size_t) r * np + nodeOf(i)] = pairs[(size_t) r * n + i];
		if((rc = dP32.ensure(h32.size())) != HU_OK) return rc;
		HIPCHK(hipMemcpy(dP32.p, h32.data(), h32.size() * 4, hipMemcpyHostToDevice));
	}
	const size_t m0 = (size_t) n, rsOff = (m0 + 63) & ~(size_t) 63, cap = hu_refsort_cap(m0), lds = hu_refsort_lds(d.nNodes);
	const int G = std::min(rows, getenv("HU_RS_GRID") ? atoi(getenv("HU_RS_GRID")) : HU_RS_WGS_PER_CU * 256);
	if((rc = scr.ensure((size_t) G * hu_refsort_words(m0, pair16 ? 2 : 4))) != HU_OK) return rc;
	(void) hipGetLastError();
	hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
	HuScope evg([&] { (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); });
	HIPCHK(hipEventRecord(e0, nullptr));
	if(pair16) {
		if(lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_seed_refsort<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		k_seed_refsort<uint16_t><<<G, HU_RS_THREADS, lds>>>(d, dP16.p, rows, dSt.p, dEn.p, k, scr.p, hu_refsort_words(m0, 2), cap, (int) rsOff, (int) hu_refsort_tabcap(m0), dCnt.p, dId.p, dDN.p, dPN.p, dBail.p);
	}
	else {
		if(lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_seed_refsort<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		k_seed_refsort<uint32_t><<<G, HU_RS_THREADS, lds>>>(d, dP32.p, rows, dSt.p, dEn.p, k, scr.p, hu_refsort_words(m0, 4), cap, (int) rsOff, (int) hu_refsort_tabcap(m0), dCnt.p, dId.p, dDN.p, dPN.p, dBail.p);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(e1, nullptr));
	HIPCHK(hipDeviceSynchronize());
	if(getenv("HU_RS_TIMING")) { float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1); fprintf(stderr, "[hu] k_seed_refsort: %d rows x %lld elements, grid %d: %.3f ms\n", rows, (long long) n, G, ms);
#ifdef HU_RS_PROF
		unsigned long long pr[176]; (void) hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_rs_prof), sizeof pr);
		static const char* nm[10] = {"idle", "pivot", "passA", "scan", "cut", "B1", "B2/tiny", "fin-load", "finisher", "trace-back"};
		for(int i = 0; i < 10; ++i) { fprintf(stderr, "[hu]   %-12s %12llu ticks |", nm[i], pr[i]); for(int l = 0; l < 10; ++l) fprintf(stderr, " %9llu", pr[16 + 10 * l + i] / 1000); fprintf(stderr, "\n"); }
		fprintf(stderr, "[hu]   level 0 sums: cut %llu, j_m %llu, m %llu; range sizes of levels 0 / 1 / 2: %llu %llu %llu\n", pr[10], pr[11], pr[12], pr[13], pr[14], pr[15]);
		unsigned long long z[176] = {0}; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_rs_prof), z, sizeof z);
#endif
	}
	std::vector<int32_t> cnt(rows), ids((size_t) rows * HU_MAX_SEEDS), hb((size_t) rows + 2);
	HIPCHK(hipMemcpy(cnt.data(), dCnt.p, (size_t) rows * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(ids.data(), dId.p, ids.size() * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(hb.data(), dBail.p, hb.size() * 4, hipMemcpyDeviceToHost));
	for(int r = 0; r < rows; ++r) { out_cnt[r] = cnt[r]; for(int s = 0; s < k; ++s) { const int32_t nd = ids[(size_t) r * HU_MAX_SEEDS + s]; out_idx[(size_t) r * k + s] = s < cnt[r] ? (nd > root_place ? nd - 1 : nd) : -1; } }
	for(int i = 0; i < hb[0]; ++i) {
		out_cnt[hb[2 + i] & 0x3ffffff] = -1;
		if(getenv("HU_RS_TIMING")) fprintf(stderr, "[hu]   row %d left to the host: reason %d (1 depth, 2 tables, 3 NaN, 4 no stopper, 5 stash, 6 finisher)\n", hb[2 + i] & 0x3ffffff, hb[2 + i] >> 26);
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_sort_prefix_device"); }

/* the reads the device sort listed for a NaN distance (entry = read | 3 << 26) take their seeds from the (dist, node id) selection with NaN last —
 * what the reference-order rule falls back to where std::sort is undefined — computed for the whole batch by k_seed_topk into spare arrays */
__global__ __launch_bounds__(64) void k_take_nan_rows(HuDbDev db, const int32_t* __restrict__ bail, const void* __restrict__ pairs, int p16,
		const int32_t* __restrict__ tCnt, const int32_t* __restrict__ tId, const uint32_t* __restrict__ tDN,
		int32_t* __restrict__ seedCnt, int32_t* __restrict__ seedId, uint32_t* __restrict__ seedDN, uint32_t* __restrict__ parDN) {
	if((int) blockIdx.x >= bail[0]) return;
	const int ent = bail[2 + blockIdx.x];
	if((ent >> 26) != 3) return;
	const int r = ent & 0x3ffffff, s = threadIdx.x;
	const int cnt = tCnt[r];
	if(s == 0) seedCnt[r] = cnt;
	if(s < cnt) {
		const int node = tId[(size_t) r * HU_MAX_SEEDS + s];
		seedId[(size_t) r * HU_MAX_SEEDS + s] = node;
		seedDN[(size_t) r * HU_MAX_SEEDS + s] = tDN[(size_t) r * HU_MAX_SEEDS + s];
		parDN[(size_t) r * HU_MAX_SEEDS + s] = hu_pair_load(pairs, (size_t) r * db.nNodesPad + db.parent[node], p16);
	}
}

/* the reference-order mode under a height filter (-H): std::sort runs over the nodes that pass it, in node order — the pair rows compacted to
 * those nodes (k_compact_rows), sorted as the rows of a tree whose root stands behind the last of them, the places mapped back (k_map_seeds) */
template<class PT>
__global__ __launch_bounds__(256) void k_compact_rows(const PT* __restrict__ pairs, size_t np, const int32_t* __restrict__ hv, int m, PT* __restrict__ out, size_t npC, int nRows) {
	const int p = blockIdx.x * 256 + threadIdx.x;
	if(p >= (int) npC) return;
	const int src = p < m ? hv[p] : 0;
	for(int row = blockIdx.y; row < nRows; row += gridDim.y)       /* gridDim.y ends at 65,535: a larger batch walks its rows */
		out[(size_t) row * npC + p] = p < m ? pairs[(size_t) row * np + src] : HuPair<PT>::pack(1u);
}
__global__ __launch_bounds__(64) void k_map_seeds(HuDbDev db, const int32_t* __restrict__ hv, const void* __restrict__ pairs, int p16,
		const int32_t* __restrict__ seedCnt, int32_t* __restrict__ seedId, uint32_t* __restrict__ seedDN, uint32_t* __restrict__ parDN) {
	const int r = blockIdx.x, s = threadIdx.x;
	if(s >= seedCnt[r]) return;
	const int node = hv[seedId[(size_t) r * HU_MAX_SEEDS + s]];
	seedId[(size_t) r * HU_MAX_SEEDS + s] = node;
	seedDN[(size_t) r * HU_MAX_SEEDS + s] = hu_pair_load(pairs, (size_t) r * db.nNodesPad + node, p16);
	parDN[(size_t) r * HU_MAX_SEEDS + s] = hu_pair_load(pairs, (size_t) r * db.nNodesPad + db.parent[node], p16);
}

static int seed_order_libstdcxx(hu_batch* b, const hu_opts* o, const std::vector<int32_t>* only = nullptr);

/* does the scan of this batch prepare level 0 of the device sort (k_ref_pivots + the masks of k_seed_pdist2<PT, true>)?  Whenever k_seed_refsort will run on the
 * pair rows as they are (no height filter: its rows are compacted first) and the row is a streaming level (more places than the kernel holds in LDS) */
static bool ref_fused_l0(const hu_batch* b, const hu_opts* o, bool pair16) {
	const HuDbDev& d = b->db->dev;
	if(b->knob.refsort_host || b->knob.ref_nofuse || d.nNodes < 3 || o->max_height != INFINITY) return false;
	const size_t m0 = (size_t) d.nNodes - 1;
	if(hu_refsort_lds(d.nNodes) > 150 * 1024 || m0 >= ((size_t) 1 << 24)) return false;
	return (int) m0 > (pair16 ? hu_refsort_lcap<uint16_t>(m0) : hu_refsort_lcap<uint32_t>(m0));
}

/* the same on the device (k_seed_refsort: data-parallel Hoare partitions, hu_kern_refsort.h); the reads it lists — a NaN distance, the
 * heap-sort branch of introsort — are finished by the host function */
static int seed_order_libstdcxx_device(hu_batch* b, const hu_opts* o) {
	const HuDbDev& d0 = b->db->dev;
	const int n = b->n;
	if(b->knob.refsort_host || d0.nNodes < 3) return seed_order_libstdcxx(b, o);
	int rc;
	/* a height filter: the rows compacted to the nodes that pass it, sorted as a tree of those nodes + a root behind them */
	const bool filtered = o->max_height != INFINITY;
	HuDbDev d = d0;
	const void* pairsIn = b->dPairs.p;
	if(filtered) {
		if(b->refHvHeight != o->max_height || !b->dHv.p) {
			std::vector<int32_t> hv;
			for(int i = 0; i < d0.nNodes; ++i) if(i != d0.root && b->db->height[i] <= o->max_height) hv.push_back(i);
			if((rc = b->dHv.ensure(std::max<size_t>(hv.size(), 1))) != HU_OK) return rc;
			if(!hv.empty()) HIPCHK(hipMemcpyAsync(b->dHv.p, hv.data(), hv.size() * 4, hipMemcpyHostToDevice, b->stream));
			HIPCHK(hu_wait(b->stream));      /* hv is a local */
			b->refHvCount = (int) hv.size(); b->refHvHeight = o->max_height;
		}
		const int m = b->refHvCount;
		if(m < 2 || n < 1) return seed_order_libstdcxx(b, o);
		const size_t npC = ((size_t) m + 1 + HU_NODE_PAD - 1) / HU_NODE_PAD * HU_NODE_PAD, pb = b->pair16 ? 2 : 4;
		if((rc = b->dPairsC.ensure(((size_t) n * npC * pb + 7) / 8)) != HU_OK || (rc = b->dZeroPar.ensure(npC)) != HU_OK) return rc;
		HIPCHK(hipMemsetAsync(b->dZeroPar.p, 0, npC * 4, b->stream));
		const dim3 gc((unsigned)((npC + 255) / 256), (unsigned) std::min(n, 65535));
		if(b->pair16) k_compact_rows<uint16_t><<<gc, 256, 0, b->stream>>>((const uint16_t*) b->dPairs.p, (size_t) d0.nNodesPad, b->dHv.p, m, (uint16_t*) b->dPairsC.p, npC, n);
		else k_compact_rows<uint32_t><<<gc, 256, 0, b->stream>>>((const uint32_t*) b->dPairs.p, (size_t) d0.nNodesPad, b->dHv.p, m, (uint32_t*) b->dPairsC.p, npC, n);
		HIPCHK(hipGetLastError());
		d.nNodes = m + 1; d.nNodesPad = (int32_t) npC; d.root = m; d.parent = b->dZeroPar.p;      /* (the kernel's own parent pairs are overwritten by k_map_seeds) */
		pairsIn = b->dPairsC.p;
	}
	const size_t m0 = (size_t) d.nNodes - 1;
	const size_t rsOff = (m0 + 63) & ~(size_t) 63, cap = hu_refsort_cap(m0);
	const size_t lds = hu_refsort_lds(d.nNodes);
	if(lds > 150 * 1024 || m0 >= ((size_t) 1 << 24)) {      /* the kernel's LDS index of the level tables (every 64th entry) does not fit (above ~9 M nodes; until round 4 the whole level-0 tables had to: ~600 k), or a place does not fit the 24 bits of the trace-back's tag */
		static std::atomic<bool> said{false};
		if(!said.exchange(true)) fprintf(stderr, "[hu] reference seed order: a tree of %d nodes is beyond the device sort (k_seed_refsort: %zu KB of LDS tables, limit 150): every read takes the "
				"host restatement of std::sort — tens of times slower (DESIGN.md section 4); --seed-order stable runs on the device at any size\n", d.nNodes, lds / 1024);
		b->nRefBail = n; b->refHostAll = 1;
		return seed_order_libstdcxx(b, o);
	}
	b->refHostAll = 0;
	int G = std::min(n, std::max(1, std::min(b->knob.refsort_wgs, 8 * 256)));        /* HU_RS_WGS_PER_CU workgroups per CU (launch bounds, hu_kern_refsort.h); reads are handed out through a counter */
	const size_t perWg = hu_refsort_words(m0, b->pair16 ? 2 : 4);
	{ const size_t budget = (size_t) 5 << 30; G = (int) std::max<size_t>(1, std::min<size_t>((size_t) G, budget / (perWg * 8))); }
	if((rc = b->dRefScratch.ensure((size_t) G * perWg)) != HU_OK || (rc = b->dBail.ensure((size_t) n + 2)) != HU_OK) return rc;
	HIPCHK(hipMemsetAsync(b->dBail.p, 0, 8, b->stream));
	hipEvent_t e0 = nullptr, e1 = nullptr;
	HuScope evg([&] { if(e0) (void) hipEventDestroy(e0); if(e1) (void) hipEventDestroy(e1); });
	if(b->knob.trace) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, b->stream)); }
	const bool fused = b->refFused && !filtered;      /* the scan of this batch left the pivots and stopper masks of level 0 */
	const uint32_t* l0piv = fused ? b->dRefPiv.p : nullptr; const unsigned long long* l0m = fused ? b->dRefL0.p : nullptr;
	if(b->pair16) {
		if(lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_seed_refsort<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		k_seed_refsort<uint16_t><<<G, HU_RS_THREADS, lds, b->stream>>>(d, (const uint16_t*) pairsIn, n, b->dStart.p, b->dEnd.p, o->max_nseed,
				b->dRefScratch.p, perWg, cap, (int) rsOff, (int) hu_refsort_tabcap(m0), b->dSeedCnt.p, b->dSeedId.p, b->dSeedDN.p, b->dParDN.p, b->dBail.p, l0piv, l0m);
	}
	else {
		if(lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_seed_refsort<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		k_seed_refsort<uint32_t><<<G, HU_RS_THREADS, lds, b->stream>>>(d, (const uint32_t*) pairsIn, n, b->dStart.p, b->dEnd.p, o->max_nseed,
				b->dRefScratch.p, perWg, cap, (int) rsOff, (int) hu_refsort_tabcap(m0), b->dSeedCnt.p, b->dSeedId.p, b->dSeedDN.p, b->dParDN.p, b->dBail.p, l0piv, l0m);
	}
	HIPCHK(hipGetLastError());
	if(filtered) { k_map_seeds<<<n, 64, 0, b->stream>>>(d0, b->dHv.p, b->dPairs.p, b->pair16 ? 1 : 0, b->dSeedCnt.p, b->dSeedId.p, b->dSeedDN.p, b->dParDN.p); HIPCHK(hipGetLastError()); }
	if(e1) HIPCHK(hipEventRecord(e1, b->stream));
	b->hBail.resize((size_t) n + 2);
	PinnedVec<int32_t>& hb = b->hBail;
	HIPCHK(hipMemcpyAsync(hb.data(), b->dBail.p, ((size_t) n + 2) * 4, hipMemcpyDeviceToHost, b->stream));
	HIPCHK(hu_wait(b->stream));
	b->nRefBail = hb[0];
	if(e1) { float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1); fprintf(stderr, "[hu] k_seed_refsort: %d reads, grid %d, %s pairs%s: %.3f ms, %d reads left to the host\n", n, G, b->pair16 ? "16-bit" : "32-bit", fused ? ", level 0 from the scan" : "", ms, hb[0]);
#ifdef HU_RS_PROF
		unsigned long long pr[176]; (void) hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_rs_prof), sizeof pr);
		static const char* nm[10] = {"idle", "pivot", "passA", "scan", "cut", "B1", "B2/tiny", "fin-load", "finisher", "trace-back"};
		for(int i = 0; i < 10; ++i) { fprintf(stderr, "[hu]   %-12s %12llu ticks |", nm[i], pr[i]); for(int l = 0; l < 10; ++l) fprintf(stderr, " %9llu", pr[16 + 10 * l + i] / 1000); fprintf(stderr, "\n"); }
		fprintf(stderr, "[hu]   level 0 sums: cut %llu, j_m %llu, m %llu; range sizes of levels 0 / 1 / 2: %llu %llu %llu\n", pr[10], pr[11], pr[12], pr[13], pr[14], pr[15]);
		unsigned long long z[176] = {0}; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_rs_prof), z, sizeof z);
#endif
	}
	if(hb[0] > 0) {
		std::vector<int32_t> only;
		int nNan = 0;
		for(int i = 0; i < hb[0]; ++i) nNan += (hb[2 + i] >> 26) == 3;
		const bool nanOnDevice = nNan > 16;       /* a database with partial sequences: nearly every read meets a node it shares no column with */
		if(nanOnDevice) {
			if((rc = b->dNanCnt.ensure((size_t) n)) != HU_OK || (rc = b->dNanId.ensure((size_t) n * HU_MAX_SEEDS)) != HU_OK || (rc = b->dNanDN.ensure((size_t) n * HU_MAX_SEEDS)) != HU_OK) return rc;
			if(b->pair16) k_seed_topk<uint16_t><<<n, 256, 0, b->stream>>>(d0, (const uint16_t*) b->dPairs.p, o->max_height, o->max_nseed, b->dNanCnt.p, b->dNanId.p, b->dNanDN.p, b->knob.topk_fast_min);
			else k_seed_topk<uint32_t><<<n, 256, 0, b->stream>>>(d0, b->dPairs.p, o->max_height, o->max_nseed, b->dNanCnt.p, b->dNanId.p, b->dNanDN.p, b->knob.topk_fast_min);
			k_take_nan_rows<<<hb[0], 64, 0, b->stream>>>(d0, b->dBail.p, b->dPairs.p, b->pair16 ? 1 : 0, b->dNanCnt.p, b->dNanId.p, b->dNanDN.p, b->dSeedCnt.p, b->dSeedId.p, b->dSeedDN.p, b->dParDN.p);
			HIPCHK(hipGetLastError());
			if(b->knob.trace) fprintf(stderr, "[hu] reference seed order: %d reads with a NaN distance take the (dist, node id) selection on the device\n", nNan);
		}
		for(int i = 0; i < hb[0]; ++i) if(!(nanOnDevice && (hb[2 + i] >> 26) == 3)) only.push_back(hb[2 + i] & 0x3ffffff);
		std::sort(only.begin(), only.end());
		if(!only.empty()) return seed_order_libstdcxx(b, o, &only);
	}
	return HU_OK;
}

static int seed_order_libstdcxx(hu_batch* b, const hu_opts* o, const std::vector<int32_t>* only) {
	const hu_db* db = b->db;
	const HuDbDev& d = db->dev;
	const size_t n = (size_t) b->n, np = (size_t) d.nNodesPad, K = (size_t) o->max_nseed;
	if(only) { /* a few reads the device left over: their rows one by one */
		const size_t rowB = np * (b->pair16 ? 2 : 4);
		std::vector<uint8_t> rowBuf(rowB);
		std::vector<uint64_t> a;
		const bool p16 = b->pair16;
		for(int32_t r : *only) {
			HIPCHK(hipMemcpyAsync(rowBuf.data(), (const uint8_t*) b->dPairs.p + (size_t) r * rowB, rowB, hipMemcpyDeviceToHost, b->stream));
			HIPCHK(hu_wait(b->stream));
			const uint16_t* q16 = (const uint16_t*) rowBuf.data(); const uint32_t* q32 = (const uint32_t*) rowBuf.data();
			auto pairOf = [&](int i) -> uint32_t { return p16 ? (((uint32_t)(q16[i] >> 8) << 16) | (q16[i] & 0xffu)) : q32[i]; };
			a.clear();
			bool nan = false;
			for(int i = 0; i < d.nNodes; ++i) {
				if(i == d.root || !(db->height[i] <= o->max_height)) continue;
				const uint32_t pr = pairOf(i); const uint64_t dd = pr >> 16, N = pr & 0xffffu;
				if(N == 0) { nan = true; a.push_back((((uint64_t) 1 << 39) + 1) << 24 | (uint64_t) i); continue; }
				a.push_back(((dd << 39) / N) << 24 | (uint64_t) i);
			}
			const size_t keep = std::min(K, a.size());
			if(!nan) hu_sort_prefix_packed(a.data(), a.size(), K); else std::partial_sort(a.begin(), a.begin() + keep, a.end());
			int32_t cnt = (b->hAlns[r].status == HU_READ_OK && b->hEnd[r] >= b->hStart[r]) ? (int32_t) keep : 0;
			int32_t ids[HU_MAX_SEEDS]; uint32_t dn[HU_MAX_SEEDS], pn[HU_MAX_SEEDS];
			for(size_t s = 0; s < keep; ++s) { const int id = (int)(a[s] & 0xffffffu); ids[s] = id; dn[s] = pairOf(id); pn[s] = pairOf(db->parent[id]); }
			HIPCHK(hipMemcpyAsync(b->dSeedCnt.p + r, &cnt, 4, hipMemcpyHostToDevice, b->stream));
			HIPCHK(hipMemcpyAsync(b->dSeedId.p + (size_t) r * HU_MAX_SEEDS, ids, keep * 4, hipMemcpyHostToDevice, b->stream));
			HIPCHK(hipMemcpyAsync(b->dSeedDN.p + (size_t) r * HU_MAX_SEEDS, dn, keep * 4, hipMemcpyHostToDevice, b->stream));
			HIPCHK(hipMemcpyAsync(b->dParDN.p + (size_t) r * HU_MAX_SEEDS, pn, keep * 4, hipMemcpyHostToDevice, b->stream));
			HIPCHK(hu_wait(b->stream));      /* the staging arrays are locals */
		}
		return HU_OK;
	}
	const size_t rowBytes = np * (b->pair16 ? 2 : 4);
	const size_t CH = std::max<size_t>(1, std::min<size_t>(n, (192u << 20) / rowBytes));
	b->hSeedCnt.assign(n, 0); b->hSeedId.assign(n * HU_MAX_SEEDS, 0); b->hSeedDN.assign(n * HU_MAX_SEEDS, 0);
	PinnedVec<uint32_t> hPar(n * HU_MAX_SEEDS, 0);
	PinnedVec<uint8_t> buf[2];
	buf[0].resize(CH * rowBytes); buf[1].resize(CH * rowBytes);
	hipEvent_t ev[2] = {nullptr, nullptr};
	HuScope guard([&] { for(int i = 0; i < 2; ++i) if(ev[i]) (void) hipEventDestroy(ev[i]); });
	for(int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
	auto copy = [&](size_t c) -> int { /* chunk c into buffer c & 1 */
		const size_t r0 = c * CH, cnt = std::min(CH, n - r0);
		HIPCHK(hipMemcpyAsync(buf[c & 1].data(), (const uint8_t*) b->dPairs.p + r0 * rowBytes, cnt * rowBytes, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hipEventRecord(ev[c & 1], b->stream));
		return HU_OK;
	};
	const size_t nChunks = (n + CH - 1) / CH;
	const double maxH = o->max_height;
	const bool p16 = b->pair16;
	const uint64_t nanKey = ((uint64_t) 1 << 39) + 1;
	static const std::vector<uint64_t> key16 = [] { /* [d << 8 | N] -> floor(d 2^39 / N); N = 0 -> beyond every distance */
		std::vector<uint64_t> t(65536);
		for(uint32_t v = 0; v < 65536; ++v) { const uint64_t dd = v >> 8, N = v & 0xffu; t[v] = N ? (dd << 39) / N : ((uint64_t) 1 << 39) + 1; }
		return t;
	}();
	int rc;
	if(nChunks && (rc = copy(0)) != HU_OK) return rc;
	for(size_t c = 0; c < nChunks; ++c) {
		HIPCHK(hipEventSynchronize(ev[c & 1]));
		if(c + 1 < nChunks && (rc = copy(c + 1)) != HU_OK) return rc;
		const size_t r0 = c * CH, cnt = std::min(CH, n - r0);
		const uint8_t* rows = buf[c & 1].data();
		/* ~0.5 ms of host work per read (2 x 10^5 keys, ~4 x 10^5 element visits of the partitions): spread over up to HU_SORT_THREADS (default 32)
		 * threads per batch in flight — the per-batch pool of the <= 50-record stages (8 threads, and serial below 512 items) is too small for it */
		static const unsigned sortThreads = [] { const char* e = getenv("HU_SORT_THREADS"); int v = e ? atoi(e) : 32; unsigned hw = std::thread::hardware_concurrency(); if(hw < 1) hw = 1; return (unsigned) std::min<int>(std::max(v, 1), (int) hw); }();
		std::atomic<size_t> next{0};
		auto oneRead = [&](size_t k) {
			const size_t r = r0 + k;
			if(b->hAlns[r].status != HU_READ_OK || b->hEnd[r] < b->hStart[r]) return;
			static thread_local std::vector<uint64_t> a;
			a.clear();
			const uint16_t* q16 = (const uint16_t*)(rows + k * rowBytes); const uint32_t* q32 = (const uint32_t*)(rows + k * rowBytes);
			auto pairOf = [&](int i) -> uint32_t { return p16 ? (((uint32_t)(q16[i] >> 8) << 16) | (q16[i] & 0xffu)) : q32[i]; };
			bool nan = false;
			for(int i = 0; i < d.nNodes; ++i) {
				if(i == d.root || !(db->height[i] <= maxH)) continue;
				if(p16) { /* d, N <= 255: the key of every (d, N) from a table (a 64-bit division per node costs as much as the partitions) */
					const uint64_t kk = key16[q16[i]];
					nan |= kk == nanKey;
					a.push_back(kk << 24 | (uint64_t) i);
					continue;
				}
				const uint32_t pr = q32[i]; const uint64_t dd = pr >> 16, N = pr & 0xffffu;
				if(N == 0) { nan = true; a.push_back(nanKey << 24 | (uint64_t) i); continue; }    /* beyond every d / N <= 1 */
				a.push_back(((dd << 39) / N) << 24 | (uint64_t) i);
			}
			const size_t keep = std::min(K, a.size());
			if(!nan) hu_sort_prefix_packed(a.data(), a.size(), K);
			else std::partial_sort(a.begin(), a.begin() + keep, a.end());        /* (dist, node id), NaN last */
			b->hSeedCnt[r] = (int32_t) keep;
			for(size_t s = 0; s < keep; ++s) {
				const int id = (int)(a[s] & 0xffffffu);
				b->hSeedId[r * HU_MAX_SEEDS + s] = id;
				b->hSeedDN[r * HU_MAX_SEEDS + s] = pairOf(id);
				hPar[r * HU_MAX_SEEDS + s] = pairOf(db->parent[id]);
			}
		};
		hu_run_threads(std::min<unsigned>(sortThreads, (unsigned) cnt), [&] { for(;;) { const size_t k = next.fetch_add(4); if(k >= cnt) break; for(size_t q = k; q < std::min(cnt, k + 4); ++q) oneRead(q); } });
	}
	if(n) {
		HIPCHK(hipMemcpyAsync(b->dSeedCnt.p, b->hSeedCnt.data(), n * 4, hipMemcpyHostToDevice, b->stream));
		HIPCHK(hipMemcpyAsync(b->dSeedId.p, b->hSeedId.data(), n * HU_MAX_SEEDS * 4, hipMemcpyHostToDevice, b->stream));
		HIPCHK(hipMemcpyAsync(b->dSeedDN.p, b->hSeedDN.data(), n * HU_MAX_SEEDS * 4, hipMemcpyHostToDevice, b->stream));
		HIPCHK(hipMemcpyAsync(b->dParDN.p, hPar.data(), n * HU_MAX_SEEDS * 4, hipMemcpyHostToDevice, b->stream));
		HIPCHK(hu_wait(b->stream));      /* hPar is a local */
	}
	return HU_OK;
}

extern "C" int hu_seed_batch(hu_batch* b, const hu_opts* o) try {
	if(!b || !o) return HU_ERR_ARG;
	if(b->state < ST_ALIGNED) { hu_set_error("hu_seed_batch: reads are not aligned"); return HU_ERR_STATE; }
	if(o->max_nseed < 1 || o->max_nseed > HU_MAX_SEEDS) { hu_set_error("max_nseed must be in 1..%d", HU_MAX_SEEDS); return HU_ERR_ARG; }
	HIPCHK(hipSetDevice(b->db->device));
	const HuDbDev& d = b->db->dev;
	const size_t n = (size_t) b->n;
	int rc;
	/* Large trees without a height filter: the distance-only scan + the top-k that recomputes the (d, N) of its few candidates
	 * (k_seed_dscan, k_seed_topk_d).  Otherwise the full (d, N) pair matrix and k_seed_topk. */
	const int nBlk = d.nNodesPad / 256;
	/* scan_pairs: 1 = the pair matrix always, -1 = the distance-only scan whenever it applies, 0 (default) = the distance-only scan unless
	 * more than 2 % of the reference sequences are partial (measured at gg_97 scale with 40 % of the leaves cut: 264 k reads/s on the
	 * distance-only path, whose top-k then wades through thousands of barely overlapping candidates per read, 378 k on the pair matrix) */
	const bool manyPartial = b->db->partialFrac > 0.02 && b->knob.scan_pairs != -1;
	if(o->seed_order != HU_SEED_ORDER_STABLE && o->seed_order != HU_SEED_ORDER_LIBSTDCXX) { hu_set_error("seed_order must be HU_SEED_ORDER_STABLE or HU_SEED_ORDER_LIBSTDCXX"); return HU_ERR_ARG; }
	const bool refOrder = o->seed_order == HU_SEED_ORDER_LIBSTDCXX;      /* needs every node's (d, N): the pair matrix */
	const bool dOnly = !refOrder && b->knob.scan_pairs != 1 && !manyPartial && o->max_height == INFINITY && nBlk >= 2 * o->max_nseed && nBlk <= 2048 && d.nNodes - 1 >= o->max_nseed;
	const bool narrow = !b->knob.pairs32 && b->maxBases <= 255;      /* 8-bit distances / 16-bit pairs */
	b->pair16 = !dOnly && narrow;
	b->pairsKind = dOnly ? 0 : (b->pair16 ? 16 : 32);
	b->scanWidth = dOnly ? (narrow ? 1 : 2) : 0;
	if((rc = b->dPairs.ensure(std::max<size_t>(n, 1) * d.nNodesPad / (dOnly ? (narrow ? 4 : 2) : (b->pair16 ? 2 : 1)))) != HU_OK) return rc;
	if((rc = b->dSeedCnt.ensure(n)) != HU_OK) return rc;
	if((rc = b->dSeedId.ensure(n * HU_MAX_SEEDS)) != HU_OK) return rc;
	if((rc = b->dSeedDN.ensure(n * HU_MAX_SEEDS)) != HU_OK) return rc;
	if((rc = b->dParDN.ensure(n * HU_MAX_SEEDS)) != HU_OK) return rc;
	uint32_t* bmin = nullptr;
	if(dOnly) {
		if((rc = b->dBmin.ensure(std::max<size_t>(n, 1) * nBlk + 16)) != HU_OK) return rc;
		if((rc = b->dRetry.ensure(n + 1)) != HU_OK) return rc;
		bmin = b->dBmin.p;
	}
	b->refFused = refOrder && n > 0 && ref_fused_l0(b, o, b->pair16);
	if(b->refFused) { /* level 0 of the device sort rides on the scan: the pivots first (four pairs per read straight from the planes) */
		if((rc = b->dRefPiv.ensure(n * 4)) != HU_OK || (rc = b->dRefL0.ensure(n * (size_t)(d.nNodesPad / 64) * 2)) != HU_OK) return rc;
		k_ref_pivots<<<(unsigned)((n + 63) / 64), 64, 0, b->stream>>>(d, read_planes(b), b->n, b->dRefPiv.p);
		HIPCHK(hipGetLastError());
	}
	uint32_t* stat = bmin && b->knob.trace ? bmin + n * nBlk : nullptr;      /* block path: reads served, blocks read, candidates, reads passed on */
	if(stat) HIPCHK(hipMemsetAsync(stat, 0, 64, b->stream));
	(void) hipGetLastError();
	if(n) {
		const int tiles = (b->n + HU_READ_TILE - 1) / HU_READ_TILE;
		const dim3 grid(tiles, d.nNodesPad / 256);
		{
			Timer t(b, HU_T_SEED_PDIST);
			const dim3 grid4(tiles, (d.nNodesPad + 1023) / 1024);
			if(dOnly && narrow) k_seed_dscan4<uint8_t><<<grid4, 256, (size_t)(b->knob.scan_lds_pad > 0 && b->knob.scan_lds_pad <= 44 ? b->knob.scan_lds_pad : 0) * 1024, b->stream>>>(d, b->dRp.p, b->dTileQ.p, (uint8_t*) b->dPairs.p, b->dSlotRead.p, bmin, (const uint2*) b->dTileSpan.p);
			else if(dOnly) k_seed_dscan4<uint16_t><<<grid4, 256, 0, b->stream>>>(d, b->dRp.p, b->dTileQ.p, (uint16_t*) b->dPairs.p, b->dSlotRead.p, bmin, (const uint2*) b->dTileSpan.p);
			else if(b->refFused && b->pair16) k_seed_pdist2<uint16_t, true><<<grid, 256, 0, b->stream>>>(d, b->dRp.p, b->dTileQ.p, b->dTileIns.p, (uint16_t*) b->dPairs.p, b->dSlotRead.p, b->dRefPiv.p, b->dRefL0.p);
			else if(b->refFused) k_seed_pdist2<uint32_t, true><<<grid, 256, 0, b->stream>>>(d, b->dRp.p, b->dTileQ.p, b->dTileIns.p, b->dPairs.p, b->dSlotRead.p, b->dRefPiv.p, b->dRefL0.p);
			else if(b->pair16) k_seed_pdist2<uint16_t><<<grid, 256, 0, b->stream>>>(d, b->dRp.p, b->dTileQ.p, b->dTileIns.p, (uint16_t*) b->dPairs.p, b->dSlotRead.p);
			else k_seed_pdist2<uint32_t><<<grid, 256, 0, b->stream>>>(d, b->dRp.p, b->dTileQ.p, b->dTileIns.p, b->dPairs.p, b->dSlotRead.p);
		}
		{
			Timer t(b, HU_T_SEED_TOPK);
			if(dOnly) { /* the straight path for every read, then the general one for the reads the first launch listed (usually none) */
				int32_t* retry = b->dRetry.p;
				HIPCHK(hipMemsetAsync(retry, 0, 4, b->stream));
				#define TOPK_ARGS(T) d, (const T*) b->dPairs.p, bmin, read_planes(b), o->max_nseed, b->dSeedCnt.p, b->dSeedId.p, b->dSeedDN.p, b->dParDN.p, stat, retry
				if(narrow) { k_seed_topk_straight<uint8_t><<<b->n, 256, 0, b->stream>>>(TOPK_ARGS(uint8_t), b->knob.topk_general); k_seed_topk_d<uint8_t, true><<<std::min(b->n, 1024), 256, 0, b->stream>>>(TOPK_ARGS(uint8_t)); }
				else { k_seed_topk_straight<uint16_t><<<b->n, 256, 0, b->stream>>>(TOPK_ARGS(uint16_t), b->knob.topk_general); k_seed_topk_d<uint16_t, true><<<std::min(b->n, 1024), 256, 0, b->stream>>>(TOPK_ARGS(uint16_t)); }
				#undef TOPK_ARGS
			}
			else if(refOrder) { if((rc = seed_order_libstdcxx_device(b, o)) != HU_OK) return rc; }
			else {
				if(b->pair16) k_seed_topk<uint16_t><<<b->n, 256, 0, b->stream>>>(d, (const uint16_t*) b->dPairs.p, o->max_height, o->max_nseed, b->dSeedCnt.p, b->dSeedId.p, b->dSeedDN.p, b->knob.topk_fast_min);
				else k_seed_topk<uint32_t><<<b->n, 256, 0, b->stream>>>(d, b->dPairs.p, o->max_height, o->max_nseed, b->dSeedCnt.p, b->dSeedId.p, b->dSeedDN.p, b->knob.topk_fast_min);
				k_parent_pairs<<<((unsigned) n * HU_MAX_SEEDS + 255) / 256, 256, 0, b->stream>>>(d, b->n, b->dPairs.p, b->pair16 ? 1 : 0, b->dSeedCnt.p, b->dSeedId.p, b->dParDN.p);
			}
		}
		k_seed_drop_empty<<<(b->n + 255) / 256, 256, 0, b->stream>>>(b->n, b->dStart.p, b->dEnd.p, b->dSeedCnt.p);
		HIPCHK(hipGetLastError());
		if(stat) {
			uint32_t h[16];
			HIPCHK(hipMemcpyAsync(h, stat, 64, hipMemcpyDeviceToHost, b->stream)); HIPCHK(hu_wait(b->stream));
			fprintf(stderr, "[hu] top-k after the distance-only scan: %u of %zu reads on the block path (%.1f blocks, %.1f candidates per read), %u by the exact recomputation; "
				"%u of them through the general launch\n",
				h[0] + h[8], n, h[0] + h[8] ? (double) h[1] / (h[0] + h[8]) : 0.0, h[0] + h[8] ? (double) h[2] / (h[0] + h[8]) : 0.0, h[3], h[8]);
		}
	}
	b->seedCap = o->max_nseed;
	b->state = ST_SEEDED;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_seed_batch"); }

/* A read whose region is empty (not aligned, outside the resident message window, a segment of no columns) has no seeds, whatever the
 * top-k made of its all-zero distance row ("reads without bases take the first nodes by id"): the stages behind form message addresses
 * from (node, region start), and with no seeds none of them touches such a read.  This is the guard at the source for the fault of
 * round 2 (gpurun_out/t_r2e.log: an out-of-window read kept the region (0, -1), its seed list named node 0, and k_estimate_prod loaded
 * up[(0 * winLen + (0 - winStart)) * 4] — 40 columns BEFORE the first message of the window: a GPU memory fault, which ends the process). */
__global__ void k_seed_drop_empty(int n, const int32_t* __restrict__ rstart, const int32_t* __restrict__ rend, int32_t* __restrict__ seedCnt) {
	const int r = blockIdx.x * 256 + threadIdx.x;
	if(r < n && rend[r] < rstart[r]) seedCnt[r] = 0;
}

/* given seed nodes: their (d, N) and their parents' over the current regions, straight from the bit-planes (no scan) */
__global__ void k_seed_given(HuDbDev db, HuReadPlanes R, int n, const int32_t* __restrict__ cnt, const int32_t* __restrict__ ids,
		const int32_t* __restrict__ distIds, int32_t* __restrict__ seedCnt, int32_t* __restrict__ seedId, uint32_t* __restrict__ seedDN, uint32_t* __restrict__ parDN) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if(i >= n * HU_MAX_SEEDS) return;
	const int r = i / HU_MAX_SEEDS, sl = i % HU_MAX_SEEDS;
	if(sl == 0) seedCnt[r] = cnt[r];
	if(sl < cnt[r]) {
		seedId[i] = ids[i];
		seedDN[i] = pair_exact(db, R, r, distIds ? distIds[i] : ids[i]);
		parDN[i] = pair_exact(db, R, r, db.parent[ids[i]]);
	}
}

extern "C" int hu_seed_batch_given(hu_batch* b, const int32_t* n_seeds, const int32_t* ids, const int32_t* dist_ids, int stride) try {
	if(!b || (b->n > 0 && (!n_seeds || !ids)) || stride < 1) { hu_set_error("hu_seed_batch_given: bad argument"); return HU_ERR_ARG; }
	if(b->state < ST_ALIGNED) { hu_set_error("hu_seed_batch_given: reads are not aligned"); return HU_ERR_STATE; }
	HIPCHK(hipSetDevice(b->db->device));
	const HuDbDev& d = b->db->dev;
	const size_t n = (size_t) b->n;
	/* every id names the child end of a branch: a node other than the root (checked here: the kernels index with them) */
	std::vector<int32_t> pk((n * 3) * HU_MAX_SEEDS + n, 0);
	int32_t* pc = pk.data(); int32_t* pi = pc + n; int32_t* pd = pi + n * HU_MAX_SEEDS;
	for(size_t r = 0; r < n; ++r) {
		const int c = n_seeds[r];
		if(c < 0 || c > HU_MAX_SEEDS || c > stride) { hu_set_error("hu_seed_batch_given: read %zu has %d seeds (0..%d)", r, c, std::min(stride, (int) HU_MAX_SEEDS)); return HU_ERR_ARG; }
		pc[r] = c;
		for(int k = 0; k < c; ++k) {
			const int32_t id = ids[r * stride + k], di = dist_ids ? dist_ids[r * stride + k] : id;
			if(id < 0 || id >= d.nNodes || b->db->parent[id] < 0 || di < 0 || di >= d.nNodes) { hu_set_error("hu_seed_batch_given: read %zu seed %d names node %d / %d", r, k, id, di); return HU_ERR_ARG; }
			pi[r * HU_MAX_SEEDS + k] = id; pd[r * HU_MAX_SEEDS + k] = di;
		}
	}
	int rc;
	b->pair16 = false; b->pairsKind = 0; b->scanWidth = 0;    /* no pair matrix: the pairs of the given nodes come straight from the planes */
	if((rc = b->dSeedCnt.ensure(n)) != HU_OK) return rc;
	if((rc = b->dSeedId.ensure(n * HU_MAX_SEEDS)) != HU_OK) return rc;
	if((rc = b->dSeedDN.ensure(n * HU_MAX_SEEDS)) != HU_OK) return rc;
	if((rc = b->dParDN.ensure(n * HU_MAX_SEEDS)) != HU_OK) return rc;
	if((rc = b->dGiven.ensure(pk.size())) != HU_OK) return rc;
	(void) hipGetLastError();
	if(n) {
		HIPCHK(hipMemcpyAsync(b->dGiven.p, pk.data(), pk.size() * 4, hipMemcpyHostToDevice, b->stream));
		{
			Timer t(b, HU_T_SEED_PDIST);
			k_seed_given<<<((unsigned) n * HU_MAX_SEEDS + 255) / 256, 256, 0, b->stream>>>(d, read_planes(b), b->n, b->dGiven.p, b->dGiven.p + n,
					dist_ids ? b->dGiven.p + n + n * HU_MAX_SEEDS : nullptr, b->dSeedCnt.p, b->dSeedId.p, b->dSeedDN.p, b->dParDN.p);
			k_seed_drop_empty<<<(b->n + 255) / 256, 256, 0, b->stream>>>(b->n, b->dStart.p, b->dEnd.p, b->dSeedCnt.p);
		}
		HIPCHK(hipGetLastError());
		HIPCHK(hu_wait(b->stream)); /* pk is a local */
	}
	b->seedCap = 1;
	for(size_t r = 0; r < n; ++r) b->seedCap = std::max(b->seedCap, (int) n_seeds[r]);
	b->state = ST_SEEDED;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_seed_batch_given"); }

/* (read, seed) slots -> sort key = seed node (invalid slots last) */
__global__ void k_seed_sortkeys(int n, const int32_t* __restrict__ seedCnt, const int32_t* __restrict__ seedId, uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if(i >= n * HU_MAX_SEEDS) return;
	key[i] = (i % HU_MAX_SEEDS) < seedCnt[i / HU_MAX_SEEDS] ? (uint32_t) seedId[i] : 0xffffffffu;
	val[i] = (uint32_t) i;
}

/* The estimate / placement kernels hold a read's whole alignment region in registers, so their shape (sites per thread, waves per candidate,
 * workgroups per CU) goes with the WIDEST region of the launch — and with seeds from the index instead of the truth a few reads per batch have
 * seeds that land far apart: regions of 2,000 - 6,000 columns beside 8,000 reads of ~800 (measured on the 1 M-read pool at 150 bases: 1 - 4 such
 * reads in five batches of eight, placement 23 ms instead of 5).  Those reads get a launch of their own: the class boundary above the
 * (W + 1)-th widest region, W <= 64, splits the batch when the widest region lies in a higher class. */
static void plan_width_split(hu_batch* b) {
	b->rMain = 0; b->wideReads.clear();
	if(!b->knob.width_split || b->n < 256 || (int) b->hStart.size() < b->n || (int) b->hEnd.size() < b->n) return;
	std::vector<int> R((size_t) b->n);
	int maxAll = 0;
	for(int r = 0; r < b->n; ++r) { R[r] = std::max(0, b->hEnd[r] - b->hStart[r] + 1); maxAll = std::max(maxAll, R[r]); }
	const int W = std::min(64, b->n / 128);
	std::vector<int> tmp(R);
	std::nth_element(tmp.begin(), tmp.begin() + W, tmp.end(), std::greater<int>());
	const int kth = tmp[W];
	static const int bounds[] = {512, 768, 1024, 1536, 2048, 3072};
	int bound = 0;
	for(int x : bounds) if(kth <= x) { bound = x; break; }
	if(!bound || bound >= maxAll) return;
	b->rMain = bound;
	for(int r = 0; r < b->n; ++r) if(R[r] > bound) b->wideReads.push_back(r);
	if(b->knob.trace) fprintf(stderr, "[hu] width split: %zu of %d reads beyond %d columns (widest %d) take a launch of their own\n", b->wideReads.size(), b->n, bound, maxAll);
}

extern "C" int hu_estimate_batch(hu_batch* b, const hu_opts* o) try {
	if(!b || !o) return HU_ERR_ARG;
	if(b->state < ST_SEEDED) { hu_set_error("hu_estimate_batch: no seeds"); return HU_ERR_STATE; }
	HIPCHK(hipSetDevice(b->db->device));
	const size_t n = (size_t) b->n;
	int rc;
	if((rc = b->dEst.ensure(n * HU_MAX_SEEDS)) != HU_OK) return rc;
	(void) hipGetLastError();
	if(n) {
		Timer t(b, HU_T_ESTIMATE);
		plan_width_split(b);
		int maxAll = 1, maxMain = 1;
		for(int r = 0; r < b->n; ++r) { const int R = b->hEnd[r] - b->hStart[r] + 1; maxAll = std::max(maxAll, R); if(!b->rMain || R <= b->rMain) maxMain = std::max(maxMain, R); }
		const bool stream = b->knob.streaming_sep != 0;
		if(b->rMain) { /* the wide reads' (read, seed) slots: the second launch's list */
			b->hWideOrd.clear();
			for(int r : b->wideReads) for(int sd = 0; sd < HU_MAX_SEEDS; ++sd) b->hWideOrd.push_back((uint32_t) r * HU_MAX_SEEDS + sd);
			if((rc = b->dWideOrd.ensure(b->hWideOrd.size())) != HU_OK) return rc;
			HIPCHK(hipMemcpyAsync(b->dWideOrd.p, b->hWideOrd.data(), b->hWideOrd.size() * 4, hipMemcpyHostToDevice, b->stream));
		}
		for(int pass = 0; pass < (b->rMain ? 2 : 1); ++pass) { /* pass 1: the reads beyond rMain, on the kernel their width asks for */
			const int maxR = pass ? maxAll : maxMain;
			HuDbDev dev = b->db->dev;
			if(b->rMain) { dev.rLo = pass ? b->rMain : -1; dev.rHi = pass ? 0x7fffffff : b->rMain; dev.wideList = pass ? b->dWideOrd.p : nullptr; }
			#define EST_ARGS dev, b->db->mdl, b->dCodes.p, b->dStart.p, b->dEnd.p, b->dParDN.p, b->dSeedCnt.p, b->dSeedId.p, b->dSeedDN.p, o->weighted, b->dEst.p
			const unsigned eg = pass ? (unsigned) b->hWideOrd.size() : (unsigned) b->n * HU_MAX_SEEDS;
			/* launch order of the table-driven kernels: by seed node */
			const uint32_t* order = nullptr;
			if(pass) order = b->dWideOrd.p;
			else if(!b->knob.est_unsorted) {
				if((rc = b->dSortK.ensure((size_t) eg * 2)) != HU_OK || (rc = b->dSortV.ensure((size_t) eg * 2)) != HU_OK) return rc;
				k_seed_sortkeys<<<(eg + 255) / 256, 256, 0, b->stream>>>(b->n, b->dSeedCnt.p, b->dSeedId.p, b->dSortK.p, b->dSortV.p);
				if((rc = order_by_key(b, b->dSortK.p, b->dSortV.p, (size_t) eg, (uint32_t) b->db->dev.nNodes, b->dSortV.p + eg, false)) != HU_OK) return rc;      /* empty slots (key 0xffffffff) last */
				order = b->dSortV.p + eg;
			}
			/* register-resident variant (messages cross HBM once) while a read's region fits 256 x SPT sites, the
			 * two-pass streaming kernel beyond.  Measured at R = 1363: 256 threads 11.6 ms, 512 13.1, 1024 25.4;
			 * streaming 12.3 ms with twice the HBM traffic. */
			const int spt = (maxR + 255) / 256;
			/* sorted launch: the valid slots come first and number at most n x (seeds per read), so only that many workgroups
			 * start; and XCD x (workgroups b = x mod 8) walks the contiguous eighth x of the sorted list, so that the reads sharing
			 * a seed node share an L2 (5.13 -> 4.55 ms; without the trim the empty slots all fall to one XCD: 5.48 ms) */
			const int xm = b->knob.xcd_map;
			const unsigned egl = pass ? eg : order ? std::min<unsigned>(eg, (unsigned) b->n * (unsigned) b->seedCap) : eg;
			const int var = b->knob.est_var;
			if(stream || spt > 12) k_estimate<<<eg, 64, 0, b->stream>>>(EST_ARGS);
			else if(var == 2) { /* the per-site log() form, kept for comparison */
				if(spt <= 2) k_estimate_blk<2, 256><<<eg, 256, 0, b->stream>>>(EST_ARGS);
				else if(spt <= 4) k_estimate_blk<4, 256><<<eg, 256, 0, b->stream>>>(EST_ARGS);
				else if(spt <= 6) k_estimate_blk<6, 256><<<eg, 256, 0, b->stream>>>(EST_ARGS);
				else if(spt <= 8) k_estimate_blk<8, 256><<<eg, 256, 0, b->stream>>>(EST_ARGS);
				else k_estimate_blk<12, 256><<<eg, 256, 0, b->stream>>>(EST_ARGS);
			}
			else if(var == 1 && spt <= 6) k_estimate_prod<12, 2><<<egl, 128, 0, b->stream>>>(EST_ARGS, order, xm);
			else if(var == 3 && spt <= 6) k_estimate_prod<6, 4, 1><<<egl, 256, 0, b->stream>>>(EST_ARGS, order, xm);
			else if(spt <= 2) k_estimate_prod<2, 4><<<egl, 256, 0, b->stream>>>(EST_ARGS, order, xm);
			else if(spt <= 4) k_estimate_prod<4, 4><<<egl, 256, 0, b->stream>>>(EST_ARGS, order, xm);
			/* measured and not kept: <3, 8, 3> (512 threads x 3 sites, 66 VGPRs, three pairs per CU on 24 waves): 5.04 ms against 4.40;
			 * <6, 4, 5> / <6, 4, 6> (five / six workgroups per CU by launch bounds): 96 / 80 VGPRs with 140 / 204 B of scratch, 9.2 / 11.3 ms against 4.4 */
			else if(spt <= 6) k_estimate_prod<6, 4, 4><<<egl, 256, (size_t)(b->knob.est_lds_pad > 0 && b->knob.est_lds_pad <= 60 ? b->knob.est_lds_pad : 0) * 1024, b->stream>>>(EST_ARGS, order, xm);   /* 128 VGPRs: four workgroups per CU (7.4 -> 6.7 ms) */
			else if(spt <= 8) k_estimate_prod<8, 4><<<egl, 256, 0, b->stream>>>(EST_ARGS, order, xm);   /* (512 threads x 4 sites measured slower here: 6.99 against 6.36 ms at R ~ 1,850) */
			else if(var == 4) k_estimate_prod<12, 4><<<egl, 256, 0, b->stream>>>(EST_ARGS, order, xm);    /* est_var = 4: 256 threads x 12 sites, 231 VGPRs, two workgroups of four waves per CU */
			/* regions of 2,049 .. 3,072 columns (merged mate pairs): 512 threads x 6 sites, 120 VGPRs — the same two pairs per CU as with 256 x 12, but sixteen
			 * waves instead of eight work on them and a pair's life is shorter: 5.67 -> 4.92 ms per 4,096 pairs of 2 x 250 bases, 265.6 k -> 276.6 k pairs/s */
			else k_estimate_prod<6, 8, 2><<<egl, 512, 0, b->stream>>>(EST_ARGS, order, xm);
			#undef EST_ARGS
		}
	}
	HIPCHK(hipGetLastError());
	if(n) { /* gap / base site counts per read for the split placement kernel (read back in the filter stage) */
		if((rc = b->dPermCnt.ensure(n * 2)) != HU_OK) return rc;
		k_site_count<<<b->n, 64, 0, b->stream>>>(b->db->dev, b->n, b->dCodes.p, b->dStart.p, b->dEnd.p, b->dPermCnt.p);
		HIPCHK(hipGetLastError());
	}
	b->state = ST_ESTIMATED;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_estimate_batch"); }


/* the host mirror of the device's candidate list (offsets + every candidate's PTPlacement), filled on demand */
static int sync_host_cands(hu_batch* b) {
	if(b->hostCands) return HU_OK;
	const size_t n = (size_t) b->n;
	std::vector<int32_t> off(n + 1, 0);
	b->places.resize(b->nc);
	HIPCHK(hipSetDevice(b->db->device));
	if(n) HIPCHK(hipMemcpyAsync(off.data(), b->dCandOff.p, (n + 1) * 4, hipMemcpyDeviceToHost, b->stream));
	if(b->nc) HIPCHK(hipMemcpyAsync(b->places.data(), b->dPlaces.p, b->nc * sizeof(HuPlaceRec), hipMemcpyDeviceToHost, b->stream));
	HIPCHK(hu_wait(b->stream));
	b->candOffs.assign(off.begin(), off.end());      /* [n + 1], all zero for an empty batch */
	b->hostCands = true;
	return HU_OK;
}
/* offsets, candidate count and the site-count maxima of the batch from the per-read candidate counts on the device */
static int scan_cands(hu_batch* b, bool withPerm) {
	int rc;
	if((rc = b->dCandOff.ensure((size_t) b->n + 1)) != HU_OK || (rc = b->dMeta.ensure(4)) != HU_OK) return rc;
	k_cand_scan<<<1, 1024, 0, b->stream>>>(b->n, b->dCandCnt.p, b->dCandOff.p, withPerm ? b->dPermCnt.p : nullptr, b->dStart.p, b->dEnd.p, b->dMeta.p, b->rMain);
	HIPCHK(hipGetLastError());
	b->hMeta.assign(4, 0);
	int32_t* meta = b->hMeta.data();
	HIPCHK(hipMemcpyAsync(meta, b->dMeta.p, 12, hipMemcpyDeviceToHost, b->stream));
	if(b->rMain) { b->hCandOffAll.resize((size_t) b->n + 1); HIPCHK(hipMemcpyAsync(b->hCandOffAll.data(), b->dCandOff.p, ((size_t) b->n + 1) * 4, hipMemcpyDeviceToHost, b->stream)); }   /* the wide reads' candidates: the placement stage lists them */
	HIPCHK(hu_wait(b->stream));
	b->nc = (size_t) meta[0]; b->maxGapSites = meta[1]; b->maxBaseSites = meta[2];
	return HU_OK;
}

extern "C" int hu_filter_batch(hu_batch* b, const hu_opts* o) try {
	if(!b || !o) return HU_ERR_ARG;
	if(b->state < ST_ESTIMATED) { hu_set_error("hu_filter_batch: no estimates"); return HU_ERR_STATE; }
	if(!(o->max_error >= 0)) { hu_set_error("max_error must be >= 0"); return HU_ERR_ARG; }
	HIPCHK(hipSetDevice(b->db->device));
	const size_t n = (size_t) b->n;
	if(b->knob.inject_fault == 1) parallel_for(std::max<size_t>(n, 600), [&](size_t r) { if(r + 1 == std::max<size_t>(n, 600)) throw std::bad_alloc(); });   /* fault injection: a worker of the host pool throws */
	int rc;
	if((rc = b->dCandCnt.ensure(std::max<size_t>(n, 1))) != HU_OK || (rc = b->dFiltSlot.ensure(std::max<size_t>(n, 1) * HU_MAX_SEEDS)) != HU_OK) return rc;
	(void) hipGetLastError();
	b->nc = 0; b->maxGapSites = b->maxBaseSites = 0;
	if(n) {
		/* filterPlacements (src/HmmUFOtu_main.cpp:162-173) per read on the device: libstdc++'s std::sort(rbegin, rend, compareByLoglik) restated (hu_kern_rank.h).
		 * A read that is not placed has no seeds (k_seed_drop_empty), hence no candidates */
		k_filter<<<(unsigned)((n + 3) / 4), 256, 0, b->stream>>>(b->n, b->dSeedCnt.p, b->dEst.p, o->max_error, b->dCandCnt.p, b->dFiltSlot.p, b->knob.sort_seq);
		HIPCHK(hipGetLastError());
		if((rc = scan_cands(b, true)) != HU_OK) return rc;
		if((rc = b->dCands.ensure(std::max<size_t>(b->nc, 1))) != HU_OK || (rc = b->dPlaces.ensure(std::max<size_t>(b->nc, 1))) != HU_OK) return rc;
		k_build_cands<<<(unsigned)((n * HU_MAX_SEEDS + 255) / 256), 256, 0, b->stream>>>(b->db->dev, b->n, b->dCandCnt.p, b->dCandOff.p, b->dFiltSlot.p, b->dSeedId.p, b->dEst.p, b->dCands.p, b->dPlaces.p);
		HIPCHK(hipGetLastError());
	}
	b->hostCands = false; b->placesGiven = false;
	b->state = ST_FILTERED;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_filter_batch"); }

extern "C" int hu_batch_set_candidates(hu_batch* b, const int64_t* offs, const hu_place_rec* recs, int placed) try {
	if(!b || !offs) { hu_set_error("hu_batch_set_candidates: null argument"); return HU_ERR_ARG; }
	if(b->state < ST_ALIGNED) { hu_set_error("hu_batch_set_candidates: reads are not aligned"); return HU_ERR_STATE; }
	const hu_db* db = b->db;
	const size_t n = (size_t) b->n;
	if(offs[0] != 0) { hu_set_error("hu_batch_set_candidates: offs[0] must be 0"); return HU_ERR_ARG; }
	for(size_t r = 0; r < n; ++r) {
		const int64_t k = offs[r + 1] - offs[r];
		if(k < 0 || k > HU_MAX_SEEDS || (k > 0 && (!recs || b->hAlns[r].status != HU_READ_OK || b->hEnd[r] < b->hStart[r]))) {
			hu_set_error("hu_batch_set_candidates: read %zu: %lld candidates (0..%d, none for a read that is not placed)", r, (long long) k, HU_MAX_SEEDS); return HU_ERR_ARG;
		}
		for(int64_t c = offs[r]; c < offs[r + 1]; ++c) {
			const int32_t u = recs[c].c_node;
			if(u < 0 || u >= db->dev.nNodes || db->parent[u] < 0) { hu_set_error("hu_batch_set_candidates: read %zu names node %d (a branch is named by its child end: a non-root node)", r, u); return HU_ERR_ARG; }
		}
	}
	HIPCHK(hipSetDevice(db->device));
	b->candOffs.assign(offs, offs + n + 1);
	const size_t nc = (size_t) offs[n];
	b->places.resize(nc); b->hCands.resize(nc);
	std::vector<int32_t> hCnt(std::max<size_t>(n, 1), 0);
	for(size_t r = 0; r < n; ++r) hCnt[r] = (int32_t)(offs[r + 1] - offs[r]);
	for(size_t r = 0; r < n; ++r) for(int64_t c = offs[r]; c < offs[r + 1]; ++c) {
		const hu_place_rec& q = recs[c];
		HostPlace p;
		memset(&p, 0, sizeof(p));
		p.seedIdx = (int32_t)(c - offs[r]); p.cNode = q.c_node; p.pNode = db->parent[q.c_node]; p.wuv = db->blen[q.c_node];
		p.ratio = q.ratio; p.wnr = q.wnr; p.estLoglik = q.est_loglik; p.rootLoglik = NAN; p.qPlace = p.qTaxon = NAN;
		if(placed) { p.loglik = q.loglik; p.height = q.height; p.aNode = q.a_node == p.pNode ? p.pNode : p.cNode; p.rootLoglik = q.root_loglik; }
		else { p.loglik = q.est_loglik; p.aNode = p.ratio <= 0.5 ? p.cNode : p.pNode; }
		b->places[c] = p;
		HuCand cd; cd.read = (int32_t) r; cd.node = p.cNode; cd.ratio0 = p.ratio; cd.wnr0 = p.wnr;
		b->hCands[c] = cd;
	}
	{ /* the list goes to the device, where the later stages read it */
		int rc;
		if((rc = b->dCandCnt.ensure(std::max<size_t>(n, 1))) != HU_OK || (rc = b->dCands.ensure(std::max<size_t>(nc, 1))) != HU_OK ||
				(rc = b->dPlaces.ensure(std::max<size_t>(nc, 1))) != HU_OK || (rc = b->dPermCnt.ensure(std::max<size_t>(n, 1) * 2)) != HU_OK) return rc;
		(void) hipGetLastError();
		if(n) {
			HIPCHK(hipMemcpyAsync(b->dCandCnt.p, hCnt.data(), n * 4, hipMemcpyHostToDevice, b->stream));
			if(nc) {
				HIPCHK(hipMemcpyAsync(b->dCands.p, b->hCands.data(), nc * sizeof(HuCand), hipMemcpyHostToDevice, b->stream));
				HIPCHK(hipMemcpyAsync(b->dPlaces.p, b->places.data(), nc * sizeof(HuPlaceRec), hipMemcpyHostToDevice, b->stream));
			}
			/* gap / base site counts of the regions for the split placement kernel (the estimate stage may not have run on this batch) */
			k_site_count<<<b->n, 64, 0, b->stream>>>(db->dev, b->n, b->dCodes.p, b->dStart.p, b->dEnd.p, b->dPermCnt.p);
			HIPCHK(hipGetLastError());
			plan_width_split(b);
			if((rc = scan_cands(b, true)) != HU_OK) return rc;       /* offsets + counts on the device; synchronises: the host arrays above are done with */
		}
		else { b->nc = 0; b->maxGapSites = b->maxBaseSites = 0; }
	}
	b->hostCands = true;                 /* candOffs / places above ARE the list */
	b->placesGiven = placed != 0;
	b->fixedRoot = false;
	b->state = placed ? ST_PLACED : ST_FILTERED;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_set_candidates"); }

__global__ void k_cand_sortkeys(int nc, const HuCand* __restrict__ cands, uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if(i >= nc) return;
	key[i] = (uint32_t) cands[i].node; val[i] = (uint32_t) i;
}

extern "C" int hu_place_batch(hu_batch* b, const hu_opts* o) try {
	if(!b || !o) return HU_ERR_ARG;
	if(b->state < ST_FILTERED) { hu_set_error("hu_place_batch: candidates are not filtered"); return HU_ERR_STATE; }
	HIPCHK(hipSetDevice(b->db->device));
	const size_t nc = b->nc;            /* the candidates are on the device already (hu_filter_batch / hu_batch_set_candidates) */
	int rc;
	if((rc = b->dPlaceOut.ensure(std::max<size_t>(nc, 1))) != HU_OK) return rc;
	(void) hipGetLastError();
	if(nc) {
		Timer t(b, HU_T_PLACE);
		int maxAll = 1, maxMain = 1;
		for(int r = 0; r < b->n; ++r) { const int R = b->hEnd[r] - b->hStart[r] + 1; maxAll = std::max(maxAll, R); if(!b->rMain || R <= b->rMain) maxMain = std::max(maxMain, R); }
		int passes = 1;
		if(b->rMain) { /* the candidates of the wide reads: the second launch's list (plan_width_split) */
			b->hWideCand.clear();
			if(b->hCandOffAll.size() < (size_t) b->n + 1) { hu_set_error("hu_place_batch: candidate offsets of the wide reads are missing"); return HU_ERR_STATE; }
			for(int r : b->wideReads) for(int32_t c = b->hCandOffAll[r]; c < b->hCandOffAll[r + 1]; ++c) b->hWideCand.push_back((uint32_t) c);
			if(!b->hWideCand.empty()) {
				if((rc = b->dWideCand.ensure(b->hWideCand.size())) != HU_OK) return rc;
				HIPCHK(hipMemcpyAsync(b->dWideCand.p, b->hWideCand.data(), b->hWideCand.size() * 4, hipMemcpyHostToDevice, b->stream));
				passes = 2;
			}
		}
		for(int pass = 0; pass < passes; ++pass) { /* pass 1: the candidates of the reads beyond rMain, on the kernel their width asks for */
			const int maxR = pass ? maxAll : maxMain;
			HuDbDev dev = b->db->dev;
			if(b->rMain) { dev.rLo = pass ? b->rMain : -1; dev.rHi = pass ? 0x7fffffff : b->rMain; dev.wideList = pass ? b->dWideCand.p : nullptr; }
			const unsigned grid = pass ? (unsigned) b->hWideCand.size() : (unsigned) nc;
			const int spt2 = (maxR + 127) / 128, spt4 = (maxR + 255) / 256;  /* sites per thread with 2 / 4 waves per candidate */
			const bool stream = b->knob.streaming_sep != 0 || spt4 > 12;
			#define PL_ARGS dev, b->db->mdl, b->dCodes.p, b->dStart.p, b->dEnd.p, b->dCands.p, b->dPlaceOut.p
			if(stream) { /* regions of more than 3,072 columns: one wave per candidate, messages re-streamed per sweep */
				const size_t lds = (size_t)(3 * HU_MAX_DGK * 4 + HU_MAX_DGK * 5 * 4 + maxR) * sizeof(double);
				if(lds > 160 * 1024) { hu_set_error("alignment region of %d columns does not fit the placement kernel's LDS", maxR); return HU_ERR_ARG; }
				if(lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*) k_place, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
				k_place<<<grid, 64, lds, b->stream>>>(PL_ARGS);
			}
			else { /* one workgroup per candidate, messages and per-site ratios register-resident.  Measured on MI355X
			        * (8192 reads, R = 1363, 25.6 candidates per read): 4 waves x 6 sites 11.9 ms, 2 waves x 12 sites with two
			        * workgroups per SIMD pair 9.5 ms (fewer reduction / exchange / loop instructions per candidate) */
				const int var = b->knob.place_var;
				const int xm = b->knob.xcd_map;   /* an eighth of the node-sorted list per XCD, as in the estimate stage */
				const uint32_t* order = nullptr;
				if(pass) order = b->dWideCand.p;
				else if(!b->knob.place_unsorted) { /* launch order: by candidate node */
					if((rc = b->dSortK.ensure(nc * 2)) != HU_OK || (rc = b->dSortV.ensure(nc * 2)) != HU_OK) return rc;
					k_cand_sortkeys<<<(unsigned)((nc + 255) / 256), 256, 0, b->stream>>>((int) nc, b->dCands.p, b->dSortK.p, b->dSortV.p);
					if((rc = order_by_key(b, b->dSortK.p, b->dSortV.p, nc, (uint32_t) b->db->dev.nNodes, b->dSortV.p + nc, false)) != HU_OK) return rc;
					order = b->dSortV.p + nc;
				}
				#define PL_GO(S, NW, E, R, O) k_place_blk<S, NW, E, R, O><<<grid, 64 * NW, 0, b->stream>>>(PL_ARGS, nullptr, order, nullptr, nullptr, xm)
				if((var == 99 || var == 98) && spt4 <= 6) { /* diagnostic: per-phase s_memtime stamps, averaged over the candidates, to stderr */
					DBuf<long long> ddb;
					if((rc = ddb.ensure(nc * 12)) != HU_OK) return rc;
					long long* dd = ddb.p;
					HIPCHK(hipMemsetAsync(dd, 0, nc * 12 * sizeof(long long), b->stream));
					if(var == 99) k_place_blk<6, 4, 3, 0, 1, true><<<grid, 256, 0, b->stream>>>(PL_ARGS, dd);
					else k_place_blk<12, 2, 3, 0, 2, true, 1><<<grid, 128, 0, b->stream>>>(PL_ARGS, dd);
					std::vector<long long> hd(nc * 12);
					HIPCHK(hipMemcpyAsync(hd.data(), dd, nc * 12 * sizeof(long long), hipMemcpyDeviceToHost, b->stream));
					HIPCHK(hu_wait(b->stream));
					double acc[8] = {0}, ae[4] = {0};
					for(size_t c = 0; c < nc; ++c) { for(int i = 0; i < 8; ++i) acc[i] += (double) hd[c * 8 + i]; for(int i = 0; i < 4; ++i) ae[i] += (double) hd[nc * 8 + c * 4 + i]; }
					fprintf(stderr, "[place dbg] per candidate (s_memtime ticks): load %.0f tables %.0f sweeps %.0f em %.0f total %.0f | outer %.2f em steps %.2f\n",
							acc[0] / nc, acc[1] / nc, acc[2] / nc, acc[3] / nc, acc[4] / nc, acc[5] / nc, acc[6] / nc);
					fprintf(stderr, "[place dbg] inside the EM steps, per step: arithmetic %.0f wave reduction %.0f exchange between the waves %.0f tail %.0f\n",
							ae[0] / acc[6], ae[1] / acc[6], ae[2] / acc[6], ae[3] / acc[6]);
				}
				else if(var == 4 && spt4 <= 6) PL_GO(6, 4, 1, 0, 1);
				else if(var == 5 && spt4 <= 6) PL_GO(6, 4, 2, 1, 1);
				else if(var == 6 && spt2 <= 12) PL_GO(12, 2, 1, 0, 2);
				else if(spt2 <= 4) PL_GO(4, 2, 3, 0, 2);
				else if(spt2 <= 12 && !(var == 7)) {
					/* 8 or 12 sites per thread.  When every read of the batch fits, its gap sites and its base sites go to separate
					 * slots (k_place_blk GS: 6 + 2 slots = up to 768 gap and 256 base sites, 10 + 2 = 1,280 and 256): the gap slots
					 * need no per-site table.  The counts come from k_site_count (estimate stage, read back by the filter stage). */
					const int S = spt2 <= 8 ? 8 : 12, G = S - 2;
					/* every read with a region fits the slots: the largest counts of the batch come with the candidate count (k_cand_scan) */
					const bool split = !pass && !b->knob.place_nosplit && b->maxGapSites <= G * 128 && b->maxBaseSites <= (S - G) * 128;
					if(b->knob.trace) fprintf(stderr, "[hu] place: %zu candidates, max region %d, %d sites per thread, %s, %s\n", nc, maxR, S, split ? "gap/base split slots" : "column order", "EM across both waves");
					if(split) {
						if((rc = b->dPerm.ensure((size_t) b->n * S * 128)) != HU_OK) return rc;
						k_site_perm<<<b->n, 64, 0, b->stream>>>(dev, b->n, b->dCodes.p, b->dStart.p, b->dEnd.p, G * 128, (S - G) * 128, b->dPerm.p);
						/* regions of <= 1,024 sites (150-base reads): the whole v message in LDS (VL = 3, 24 KB per workgroup), 168 VGPRs, three waves per
						 * SIMD: FIVE workgroups per CU instead of four (the LDS holds five) — the kernel's time goes with the resident candidates
						 * (DESIGN.md section 7): 4.18 -> 3.78 ms per 8,192 reads at gg_97 scale, 853 k -> 896 k reads/s.  Same arithmetic (results equal to 1e-12, iteration counts identical).
						 * Measured and not kept: the model constants read from global memory instead of 1.5 KB of LDS, which lets a sixth workgroup in — 32 B of scratch, 3.97 ms. */
						if(S == 8 && var != 6) k_place_blk<8, 2, 3, 0, 3, false, 3, 6><<<grid, 128, 3 * 8 * 128 * sizeof(double), b->stream>>>(PL_ARGS, nullptr, order, b->dPerm.p, b->dPermCnt.p, xm);
						else if(S == 8) k_place_blk<8, 2, 3, 0, 2, false, 0, 6><<<grid, 128, 0, b->stream>>>(PL_ARGS, nullptr, order, b->dPerm.p, b->dPermCnt.p, xm);   /* place_var = 6: v in registers, two waves per SIMD */
						else k_place_blk<12, 2, 3, 0, 2, false, 1, 10><<<grid, 128, (size_t)(b->knob.place_lds_pad > 0 && b->knob.place_lds_pad <= 44 ? b->knob.place_lds_pad : 0) * 1024, b->stream>>>(PL_ARGS, nullptr, order, b->dPerm.p, b->dPermCnt.p, xm);
					}
					else if(S == 8) PL_GO(8, 2, 3, 0, 2);
					else k_place_blk<12, 2, 3, 0, 2, false, 1><<<grid, 128, 0, b->stream>>>(PL_ARGS, nullptr, order, nullptr, nullptr, xm);
				}
				else if(var == 7 && spt2 <= 12) PL_GO(12, 2, 3, 0, 2);
				else if(spt4 <= 8) PL_GO(8, 4, 3, 0, 1);
				/* 256 VGPRs: one candidate of four waves per CU.  Measured and not kept: six waves x 8 sites with the v message in LDS (72 KB; 156 VGPRs, two
				 * candidates on twelve waves per CU): 8.53 ms against 8.2 per 4,096 pairs of 2 x 250 bases, 244 k against 277 k pairs/s — the sweeps read v
				 * through the LDS return path and every EM step crosses six waves */
				/* regions of 2,049 .. 3,072 columns (merged mate pairs).  With everything in registers the kernel needs all 256 VGPRs (+ AGPRs): ONE candidate of
				 * four waves per CU.  One component of v in LDS (VL = 1, 24 KB per workgroup) -> 239 VGPRs, two waves per SIMD, TWO candidates per CU:
				 * 8.2 -> 4.96 ms per 4,096 pairs of 2 x 250 bases, 277 k -> 309 k pairs/s (the kernel's time goes with the resident candidates) */
				else if(var == 9) PL_GO(12, 4, 3, 0, 1);     /* place_var = 9: the all-register form */
				else k_place_blk<12, 4, 3, 0, 2, false, 1><<<grid, 256, 0, b->stream>>>(PL_ARGS, nullptr, order, nullptr, nullptr, xm);
				#undef PL_GO
			}
			#undef PL_ARGS
		}
		HIPCHK(hipGetLastError());
		if(o->fix_root_loglik) { /* the intended root log-likelihood at the optimised lengths (documented deviation, off by default) */
			if((rc = b->dRootLL.ensure(nc)) != HU_OK) return rc;
			k_root_loglik<<<(unsigned) nc, 64, 0, b->stream>>>(b->db->dev, b->db->mdl, b->dCodes.p, b->dStart.p, b->dEnd.p, b->dCands.p, b->dPlaceOut.p, (int) nc, b->dRootLL.p);
			HIPCHK(hipGetLastError());
		}
		/* nothing comes back here: the finish stage reads the results where they are */
	}
	b->fixedRoot = o->fix_root_loglik != 0 && nc > 0;
	b->placesGiven = false;
	b->state = ST_PLACED;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_place_batch"); }


/* g_alleq_first (hu_kern_rank.h) once per device and process */
static int hu_alleq_table(int device) {
	static std::mutex mu; static bool done[64] = {false};
	std::lock_guard<std::mutex> lk(mu);
	if(device < 0 || device >= 64) { hu_set_error("device %d out of range", device); return HU_ERR_ARG; }
	if(done[device]) return HU_OK;
	(void) hipGetLastError();
	k_alleq_init<<<1, 128>>>();
	HIPCHK(hipGetLastError());
	HIPCHK(hipDeviceSynchronize());
	done[device] = true;
	return HU_OK;
}

extern "C" int hu_finish_batch(hu_batch* b, const hu_opts* o) try {
	if(!b || !o) return HU_ERR_ARG;
	if(b->state < ST_PLACED) { hu_set_error("hu_finish_batch: candidates are not placed"); return HU_ERR_STATE; }
	const hu_db* db = b->db;
	const size_t n = (size_t) b->n;
	if(b->knob.inject_fault == 2) throw std::length_error("injected");
	HIPCHK(hipSetDevice(db->device));
	b->best.assign(n, hu_place_rec());
	int rc;
	if((rc = b->dBest.ensure(std::max<size_t>(n, 1))) != HU_OK) return rc;
	(void) hipGetLastError();
	if(n) {
		/* PTPlacement after placeSeq (the F4 constant as loglik), calcQValues, the final std::sort and bestPlace: one thread per read
		 * on the device (hu_kern_rank.h); only the best record of every read comes back — the candidates' records stay until somebody asks */
		if((rc = hu_alleq_table(db->device)) != HU_OK) return rc;
		k_finish<<<(unsigned)((n + 63) / 64), 64, 0, b->stream>>>(db->dev, b->n, b->dStart.p, b->dEnd.p, b->dCandOff.p, b->dPlaces.p, b->dPlaceOut.p,
				b->fixedRoot ? b->dRootLL.p : nullptr, db->dLlTab, db->dAnnoId, o->max_height, o->only_ml, o->prior, b->placesGiven ? 1 : 0, b->dBest.p, b->knob.sort_seq);
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(b->best.data(), b->dBest.p, n * sizeof(hu_place_rec), hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hu_wait(b->stream));
	}
	b->hostCands = false;                /* the records changed on the device */
	b->state = ST_FINISHED;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_finish_batch"); }

extern "C" int hu_assign_batch(hu_batch* b, const hu_opts* o) try {
	int rc;
	if(!b || !o) return HU_ERR_ARG;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
	auto t0 = now();
	if(!b->fromCodes && (rc = hu_align_batch(b, o)) != HU_OK) return rc;
	auto t1 = now();
	if((rc = hu_seed_batch(b, o)) != HU_OK) return rc;
	if((rc = hu_estimate_batch(b, o)) != HU_OK) return rc;
	if((rc = hu_filter_batch(b, o)) != HU_OK) return rc;
	auto t2 = now();
	if((rc = hu_place_batch(b, o)) != HU_OK) return rc;
	auto t3 = now();
	rc = hu_finish_batch(b, o);
	auto t4 = now();
	b->wall[0] = ms(t0, t1); b->wall[1] = ms(t1, t2); b->wall[2] = ms(t2, t3); b->wall[3] = ms(t3, t4);
	return rc;
} catch(...) { return hu_catch_all("hu_assign_batch"); }
/* ------------------------------------------------------------------------------ chimera check (-C)
 * src/hmmufotu.cpp:653-691 as batch passes: each of the num_seg segments, then the two alt placements, is one
 * run of the given-seed / estimate / filter / place stages in the work batch over per-read regions. */
extern "C" void hu_default_chimera_opts(const hu_opts* o, hu_chimera_opts* co) try {
	if(!co) return;
	co->num_seg = 2; co->reserved = 0; co->max_chimera_error = (o ? o->max_error : 20.0) / co->num_seg; co->min_chimera_lod = 0;
} catch(...) { (void) hu_catch_all("hu_default_chimera_opts"); }
static hu_place_rec to_rec(const HostPlace& p, int32_t nCand);
struct SegPlace { HostPlace p; int32_t start, end; };
static bool cmpSegLoglik(const SegPlace& l, const SegPlace& r) { return l.p.loglik < r.p.loglik; }

static int segment_pass(hu_batch* w, hu_batch* b, const hu_opts* o, double maxError, const int32_t* st, const int32_t* en,
		const int32_t* cnt, const int32_t* ids, const int32_t* distIds) {
	int rc;
	if((rc = set_aligned_impl(w, b->n, b->dCodes.p, hipMemcpyDeviceToDevice, st, en, b->maxBases)) != HU_OK) return rc;
	if((rc = hu_seed_batch_given(w, cnt, ids, distIds, HU_MAX_SEEDS)) != HU_OK) return rc;
	hu_opts so = *o; so.max_error = maxError;
	if((rc = hu_estimate_batch(w, &so)) != HU_OK) return rc;
	if((rc = hu_filter_batch(w, &so)) != HU_OK) return rc;
	if((rc = hu_place_batch(w, &so)) != HU_OK) return rc;
	if((rc = hu_finish_batch(w, &so)) != HU_OK) return rc;
	return sync_host_cands(w);           /* the pools below are built on the host from every segment's candidates */
}

extern "C" int hu_chimera_batch(hu_batch* b, hu_batch* w, const hu_opts* o, const hu_chimera_opts* co, hu_chimera_rec* out) try {
	if(!b || !w || !o || !co || (b->n > 0 && !out) || b == w) { hu_set_error("hu_chimera_batch: bad argument"); return HU_ERR_ARG; }
	if(w->db != b->db || w->maxReads < b->n) { hu_set_error("hu_chimera_batch: the work batch must sit on the same database and hold %d reads", b->n); return HU_ERR_ARG; }
	if(b->state < ST_SEEDED) { hu_set_error("hu_chimera_batch: reads are not seeded"); return HU_ERR_STATE; }
	/* src/hmmufotu.cpp:325-340 */
	if(co->num_seg < 2 || co->num_seg > 6 || (co->num_seg % 2)) { hu_set_error("num_seg must be an even number in [2, 6]"); return HU_ERR_ARG; }
	if(!(co->max_chimera_error > 0)) { hu_set_error("max_chimera_error must be positive"); return HU_ERR_ARG; }
	if(!(co->min_chimera_lod >= 0)) { hu_set_error("min_chimera_lod must be non-negative"); return HU_ERR_ARG; }
	HIPCHK(hipSetDevice(b->db->device));
	const size_t n = (size_t) b->n;
	const int numSeg = co->num_seg;
	std::vector<int32_t> cnt(n), ids(n * HU_MAX_SEEDS), dist(n * HU_MAX_SEEDS), st(n), en(n), segLen(n), one(n);
	if(n) {
		HIPCHK(hipMemcpyAsync(cnt.data(), b->dSeedCnt.p, n * 4, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hipMemcpyAsync(ids.data(), b->dSeedId.p, n * HU_MAX_SEEDS * 4, hipMemcpyDeviceToHost, b->stream));
	}
	HIPCHK(hu_wait(b->stream)); /* the work batch reads b's codes on its own stream */
	for(size_t r = 0; r < n; ++r) {
		hu_chimera_rec& c = out[r];
		memset(&c, 0, sizeof(c));
		c.seg5_start = c.seg5_end = c.seg3_start = c.seg3_end = -1;
		hu_place_rec none; memset(&none, 0, sizeof(none));
		none.c_node = none.p_node = none.a_node = -1;
		none.wuv = none.ratio = none.wnr = none.loglik = none.height = none.q_place = none.q_taxon = none.anno_dist = none.est_loglik = none.root_loglik = NAN;
		c.seg5 = c.seg3 = none; c.alt5_loglik = c.alt3_loglik = c.lod = NAN;
		segLen[r] = b->hAlns[r].status == HU_READ_OK ? (b->hEnd[r] - b->hStart[r] + 1) / numSeg : 0;
		if(segLen[r] < 1 || cnt[r] < 1) { segLen[r] = 0; cnt[r] = 0; }
	}
	std::vector<std::vector<SegPlace>> pool5(n), pool3(n);
	int rc;
	for(int sg = 0; sg < numSeg; ++sg) {
		for(size_t r = 0; r < n; ++r) {
			if(segLen[r]) { st[r] = b->hStart[r] + sg * segLen[r]; en[r] = st[r] + segLen[r] - 1; }
			else { st[r] = 0; en[r] = -1; }
		}
		if((rc = segment_pass(w, b, o, co->max_chimera_error, st.data(), en.data(), cnt.data(), ids.data(), nullptr)) != HU_OK) return rc;
		for(size_t r = 0; r < n; ++r) {
			std::vector<SegPlace>& pool = sg < numSeg / 2 ? pool5[r] : pool3[r];
			for(int64_t c = w->candOffs[r]; c < w->candOffs[r + 1]; ++c) pool.push_back(SegPlace{w->places[c], st[r], en[r]});
		}
	}
	/* the same std::sort call on the same sequence: the placed logliks tie (F4), the order is its tie permutation */
	parallel_for(n, [&](size_t r) {
		std::sort(pool5[r].rbegin(), pool5[r].rend(), cmpSegLoglik);
		std::sort(pool3[r].rbegin(), pool3[r].rend(), cmpSegLoglik);
	});
	std::vector<uint8_t> ok(n, 0);
	for(size_t r = 0; r < n; ++r) ok[r] = segLen[r] && !pool5[r].empty() && !pool3[r].empty();
	for(int k = 0; k < 2; ++k) { /* k = 0: seg5's region on seg3's branch, distance to seg5's own node; k = 1 the mirror image */
		for(size_t r = 0; r < n; ++r) {
			one[r] = 0; st[r] = 0; en[r] = -1;
			if(!ok[r]) continue;
			const SegPlace& own = k == 0 ? pool5[r][0] : pool3[r][0];
			const SegPlace& oth = k == 0 ? pool3[r][0] : pool5[r][0];
			one[r] = 1; st[r] = own.start; en[r] = own.end;
			ids[r * HU_MAX_SEEDS] = oth.p.cNode; dist[r * HU_MAX_SEEDS] = own.p.cNode;
		}
		if((rc = segment_pass(w, b, o, INFINITY, st.data(), en.data(), one.data(), ids.data(), dist.data())) != HU_OK) return rc;
		for(size_t r = 0; r < n; ++r) {
			double& ll = k == 0 ? out[r].alt5_loglik : out[r].alt3_loglik;
			if(ok[r] && w->candOffs[r + 1] > w->candOffs[r]) ll = w->places[w->candOffs[r]].loglik;
		}
	}
	for(size_t r = 0; r < n; ++r) {
		if(!ok[r]) continue;
		hu_chimera_rec& c = out[r];
		const SegPlace& s5 = pool5[r][0]; const SegPlace& s3 = pool3[r][0];
		c.checked = 1;
		c.seg5_start = s5.start; c.seg5_end = s5.end; c.seg3_start = s3.start; c.seg3_end = s3.end;
		c.n_seg5 = (int32_t) pool5[r].size(); c.n_seg3 = (int32_t) pool3[r].size();
		c.seg5 = to_rec(s5.p, c.n_seg5); c.seg3 = to_rec(s3.p, c.n_seg3);
		c.lod = s5.p.loglik - c.alt5_loglik + s3.p.loglik - c.alt3_loglik;
		c.is_chimera = s5.p.aNode != s3.p.aNode && c.lod > co->min_chimera_lod;
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_chimera_batch"); }
/* host wall-clock of the last hu_assign_batch: align | seed+estimate+filter | place | finish (ms) */
extern "C" int hu_batch_refsort_stats(hu_batch* b, int32_t* left_to_host, int32_t* whole_batch_on_host) try {
	if(!b) return HU_ERR_ARG;
	if(left_to_host) *left_to_host = b->nRefBail;
	if(whole_batch_on_host) *whole_batch_on_host = b->refHostAll;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_refsort_stats"); }
extern "C" int hu_batch_wall(hu_batch* b, double* ms4) try { if(!b || !ms4) return HU_ERR_ARG; for(int i = 0; i < 4; ++i) ms4[i] = b->wall[i]; return HU_OK; } catch(...) { return hu_catch_all("hu_batch_wall"); }

/* ------------------------------------------------------------------------------ host helpers */
/* ---- column-window sharding: plan and routing (host only) */
extern "C" int hu_windows_plan(int64_t cs_len, int n_win, int64_t overlap, hu_window* out) try {
	if(cs_len < 1 || n_win < 1 || overlap < 0 || !out) { hu_set_error("hu_windows_plan: bad argument"); return HU_ERR_ARG; }
	/* n_win windows of width w at stride w - overlap cover cs_len: w = ceil((cs_len + (n_win - 1) overlap) / n_win) */
	const int64_t w = std::min<int64_t>(cs_len, (cs_len + (int64_t)(n_win - 1) * overlap + n_win - 1) / n_win);
	if(n_win > 1 && w <= overlap) { hu_set_error("hu_windows_plan: %d windows overlapping by %lld columns do not advance over %lld columns", n_win, (long long) overlap, (long long) cs_len); return HU_ERR_ARG; }
	for(int i = 0; i < n_win; ++i) {
		int64_t st = (int64_t) i * (w - overlap);
		if(st + w > cs_len) st = cs_len - w;          /* the last window ends with the consensus */
		out[i].win_start = st; out[i].win_len = w;
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_windows_plan"); }

static int route_interval(int n_win, const hu_window* win, int64_t lo, int64_t hi /* 0-based inclusive columns */) {
	int best = -1; int64_t bestMargin = -1, bestOv = -1; int bestOvW = 0;
	for(int w = 0; w < n_win; ++w) {
		const int64_t ws = win[w].win_start, we = ws + win[w].win_len - 1;
		if(lo >= ws && hi <= we) { const int64_t m = std::min(lo - ws, we - hi); if(m > bestMargin) { bestMargin = m; best = w; } }
		const int64_t ov = std::min(hi, we) - std::max(lo, ws) + 1;
		if(ov > bestOv) { bestOv = ov; bestOvW = w; }
	}
	return best >= 0 ? best : -1 - bestOvW;         /* -1 - w: no window contains the interval, w overlaps it most */
}
extern "C" int hu_route_by_region(int n_win, const hu_window* win, int n, const int32_t* cs_start, const int32_t* cs_end, int32_t* out) try {
	if(n_win < 1 || !win || n < 0 || (n > 0 && (!cs_start || !cs_end || !out))) { hu_set_error("hu_route_by_region: bad argument"); return HU_ERR_ARG; }
	for(int r = 0; r < n; ++r) { const int w = route_interval(n_win, win, (int64_t) cs_start[r] - 1, (int64_t) cs_end[r] - 1); out[r] = w >= 0 ? w : -1; }
	return HU_OK;
} catch(...) { return hu_catch_all("hu_route_by_region"); }
extern "C" int hu_route_by_seeds(const hu_db* db, int n_win, const hu_window* win, int n, const int32_t* lens, const int32_t* vpaths,
		const int32_t* mate_lens, const int32_t* mate_vpaths, int32_t* out) try {
	if(!db || n_win < 1 || !win || n < 0 || (n > 0 && (!lens || !vpaths || !out)) || ((mate_lens == nullptr) != (mate_vpaths == nullptr))) { hu_set_error("hu_route_by_seeds: bad argument"); return HU_ERR_ARG; }
	const int K = db->dev.K;
	const std::vector<int32_t>& p2cs = db->prof.p2cs;       /* [K + 1], 1-based CS column of a profile position */
	auto span = [&](int len, const int32_t* vp, int64_t& lo, int64_t& hi) -> bool { /* profile positions of the read's first and last base, from its seed paths */
		const bool s5 = vp[0] > 0, s3 = vp[6] > 0;
		if(!s5 && !s3) return false;
		const int slack = len / 8 + 4;                      /* indels between the seeds and the read's ends */
		int ps, pe;
		if(s5) ps = vp[0] - (vp[2] - 1); else ps = vp[7] - (vp[9] - 1);          /* start - (from - 1) | end - (to - 1) of the 3' seed */
		if(s3) pe = vp[7] + (len - vp[9]); else pe = vp[1] + (len - vp[3]);       /* end + (len - to) */
		ps = std::max(1, std::min(K, ps - slack)); pe = std::max(1, std::min(K, pe + slack));
		if(pe < ps) std::swap(ps, pe);
		lo = (int64_t) p2cs[ps] - 1; hi = (int64_t) p2cs[pe] - 1;
		return true;
	};
	for(int r = 0; r < n; ++r) {
		int64_t lo = 0, hi = 0, l2, h2;
		bool have = span(lens[r], vpaths + (size_t) r * 12, lo, hi);
		if(mate_lens && span(mate_lens[r], mate_vpaths + (size_t) r * 12, l2, h2)) { if(have) { lo = std::min(lo, l2); hi = std::max(hi, h2); } else { lo = l2; hi = h2; have = true; } }
		if(!have) { out[r] = 0; continue; }                 /* no seed: any window aligns it; its region decides afterwards */
		const int w = route_interval(n_win, win, lo, hi);
		out[r] = w >= 0 ? w : -1 - w;
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_route_by_seeds"); }

extern "C" int hu_build_align_path(const hu_db* db, int cs_start, int cs_end, const char* cs, int cs_from, int cs_to, int32_t* out6) try {
	if(!db || !cs || !out6) return HU_ERR_ARG;
	(void) cs_end; (void) cs_to;
	const std::vector<int32_t>& cs2p = db->prof.cs2p;
	int start = 0, end = 0, from = 0, to = 0, nIns = 0, nDel = 0;
	int i = cs_from, j = cs_start;
	for(const char* c = cs; *c; ++c) {
		const int k = (j >= 0 && j < (int) cs2p.size()) ? cs2p[j] : 0;   /* getProfileLoc */
		const bool nonGap = host_sym(*c) >= 0;                            /* abc->isSymbol */
		if(from == 0 && nonGap) from = i;
		if(nonGap) to = i;
		if(k != 0) { if(start == 0) start = k; end = k; if(!nonGap) nDel++; }
		else if(nonGap) nIns++;
		j++;
		if(nonGap) i++;
	}
	out6[0] = start; out6[1] = end; out6[2] = from; out6[3] = to; out6[4] = nIns; out6[5] = nDel;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_build_align_path"); }

extern "C" const char* hu_tsv_header(void) {
	return "id\tdescription\tseq_start\tseq_end\thmm_start\thmm_end\tCS_start\tCS_end\tcost\talignment\t"
	       "branch_id\tbranch_ratio\ttaxon_id\ttaxon_anno\tanno_dist\tloglik\tQ_placement\tQ_taxon";
}


/* which = 0: the assignment file (reads that are HU_READ_OK and not flagged); which = 1: --chimera-out (bad PE orientation
 * or flagged by the check; the placement columns are a default-constructed PTPlacement, src/hmmufotu.cpp:693-706);
 * which = 2: the assignment file of --align-only (as 0, nothing placed: default placement columns, :717-739) */
static int64_t format_tsv_impl(hu_batch* b, const char* const* ids, const char* const* descs, const char* const* annos,
		const hu_chimera_rec* chi, int info, int which) {
	if(!b || !ids) { hu_set_error("hu_batch_format_tsv: bad argument"); return HU_ERR_ARG; }
	if((which == 0 && b->state < ST_FINISHED) || b->state < ST_ALIGNED || b->fromCodes) { hu_set_error("hu_batch_format_tsv: batch is not finished"); return HU_ERR_STATE; }
	const int L = b->db->dev.csLen;
	/* the alignment rows (csLen bytes per read: 63 MB per 8,192 reads) come through a page-locked buffer the batch keeps */
	b->hRows.resize((size_t) b->n * L);
	if(b->n) {
		if(hipSetDevice(b->db->device) != hipSuccess || hipMemcpyAsync(b->hRows.data(), b->dRows.p, b->hRows.size(), hipMemcpyDeviceToHost, b->stream) != hipSuccess ||
				hu_wait(b->stream) != hipSuccess) { hu_set_error("hu_batch_format_tsv: device copy failed"); return HU_ERR_DEVICE; }
	}
	/* lines are written per read into slices of one buffer, the reads spread over the host pool: pass 1 sizes, pass 2 fills */
	const size_t n = (size_t) b->n;
	std::vector<size_t>& off = b->tsvOff;
	off.assign(n + 1, 0);
	auto put_int = [](char* p, long v) -> char* { char t[24]; int k = 0; unsigned long u = v < 0 ? 0ul - (unsigned long) v : (unsigned long) v; do { t[k++] = (char)('0' + u % 10); u /= 10; } while(u); if(v < 0) *p++ = '-'; while(k) *p++ = t[--k]; return p; };
	auto put_gd = [](char* p, double v) -> char* { return p + snprintf(p, 40, "%g", v); };   /* operator<<(ostream&, double) at default precision == printf("%g") */
	auto put_s = [](char* p, const char* s) -> char* { const size_t k = strlen(s); memcpy(p, s, k); return p + k; };
	auto wanted = [&](size_t r) {
		const HuAlnDev& a = b->hAlns[r];
		const bool flagged = chi && a.status == HU_READ_OK && chi[r].is_chimera;
		return which != 1 ? !(a.status != HU_READ_OK || flagged) : (a.status == HU_READ_CHIMERA || flagged);
	};
	const int fault = b->knob.inject_fault;
	parallel_for(n, [&](size_t r) { /* an upper bound of the line's length */
		if(fault == 3 && r + 1 == n) throw std::runtime_error("injected");
		if(!wanted(r)) { off[r + 1] = 0; return; }
		size_t len = strlen(ids[r]) + (descs && descs[r] ? strlen(descs[r]) : 0) + (size_t) L + 6 * 12 + 40 + 16 /* tabs, newline */ + 2 * 12 + 16 + 12 + 5 * 40;
		if(info) {
			const bool ck = chi && b->hAlns[r].status == HU_READ_OK && chi[r].checked;
			const int32_t t5 = ck ? chi[r].seg5.a_node : -1, t3 = ck ? chi[r].seg3.a_node : -1;
			len += 2 * 12 + 40 + 8 + (t5 >= 0 ? (annos && annos[t5] ? strlen(annos[t5]) : 0) : 10) + (t3 >= 0 ? (annos && annos[t3] ? strlen(annos[t3]) : 0) : 10);
		}
		if(which == 0 && b->best[r].c_node >= 0) len += annos && annos[b->best[r].a_node] ? strlen(annos[b->best[r].a_node]) : 0;
		else len += 48;
		off[r + 1] = len;
	});
	for(size_t r = 0; r < n; ++r) off[r + 1] += off[r];
	std::vector<char>& buf = b->tsvBuf;
	if(buf.size() < off[n] + 1) buf.resize(off[n] + 1);
	std::vector<size_t>& used = b->tsvLen;
	used.assign(n, 0);
	parallel_for(n, [&](size_t r) {
		if(off[r + 1] == off[r]) return;
		const HuAlnDev& a = b->hAlns[r];
		char* p0 = buf.data() + off[r]; char* p = p0;
		p = put_s(p, ids[r]); *p++ = '\t'; if(descs && descs[r]) p = put_s(p, descs[r]); *p++ = '\t';
		p = put_int(p, a.seqStart); *p++ = '\t'; p = put_int(p, a.seqEnd); *p++ = '\t'; p = put_int(p, a.hmmStart); *p++ = '\t'; p = put_int(p, a.hmmEnd); *p++ = '\t';
		p = put_int(p, a.csStart); *p++ = '\t'; p = put_int(p, a.csEnd); *p++ = '\t';
		p = put_gd(p, a.cost); *p++ = '\t';
		memcpy(p, &b->hRows[r * (size_t) L], (size_t) L); p += L; *p++ = '\t';
		if(info) { /* CHIMERA_TSV_HEADER columns (src/hmmufotu.cpp:57, 701-705, 742-746); unchecked reads print default placements */
			const bool ck = chi && a.status == HU_READ_OK && chi[r].checked;
			const int32_t t5 = ck ? chi[r].seg5.a_node : -1, t3 = ck ? chi[r].seg3.a_node : -1;
			p = put_int(p, t5); *p++ = '\t'; p = put_int(p, t3); *p++ = '\t';
			if(t5 >= 0) { if(annos && annos[t5]) p = put_s(p, annos[t5]); } else p = put_s(p, "UNASSIGNED");
			*p++ = '\t';
			if(t3 >= 0) { if(annos && annos[t3]) p = put_s(p, annos[t3]); } else p = put_s(p, "UNASSIGNED");
			*p++ = '\t';
			p = put_gd(p, ck ? chi[r].lod : NAN); *p++ = '\t';
		}
		hu_place_rec none; none.c_node = -1;
		const hu_place_rec& pl = which == 0 ? b->best[r] : none;
		if(pl.c_node >= 0) {
			p = put_int(p, pl.c_node); *p++ = '-'; *p++ = '>'; p = put_int(p, pl.p_node); *p++ = '\t'; p = put_gd(p, pl.ratio); *p++ = '\t';
			p = put_int(p, pl.a_node); *p++ = '\t'; if(annos && annos[pl.a_node]) p = put_s(p, annos[pl.a_node]); *p++ = '\t';
			p = put_gd(p, pl.anno_dist); *p++ = '\t'; p = put_gd(p, pl.loglik); *p++ = '\t'; p = put_gd(p, pl.q_place); *p++ = '\t'; p = put_gd(p, pl.q_taxon);
		}
		else p = put_s(p, "NULL\tnan\t-1\tUNASSIGNED\tnan\tnan\tnan\tnan"); /* default-constructed PTPlacement (src/PhyloTreeUnrooted.cpp:60-65) */
		*p++ = '\n';
		used[r] = (size_t)(p - p0);
	});
	/* close the gaps between the slices (each was sized by an upper bound) */
	size_t w = 0;
	for(size_t r = 0; r < n; ++r) { if(used[r]) { if(w != off[r]) memmove(buf.data() + w, buf.data() + off[r], used[r]); w += used[r]; } }
	b->tsvSize = w;
	return (int64_t) w;
}
static int64_t format_copy(hu_batch* b, int64_t need, char* buf, int64_t cap) {
	if(need > 0 && buf && cap > 0) memcpy(buf, b->tsvBuf.data(), (size_t) std::min<int64_t>(cap, need));
	return need;
}
extern "C" int64_t hu_batch_format_tsv(hu_batch* b, const char* const* ids, const char* const* descs, const char* const* annos,
		char* buf, int64_t cap) try {
	return format_copy(b, format_tsv_impl(b, ids, descs, annos, nullptr, 0, 0), buf, cap);
} catch(...) { return hu_catch_all("hu_batch_format_tsv"); }
extern "C" int64_t hu_batch_format_tsv_chimera(hu_batch* b, const char* const* ids, const char* const* descs, const char* const* annos,
		const hu_chimera_rec* chi, int chimera_info, int which, char* buf, int64_t cap) try {
	if(which < 0 || which > 2) { hu_set_error("hu_batch_format_tsv_chimera: which must be 0, 1 or 2"); return HU_ERR_ARG; }
	return format_copy(b, format_tsv_impl(b, ids, descs, annos, chi, chimera_info, which), buf, cap);
} catch(...) { return hu_catch_all("hu_batch_format_tsv_chimera"); }
extern "C" int64_t hu_batch_format_tsv_ptr(hu_batch* b, const char* const* ids, const char* const* descs, const char* const* annos,
		const hu_chimera_rec* chi, int chimera_info, int which, const char** text) try {
	if(which < 0 || which > 2 || !text) { hu_set_error("hu_batch_format_tsv_ptr: bad argument"); return HU_ERR_ARG; }
	const int64_t need = format_tsv_impl(b, ids, descs, annos, chi, chimera_info, which);
	*text = need >= 0 ? b->tsvBuf.data() : nullptr;
	return need;
} catch(...) { return hu_catch_all("hu_batch_format_tsv_ptr"); }
extern "C" int hu_batch_tsv_line_lengths(hu_batch* b, int64_t* lens) try {
	if(!b || (b->n > 0 && !lens)) { hu_set_error("hu_batch_tsv_line_lengths: bad argument"); return HU_ERR_ARG; }
	if(b->tsvLen.size() != (size_t) b->n) { hu_set_error("hu_batch_tsv_line_lengths: no lines were formatted for this batch"); return HU_ERR_STATE; }
	for(int r = 0; r < b->n; ++r) lens[r] = (int64_t) b->tsvLen[(size_t) r];
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_tsv_line_lengths"); }
extern "C" const char* hu_tsv_header_chimera(void) {
	return "id\tdescription\tseq_start\tseq_end\thmm_start\thmm_end\tCS_start\tCS_end\tcost\talignment\t"
	       "seg5_taxon_id\tseg3_taxon_id\tseg5_taxon_anno\tseg3_taxon_anno\tchimera_lod\t"
	       "branch_id\tbranch_ratio\ttaxon_id\ttaxon_anno\tanno_dist\tloglik\tQ_placement\tQ_taxon";
}

/* ------------------------------------------------------------------------------ results */
extern "C" int hu_batch_get_alignments(hu_batch* b, hu_align_rec* recs, char* align, char* trace, int trace_stride) try {
	if(!b) return HU_ERR_ARG;
	if(b->state < ST_ALIGNED) { hu_set_error("no alignments yet"); return HU_ERR_STATE; }
	HIPCHK(hipSetDevice(b->db->device));
	const HuDbDev& d = b->db->dev;
	if(recs) for(int r = 0; r < b->n; ++r) {
		const HuAlnDev& a = b->hAlns[r];
		recs[r].seq_start = a.seqStart; recs[r].seq_end = a.seqEnd; recs[r].hmm_start = a.hmmStart; recs[r].hmm_end = a.hmmEnd;
		recs[r].cs_start = a.csStart; recs[r].cs_end = a.csEnd; recs[r].status = a.status; recs[r].used_full = a.usedFull; recs[r].cost = a.cost;
	}
	if(align && !b->fromCodes && b->n) {
		HIPCHK(hipMemcpyAsync(align, b->dRows.p, (size_t) b->n * d.csLen, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hu_wait(b->stream));
	}
	if(trace && !b->fromCodes && trace_stride > 0) {
		std::vector<char> all(b->hDescs.empty() ? 0 : (size_t)(b->hDescs.back().traceOff + b->hDescs.back().len + d.K + 8));
		if(!all.empty()) { HIPCHK(hipMemcpyAsync(all.data(), b->dTraces.p, all.size(), hipMemcpyDeviceToHost, b->stream)); HIPCHK(hu_wait(b->stream)); }
		for(int r = 0; r < b->n; ++r) {
			const int len = std::min(b->hVit[r].traceLen, trace_stride - 1);
			memcpy(trace + (size_t) r * trace_stride, all.data() + b->hDescs[r].traceOff, len > 0 ? len : 0);
			trace[(size_t) r * trace_stride + (len > 0 ? len : 0)] = 0;
		}
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_get_alignments"); }
extern "C" int hu_batch_get_codes(hu_batch* b, int8_t* codes, int32_t* start, int32_t* end) try {
	if(!b) return HU_ERR_ARG;
	if(b->state < ST_ALIGNED) { hu_set_error("no alignments yet"); return HU_ERR_STATE; }
	HIPCHK(hipSetDevice(b->db->device));
	if(codes && b->n) { HIPCHK(hipMemcpyAsync(codes, b->dCodes.p, (size_t) b->n * b->db->dev.csLen, hipMemcpyDeviceToHost, b->stream)); HIPCHK(hu_wait(b->stream)); }
	if(start) memcpy(start, b->hStart.data(), (size_t) b->n * 4);
	if(end) memcpy(end, b->hEnd.data(), (size_t) b->n * 4);
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_get_codes"); }
extern "C" int hu_batch_get_pdist(hu_batch* b, int read, int32_t* d, int32_t* N) try {
	if(!b || read < 0 || read >= b->n) return HU_ERR_ARG;
	if(b->state < ST_SEEDED) { hu_set_error("no seed scan yet"); return HU_ERR_STATE; }
	HIPCHK(hipSetDevice(b->db->device));
	const int nn = b->db->dev.nNodes;
	std::vector<uint32_t> v(nn);
	if(b->pairsKind == 0) {
		/* no pair matrix (distance-only scan or given seeds): the pairs straight from the planes; where the scan left its
		 * distance matrix, that row is checked against them (saturated at the matrix's width) */
		DBuf<uint32_t> tmp;
		int rc;
		if((rc = tmp.ensure((size_t) nn * 2)) != HU_OK) return rc;
		k_pairs_of_read<<<(nn + 255) / 256, 256, 0, b->stream>>>(b->db->dev, read_planes(b), read, tmp.p, b->scanWidth ? tmp.p + nn : nullptr);
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(v.data(), tmp.p, (size_t) nn * 4, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hu_wait(b->stream));
		if(b->scanWidth) { /* the scan leaves the read's LISTED inserts out */
			std::vector<uint8_t> row((size_t) nn * b->scanWidth);
			std::vector<uint32_t> want(nn);
			HIPCHK(hipMemcpyAsync(row.data(), (const uint8_t*) b->dPairs.p + (size_t) read * b->db->dev.nNodesPad * b->scanWidth, row.size(), hipMemcpyDeviceToHost, b->stream));
			HIPCHK(hipMemcpyAsync(want.data(), tmp.p + nn, (size_t) nn * 4, hipMemcpyDeviceToHost, b->stream));
			HIPCHK(hu_wait(b->stream));
			const uint32_t sat = b->scanWidth == 1 ? 255u : 65535u;
			for(int i = 0; i < nn; ++i) {
				const uint32_t got = b->scanWidth == 1 ? row[i] : ((const uint16_t*) row.data())[i];
				if(got != std::min(want[i] >> 16, sat)) { hu_set_error("hu_batch_get_pdist: the distance-only scan holds %u for read %d, node %d; the planes give %u", got, read, i, want[i] >> 16); return HU_ERR_STATE; }
			}
		}
	}
	else if(b->pair16) {
		std::vector<uint16_t> h(nn);
		HIPCHK(hipMemcpyAsync(h.data(), (const uint16_t*) b->dPairs.p + (size_t) read * b->db->dev.nNodesPad, (size_t) nn * 2, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hu_wait(b->stream));
		for(int i = 0; i < nn; ++i) v[i] = ((uint32_t)(h[i] >> 8) << 16) | (h[i] & 0xffu);
	}
	else {
		HIPCHK(hipMemcpyAsync(v.data(), b->dPairs.p + (size_t) read * b->db->dev.nNodesPad, (size_t) nn * 4, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hu_wait(b->stream));
	}
	for(int i = 0; i < nn; ++i) { if(d) d[i] = (int32_t)(v[i] >> 16); if(N) N[i] = (int32_t)(v[i] & 0xffffu); }
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_get_pdist"); }
extern "C" int hu_batch_get_seeds(hu_batch* b, int32_t* n_seeds, int32_t* ids, int32_t* d, int32_t* N) try {
	return hu_batch_get_seeds_strided(b, n_seeds, ids, d, N, HU_MAX_SEEDS);
} catch(...) { return hu_catch_all("hu_batch_get_seeds"); }
extern "C" int hu_batch_get_estimates(hu_batch* b, double* ratio, double* wnr, double* loglik) try {
	return hu_batch_get_estimates_strided(b, ratio, wnr, loglik, HU_MAX_SEEDS);
} catch(...) { return hu_catch_all("hu_batch_get_estimates"); }
extern "C" int hu_batch_get_seeds_strided(hu_batch* b, int32_t* n_seeds, int32_t* ids, int32_t* d, int32_t* N, int stride) try {
	if(!b || stride < 1) return HU_ERR_ARG;
	if(b->state < ST_SEEDED) { hu_set_error("no seeds yet"); return HU_ERR_STATE; }
	HIPCHK(hipSetDevice(b->db->device));
	const size_t n = (size_t) b->n;
	std::vector<int32_t> cnt(n), id(n * HU_MAX_SEEDS); std::vector<uint32_t> dn(n * HU_MAX_SEEDS);
	if(n) {
		HIPCHK(hipMemcpyAsync(cnt.data(), b->dSeedCnt.p, n * 4, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hipMemcpyAsync(id.data(), b->dSeedId.p, n * HU_MAX_SEEDS * 4, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hipMemcpyAsync(dn.data(), b->dSeedDN.p, n * HU_MAX_SEEDS * 4, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hu_wait(b->stream));
	}
	for(size_t r = 0; r < n; ++r) {
		if(n_seeds) n_seeds[r] = cnt[r];
		for(int s = 0; s < HU_MAX_SEEDS && s < stride; ++s) {
			const bool ok = s < cnt[r];
			const size_t k = r * HU_MAX_SEEDS + s, w = r * (size_t) stride + s;
			if(ids) ids[w] = ok ? id[k] : -1;
			if(d) d[w] = ok ? (int32_t)(dn[k] >> 16) : 0;
			if(N) N[w] = ok ? (int32_t)(dn[k] & 0xffffu) : 0;
		}
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_get_seeds_strided"); }
extern "C" int hu_batch_get_estimates_strided(hu_batch* b, double* ratio, double* wnr, double* loglik, int stride) try {
	if(!b || stride < 1) return HU_ERR_ARG;
	if(b->state < ST_ESTIMATED) { hu_set_error("no estimates yet"); return HU_ERR_STATE; }
	HIPCHK(hipSetDevice(b->db->device));
	const size_t n = (size_t) b->n;
	std::vector<HuEstOut> e(n * HU_MAX_SEEDS); std::vector<int32_t> cnt(n);
	if(n) {
		HIPCHK(hipMemcpyAsync(e.data(), b->dEst.p, e.size() * sizeof(HuEstOut), hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hipMemcpyAsync(cnt.data(), b->dSeedCnt.p, n * 4, hipMemcpyDeviceToHost, b->stream));
		HIPCHK(hu_wait(b->stream));
	}
	for(size_t r = 0; r < n; ++r) for(int s = 0; s < HU_MAX_SEEDS && s < stride; ++s) {
		const size_t k = r * HU_MAX_SEEDS + s, w = r * (size_t) stride + s; const bool ok = s < cnt[r];
		if(ratio) ratio[w] = ok ? e[k].ratio : NAN;
		if(wnr) wnr[w] = ok ? e[k].wnr : NAN;
		if(loglik) loglik[w] = ok ? e[k].loglik : NAN;
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_get_estimates_strided"); }
extern "C" int hu_batch_get_candidates(hu_batch* b, int64_t* offs, int32_t* c_node, double* ratio, double* wnr, double* est_loglik, int32_t* iters) try {
	if(!b) return HU_ERR_ARG;
	if(b->state < ST_FILTERED) { hu_set_error("no candidates yet"); return HU_ERR_STATE; }
	{ const int rc = sync_host_cands(b); if(rc != HU_OK) return rc; }
	if(offs) memcpy(offs, b->candOffs.data(), b->candOffs.size() * 8);
	for(size_t c = 0; c < b->places.size(); ++c) {
		const HostPlace& p = b->places[c];
		if(c_node) c_node[c] = p.cNode; if(ratio) ratio[c] = p.ratio; if(wnr) wnr[c] = p.wnr;
		if(est_loglik) est_loglik[c] = p.estLoglik; if(iters) iters[c] = p.iters;
	}
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_get_candidates"); }
static hu_place_rec to_rec(const HostPlace& p, int32_t nCand) {
	hu_place_rec r;
	r.c_node = p.cNode; r.p_node = p.pNode; r.a_node = p.aNode; r.n_cand = nCand;
	r.wuv = p.wuv; r.ratio = p.ratio; r.wnr = p.wnr; r.loglik = p.loglik; r.height = p.height;
	r.q_place = p.qPlace; r.q_taxon = p.qTaxon; r.anno_dist = p.annoDist(); r.est_loglik = p.estLoglik; r.root_loglik = p.rootLoglik;
	return r;
}
extern "C" int hu_batch_get_candidate_places(hu_batch* b, hu_place_rec* recs) try {
	if(!b || !recs) return HU_ERR_ARG;
	if(b->state < ST_FINISHED) { hu_set_error("batch is not finished"); return HU_ERR_STATE; }
	{ const int rc = sync_host_cands(b); if(rc != HU_OK) return rc; }
	for(size_t r = 0; r + 1 < b->candOffs.size(); ++r)
		for(int64_t c = b->candOffs[r]; c < b->candOffs[r + 1]; ++c) recs[c] = to_rec(b->places[c], (int32_t)(b->candOffs[r + 1] - b->candOffs[r]));
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_get_candidate_places"); }
extern "C" int hu_batch_get_placements(hu_batch* b, hu_place_rec* best) try {
	if(!b || !best) return HU_ERR_ARG;
	if(b->state < ST_FINISHED) { hu_set_error("batch is not finished"); return HU_ERR_STATE; }
	memcpy(best, b->best.data(), b->best.size() * sizeof(hu_place_rec));
	return HU_OK;
} catch(...) { return hu_catch_all("hu_batch_get_placements"); }
