// Shared host/device declarations of the engine.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <functional>
#include <istream>
#include <vector>
#include <exception>
#include <mutex>
#include <thread>
#include "../../include/hmmufotu_amd.h"

#define HU_MAX_DGK 16
/* HU_MAX_SEEDS (64: capacity of the per-read seed lists) is part of the ABI: include/hmmufotu_amd.h */
#define HU_READ_TILE 16          /* reads per workgroup tile in the seed p-distance scan        */
#define HU_MAX_INS 24            /* bases of one read outside the profile columns handled as a list */
#define HU_NODE_PAD 256          /* node count is padded to a multiple of this                  */
#define HU_MIN_LOGLIK_EXP (-510.0) /* DBL_MIN_EXP / 2 in integer arithmetic, PhyloTreeUnrooted.cpp:68 */
#define HU_BRANCH_EPS 1e-5
#define HU_MAX_ITER 100

/* Every reversible model of the reference has Q(i,j) = s(i,j) pi(j) with symmetric s, hence a
 * real spectral form; the engine evaluates DNASubModel::Pr(t) as U diag(exp(lam t)) U1 for all
 * six model types (GTR does exactly this in the reference, src/GTR.h:116-121; the closed forms
 * of TN93/HKY85/F81/K80/JC69 are the same matrix exponentials written out). */
struct HuModelDev {
	int32_t type, dgK;
	double pi[4];
	double logpi[4];
	double U[16], U1[16], lam[4];
	double rate[HU_MAX_DGK];   /* rate[0] = 1 when dgK == 0 */
};

struct HuDbDev {
	int32_t nNodes, nNodesPad, csLen, W, WQ, root;
	int64_t winStart, winLen;
	const uint4* planes;       /* [WQ][3][nNodesPad]: 4 words (128 sites) of bit-plane p per node, in SCAN ORDER */
	const int32_t* posCol;     /* [WQ*128] scan position -> CS column (-1 = padding): the profile (match) columns
	                            * first, then the others.  A p-distance is a sum over columns in any order;
	                            * in this order the few hundred bases of a read sit in 2-3 quads of 128 positions
	                            * plus the quads of its rare inserts, instead of the ~12 quads of its CS window */
	/* the non-profile scan positions once more, transposed: for position QM*128 + x and plane p one bit per NODE,
	 * colPlanes[(x * 3 + p) * (nNodesPad / 64) + node / 64] — what a wave of 64 consecutive nodes needs of one insert
	 * position is three 8-byte words (in `planes` it is 3 KB: a whole quad's lines for one bit per node) */
	const unsigned long long* colPlanes;
	int32_t QM, pad0;
	/* per node: first and last scan position that holds a base, in the profile block [0, QM * 128) and in the non-profile block
	 * (x = first1 | last1 << 16, y = first2 | last2 << 16; first > last: none there).  A node whose two intervals miss a read's two
	 * (HuReadPlanes::rspan) shares no valid position with it: N = 0 exactly — reference sequences that do not cover the amplicon. */
	const uint2* nodeCover;
	const int32_t* parent;
	const double* blen;
	const double* height;
	/* directed-edge messages, packed once at load time into LINEAR space: 4 doubles
	 * e_i = exp(M_i) * 2^-k with k = rint(max_i M_i / ln 2) kept in upK/downK, so that the kernels
	 * never evaluate exp() on a message again (the reference does, per site, per use) */
	const double* up;          /* [n][winLen][4] */
	const double* down;
	const int32_t* upK;        /* [n][winLen] */
	const int32_t* downK;
	/* profile */
	int32_t K, L;
	const double* EM;          /* [K+1][4] */
	const double* EI;
	const double* T;           /* [K+1][8] (7 used) */
	const double* entryC;
	const double* exitC;
	const int32_t* p2cs;       /* [K+2] */
	/* the same profile, one array per field ([7][K+1], [4][K+1], [4][K+1]): along an anti-diagonal of the DP
	 * consecutive lanes sit in consecutive columns, so these loads coalesce (the [K+1][8] rows above put every
	 * lane on its own cache line: k_viterbi_lds was bound by L1 tag throughput and by its 24-B-per-cell stores) */
	const double* Tt;
	const double* EMt;
	const double* EIt;
	const double* placeConst;  /* [HU_PC_COUNT] */
	/* width classes of one launch of the estimate / placement kernels (0, 0, nullptr: every read).  The kernel shape goes with the widest
	 * alignment region it must hold, and a handful of reads per batch whose seeds land far apart have regions of thousands of columns:
	 * they get a launch of their own (the few slots listed in wideList) instead of dragging the whole batch onto the wide kernel */
	int32_t rLo, rHi;          /* rHi != 0: only reads with rLo < region columns <= rHi */
	const uint32_t* wideList;  /* streaming kernels: workgroup -> slot / candidate (the others take it as their `order`) */
};
__device__ inline bool hu_skip_width(const HuDbDev& db, int cols) { return db.rHi != 0 && (cols <= db.rLo || cols > db.rHi); }

/* one dynamic-programming phase of the banded Viterbi (src/BandedHMMP7.cpp:794-881) */
struct HuRegion {
	int32_t j0, j1, i0, i1;    /* inclusive profile / read ranges                                  */
	int32_t withB;             /* B-entry term present (absent in the downstream rectangle)       */
	int32_t band;              /* 1: only cells with -nDel <= (i-from)-(j-start) <= nIns           */
	int32_t from, start, nIns, nDel;
	int64_t off;               /* first cell of this region in the read's scratch                 */
	int64_t doff;              /* first byte of this region in the read's decision scratch: one byte per cell in
	                            * anti-diagonal order, [dg][q] with pitch (i1 - i0 + 1) rounded up to 16                        */
	/* the CORNER BLOCK of the region: rows ci0 .. i1 x columns cj0 .. j1, the only cells a later phase can look up (their row is at or
	 * below the row above the first later phase, their column likewise: consecutive phases share a corner).  The one-wave kernel
	 * (k_viterbi_wave) files (M, I, D) of these cells alone, column-major with a pitch of i1 - ci0 + 1, at cell `coff` of the read's
	 * corner scratch — a few cells per read, where the value-filing kernels keep every cell of every phase (`off`: 2-3 MB per read) */
	int32_t ci0, cj0;
	int64_t coff;
};
#define HU_MAX_REGIONS 6
#define HU_READ_NEEDS_VALUES 8   /* internal: the decision-byte traceback met a cell whose predecessor a later phase
                                  * rewrote (or left the computed cells): redone with the value-filing kernels */

struct HuReadDesc {
	int64_t baseOff;           /* offset of the read's codes in the batch's code buffer          */
	int32_t len;
	int32_t nRegions;
	int64_t scratchOff;        /* first cell of the read in the DP scratch                        */
	int64_t traceOff;
	int64_t decOff;            /* first byte of the read in the decision scratch                  */
	int64_t cornerOff;         /* first cell of the read in the corner scratch (k_viterbi_wave)   */
	HuRegion reg[HU_MAX_REGIONS];
};

void hu_set_error(const char* fmt, ...);
/* The exception barrier of the C ABI ("never abort inside the library").  Every extern "C" entry point is a
 * function-try-block whose handler calls this INSIDE catch(...): the exception in flight becomes a status code and a message in the
 * calling thread's hu_last_error() — std::bad_alloc / std::length_error (a size the host cannot back) -> HU_ERR_NOMEM, std::system_error
 * (a thread that cannot start) and anything else -> HU_ERR_STATE.  What it cannot turn into a status is a fault of the GPU itself (an
 * out-of-bounds access of a kernel ends the process from inside the HIP runtime): those are kept out by validating every index a
 * kernel forms from caller data before the launch. */
int hu_catch_all(const char* fn) noexcept;
/* the first k places of std::sort (libstdc++) on packed (key << 24 | index) elements compared on the key alone (hu_host.cpp) */
void hu_sort_prefix_packed(uint64_t* a, size_t n, size_t k);

/* model constants of the table-driven placement kernel (k_place_blk), one buffer of doubles per database */
#define HU_PC_LAM 0              /* [4]      eigenvalues                                                      */
#define HU_PC_RATE 4             /* [16]     dGamma rates (rate[0] = 1 without dGamma)                         */
#define HU_PC_W 20               /* [5][16]  W^b_mn = U_bm U_bn (b < 4), W^4_mn = sum_i pi_i^2 U_im U_in / sum pi^2 */
#define HU_PC_C 100              /* [4][4][4] C_mnk = sum_i pi_i U_im U_in U_ik                                */
#define HU_PC_CB 164             /* [5][4]   c^b = U^-1 e_b (b < 4), c^4 = U^-1 pi                              */
#define HU_PC_S 184              /* [4]      s_k = sum_i pi_i U_ik                                             */
#define HU_PC_COUNT 188
void hu_place_consts(const HuModelDev& m, double* pc);

/* host-side model preparation */
int hu_model_prepare(const hu_model_desc* d, HuModelDev* out);

/* host-side profile preparation (hu_profile.cpp) */
struct HuProfileHost {
	int K = 0, L = 0;
	std::vector<double> EM, EI, T7, entryC, exitC;
	std::vector<int32_t> p2cs, cs2p;
	int init(const hu_profile_desc* d);
};
void hu_mode_costs(int K, int mode, double* tNN, double* tNB, double* tEC, double* tCC);

/* file readers (hu_formats.cpp) */
struct HuTreeHost {
	int32_t n = 0, csLen = 0, root = 0;
	std::vector<int32_t> parent;
	std::vector<double> blen, height, annoDist;
	std::vector<int8_t> seq;
	std::vector<double> up, down;
	std::vector<std::string> names, annos;
	std::vector<int32_t> annoId;
	hu_model_desc model;
};
int hu_read_hmm(const char* path, HuProfileHost& out, std::vector<double>& EM, std::vector<double>& EI,
		std::vector<double>& T, std::vector<int32_t>& p2cs, int& K, int& L);
int hu_read_ptu(const char* path, HuTreeHost& out);
int hu_read_ptu_sink(const char* path, HuTreeHost& out, const std::function<int(bool, int64_t, const double*)>* sink);
int hu_read_hmm_stream(std::istream& in, const char* name, HuProfileHost& out, std::vector<double>& EM, std::vector<double>& EI,
		std::vector<double>& T, std::vector<int32_t>& p2cs, int& K, int& L);
int hu_read_model_text(std::istream& in, hu_model_desc& m);

/* The CPUs this process may really use: the smallest of the hardware's, the affinity mask's and the cgroup's CPU quota (a container that shows
 * 256 CPUs with a quota of 16 stops EVERY thread of the process for the rest of the period once its threads have used the quota — also the ones
 * that feed the GPU), divided by LOCAL_WORLD_SIZE when a launcher runs one rank per GPU.  HU_CPU_BUDGET overrides.  Helper threads of all pools together stay within it: a run asks for helpers and takes what is
 * left (possibly none: the caller always works itself). */
int hu_cpu_budget();
int hu_helpers_acquire(int want);
void hu_helpers_release(int n);

/* work() on up to nt threads, the caller being one of them (work is a self-scheduling loop: fewer helpers only means larger shares).
 * An exception inside a helper is carried to the calling thread and rethrown there once every helper has been joined — a std::thread
 * destroyed while joinable, or an exception leaving a thread body, would end the process; a helper that cannot be started is done without. */
template<class F> void hu_run_threads(unsigned nt, F work) {
	std::exception_ptr first;
	std::mutex m;
	auto guarded = [&] { try { work(); } catch(...) { std::lock_guard<std::mutex> lk(m); if(!first) first = std::current_exception(); } };
	std::vector<std::thread> th;
	const int got = nt > 1 ? hu_helpers_acquire((int) nt - 1) : 0;
	try { th.reserve(got); for(int t = 0; t < got; ++t) th.emplace_back(guarded); } catch(...) { }
	guarded();
	for(auto& t : th) t.join();
	hu_helpers_release(got);
	if(first) std::rethrow_exception(first);
}
/* runs f when the scope is left, by return or by exception */
template<class F> struct HuScope { F f; explicit HuScope(F f) : f(f) {} ~HuScope() { f(); } HuScope(const HuScope&) = delete; HuScope& operator=(const HuScope&) = delete; };
