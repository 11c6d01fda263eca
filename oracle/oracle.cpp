// ORACLE — TEST INFRASTRUCTURE ONLY.  C ABI over the CPU restatement (ctypes-friendly).
// See oracle_models.h for the scope note.  PARITY UNPINNED for everything floating-point; the integer rows a8
// (DigitalSeq encoding) and a9 (SeqUtils::pDist) are checked against the reference's own code where it compiles without Eigen3 /
// Boost (oracle/_ref/libref_seq.so, tests/test_ref_seq.py), row f2's file format against its vendored libcds (oracle/csfm_ref.cpp).
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <omp.h>
#include "oracle_models.h"
#include "oracle_hmm.h"
#include "oracle_phylo.h"

using namespace orc;

extern "C" {

/* ---------------- models ---------------- */
void* orc_model_new(int type, const double* pi, const double* par) { return new Model(make_model(type, pi, par)); }
void orc_model_free(void* m) { delete (Model*) m; }
void orc_model_pr(void* m, double t, double* P) { ((Model*) m)->Pr(t, P); }
void orc_model_get(void* m, double* pi4, double* Q16) {
	Model* M = (Model*) m;
	for(int i = 0; i < 4; ++i) pi4[i] = M->pi[i];
	if(Q16) for(int i = 0; i < 16; ++i) Q16[i] = M->Q[i];
}

/* ---------------- profile HMM ---------------- */
struct HmmHandle { Hmm h; std::vector<VitWork> work; };

void* orc_hmm_new(int K, int L, const double* EM, const double* EI, const double* T, const int* p2cs, int mode) {
	HmmHandle* H = new HmmHandle;
	H->h.init(K, L, EM, EI, T, p2cs);
	H->h.setMode(mode);
	H->work.resize(omp_get_max_threads() > 0 ? omp_get_max_threads() : 1);
	return H;
}
void orc_hmm_free(void* h) { delete (HmmHandle*) h; }
void orc_hmm_set_mode(void* h, int mode) { ((HmmHandle*) h)->h.setMode(mode); }
void orc_hmm_get(void* h, double* entryC, double* exitC, double* tsp4) {
	Hmm& H = ((HmmHandle*) h)->h;
	for(int k = 0; k <= H.K; ++k) { entryC[k] = H.entryC[k]; exitC[k] = H.exitC[k]; }
	tsp4[0] = H.T_NN; tsp4[1] = H.T_NB; tsp4[2] = H.T_EC; tsp4[3] = H.T_CC;
}
void orc_build_align_path(void* h, int locStart, int locEnd, const char* CS, int csFrom, int csTo, int* out6) {
	VPath v = ((HmmHandle*) h)->h.buildAlignPath(locStart, locEnd, std::string(CS), csFrom, csTo);
	out6[0] = v.start; out6[1] = v.end; out6[2] = v.from; out6[3] = v.to; out6[4] = v.nIns; out6[5] = v.nDel;
}
static std::vector<VPath> to_vpaths(const int* vp, int nvp) {
	std::vector<VPath> v;
	for(int i = 0; i < nvp; ++i) {
		VPath p{vp[i*6], vp[i*6+1], vp[i*6+2], vp[i*6+3], vp[i*6+4], vp[i*6+5]};
		if(p.isValid()) v.push_back(p);
	}
	return v;
}
static void export_aln(const HmmAlignment& a, int* ints, double* cost, char* align, char* trace, int traceCap) {
	ints[0] = a.seqStart; ints[1] = a.seqEnd; ints[2] = a.hmmStart; ints[3] = a.hmmEnd;
	ints[4] = a.csStart; ints[5] = a.csEnd; ints[6] = a.usedFull ? 1 : 0; ints[7] = a.isValid() ? 1 : 0;
	*cost = a.cost;
	if(align && (int) a.align.size() == a.L) std::memcpy(align, a.align.data(), a.L);
	if(trace && traceCap > 0) { int n = std::min((int) a.trace.size(), traceCap - 1); std::memcpy(trace, a.trace.data(), n); trace[n] = 0; }
}
/* alignSeq after the seed lookup; returns 1 if a valid alignment was produced */
int orc_align(void* h, const char* read, int n, const int* vpaths, int nvp, int* ints, double* cost, char* align, char* trace, int traceCap) {
	HmmHandle* H = (HmmHandle*) h;
	for(int i = 0; i < n; ++i) if(ABC.encode(read[i]) < 0) { for(int k = 0; k < 8; ++k) ints[k] = 0; *cost = INF; return 0; }
	HmmAlignment a = alignSeq(H->h, H->work[0], std::string(read, n), to_vpaths(vpaths, nvp));
	export_aln(a, ints, cost, align, trace, traceCap);
	return a.isValid() ? 1 : 0;
}
/* DigitalSeq(abc, name, str) (src/DigitalSeq.cpp:41-48): upper-case, drop invalid chars */
int orc_digitize(const char* s, int n, int8_t* out) {
	int m = 0;
	for(int i = 0; i < n; ++i) { char c = (char) ::toupper(s[i]); int8_t b = ABC.encode(c); if(b != -1) out[m++] = b; }
	return m;
}
/* PE merge (src/hmmufotu.cpp:629-639): returns 0 and leaves fwd untouched on bad orientation */
int orc_merge(int L, int* intsF, double* costF, char* alignF, const int* intsR, const double* costR, const char* alignR) {
	if(!(intsF[4] <= intsR[4] && intsF[5] <= intsR[5])) return 0;
	HmmAlignment a, b;
	a.K = b.K = 0; a.L = b.L = L;
	a.seqStart = intsF[0]; a.seqEnd = intsF[1]; a.hmmStart = intsF[2]; a.hmmEnd = intsF[3]; a.csStart = intsF[4]; a.csEnd = intsF[5]; a.cost = *costF; a.align.assign(alignF, L);
	b.seqStart = intsR[0]; b.seqEnd = intsR[1]; b.hmmStart = intsR[2]; b.hmmEnd = intsR[3]; b.csStart = intsR[4]; b.csEnd = intsR[5]; b.cost = *costR; b.align.assign(alignR, L);
	a.merge(b);
	intsF[0] = a.seqStart; intsF[1] = a.seqEnd; intsF[2] = a.hmmStart; intsF[3] = a.hmmEnd; intsF[4] = a.csStart; intsF[5] = a.csEnd; *costF = a.cost;
	std::memcpy(alignF, a.align.data(), L);
	return 1;
}

/* ---------------- tree ---------------- */
void* orc_tree_new(int nNodes, int csLen, const int* parent, const double* blen, const int8_t* seq,
		const double* up, const double* down, const double* height, const int* annoId,
		void* model, int dgK, const double* dgR, long winStart, long winLen) {
	Tree* t = new Tree;
	t->nNodes = nNodes; t->csLen = csLen; t->parent = parent; t->blen = blen; t->seq = seq;
	t->up = up; t->down = down; t->height = height; t->annoId = annoId;
	t->model = *(Model*) model; t->dgK = dgK;
	for(int k = 0; k < dgK && k < 16; ++k) t->dgR[k] = dgR[k];
	t->winStart = winStart; t->winLen = winLen > 0 ? winLen : csLen;
	t->root = 0;
	for(int i = 0; i < nNodes; ++i) if(parent[i] < 0) t->root = i;
	return t;
}
void orc_tree_free(void* t) { delete (Tree*) t; }
/* up / down hold the message rows of SOME nodes only: rowOf[node] = row, or -1 (a tree of 4 x 10^5 nodes holds 2 x 10^11 bytes of
 * messages; the per-read task reads those of its <= 50 seed nodes).  NULL: row = node */
void orc_tree_set_rows(void* t, const int* rowOf) { ((Tree*) t)->rowOf = rowOf; }

int orc_get_seed(void* tr, const int8_t* seq, int start, int end, double maxDiff, double maxHeight, int tieMode, int maxNSeed,
		long* ids, long* d, long* N, double* dist) {
	std::vector<PTLoc> s = getSeed(*(Tree*) tr, seq, start, end, maxDiff, maxHeight, tieMode, (size_t) maxNSeed);
	for(size_t i = 0; i < s.size(); ++i) { ids[i] = s[i].id; d[i] = s[i].d; N[i] = s[i].N; dist[i] = s[i].dist; }
	return (int) s.size();
}
void orc_pdist_all(void* tr, const int8_t* seq, int start, int end, long* d, long* N) {
	Tree* t = (Tree*) tr;
	for(int i = 0; i < t->nNodes; ++i) pdist_counts(t->S(i), seq, start, end, d[i], N[i]);
}
/* out: ratio, wnr, loglik, wuv; nodes: cNode, pNode, aNode */
void orc_estimate(void* tr, const int8_t* seq, int start, int end, long id, double dist, int weighted, double* out, int* nodes) {
	PTLoc l; l.start = start; l.end = end; l.id = id; l.dist = dist; l.d = l.N = 0;
	Placement p = estimateSeq(*(Tree*) tr, seq, l, weighted != 0);
	out[0] = p.ratio; out[1] = p.wnr; out[2] = p.loglik; out[3] = p.wuv;
	nodes[0] = p.cNode; nodes[1] = p.pNode; nodes[2] = p.aNode;
}
/* out: loglik, wnr, ratio, height; returns outer iteration count */
int orc_place(void* tr, const int8_t* seq, int start, int end, int cNode, double ratio0, double wnr0, double maxHeight, double* out, int* aNode) {
	Tree* t = (Tree*) tr;
	Placement p; p.start = start; p.end = end; p.cNode = cNode; p.pNode = t->parent[cNode]; p.ratio = ratio0; p.wnr = wnr0; p.wuv = t->blen[cNode];
	placeSeq(*t, seq, p, maxHeight);
	out[0] = p.loglik; out[1] = p.wnr; out[2] = p.ratio; out[3] = p.height; *aNode = p.aNode;
	return p.iters;
}

struct OrcOpts { double maxDiff, maxHeight, maxError; int maxNSeed, weighted, onlyML, prior, tieMode, fixRootLoglik; double tieTol; };
static AssignOpts to_opts(const OrcOpts* o) {
	AssignOpts a; a.maxDiff = o->maxDiff; a.maxHeight = o->maxHeight; a.maxError = o->maxError; a.maxNSeed = o->maxNSeed;
	a.weighted = o->weighted; a.onlyML = o->onlyML; a.prior = o->prior; a.tieMode = o->tieMode; a.fixRootLoglik = o->fixRootLoglik; a.tieTol = o->tieTol; return a;
}
static void export_place(const Placement& p, int* ni, double* nd) {
	ni[0] = p.cNode; ni[1] = p.pNode; ni[2] = p.aNode; ni[3] = p.iters;
	nd[0] = p.ratio; nd[1] = p.wnr; nd[2] = p.loglik; nd[3] = p.height; nd[4] = p.qPlace; nd[5] = p.qTaxon; nd[6] = p.annoDist(); nd[7] = p.estLoglik;
}
/* full SEP for one aligned read; outputs all final placements in output order (ints [n][4], dbl [n][8]),
 * seeds (ids/d/N) and per-seed estimates (est [nSeeds][3] = ratio,wnr,loglik).  Returns #placements. */
int orc_assign(void* tr, const int8_t* seq, int start, int end, const OrcOpts* o,
		int* ni, double* nd, int* nSeeds, long* seedIds, long* seedD, long* seedN, double* est, int* filtOrder) {
	std::vector<PTLoc> seeds; std::vector<Placement> ests; std::vector<int> filt;
	std::vector<Placement> pl = assignSeq(*(Tree*) tr, seq, start, end, to_opts(o), &seeds, &ests, &filt);
	if(filtOrder) for(size_t i = 0; i < filt.size(); ++i) filtOrder[i] = filt[i];
	for(size_t i = 0; i < pl.size(); ++i) export_place(pl[i], ni + 4 * i, nd + 8 * i);
	if(nSeeds) *nSeeds = (int) seeds.size();
	for(size_t i = 0; i < seeds.size(); ++i) {
		if(seedIds) { seedIds[i] = seeds[i].id; seedD[i] = seeds[i].d; seedN[i] = seeds[i].N; }
		if(est) { est[3*i] = ests[i].ratio; est[3*i+1] = ests[i].wnr; est[3*i+2] = ests[i].loglik; }
	}
	return (int) pl.size();
}

/* chimera check of one aligned read (src/hmmufotu.cpp:653-691).  nSeedIn < 0: seeds from getSeed as the per-read task does;
 * otherwise the given node ids are the common seeds.  oi[16] = checked, isChimera, seg5 (c,p,a,start,end), seg3 (c,p,a,start,end),
 * pooled counts n5,n3, spare; od[12] = lod, seg5 (ratio,wnr,loglik,estLoglik), seg3 (same), alt5 loglik, alt3 loglik, spare */
int orc_chimera(void* tr, const int8_t* seq, int start, int end, const OrcOpts* o, int numSeg, double maxChimeraError, double minChimeraLod,
		int nSeedIn, const long* seedIn, int* oi, double* od) {
	const Tree& t = *(Tree*) tr;
	AssignOpts a = to_opts(o);
	std::vector<PTLoc> seeds;
	if(nSeedIn < 0) seeds = getSeed(t, seq, start, end, a.maxDiff, a.maxHeight, a.tieMode, (size_t) a.maxNSeed);
	else for(int i = 0; i < nSeedIn; ++i) { PTLoc l; l.start = start; l.end = end; l.id = seedIn[i]; l.d = l.N = 0; l.dist = 0; seeds.push_back(l); }
	ChimeraResult r = chimeraCheck(t, seq, start, end, seeds, a, numSeg, maxChimeraError, minChimeraLod);
	for(int i = 0; i < 16; ++i) oi[i] = -1;
	for(int i = 0; i < 12; ++i) od[i] = NAN;
	oi[0] = r.checked; oi[1] = r.isChimera;
	if(!r.checked) return 0;
	const Placement* ps[2] = { &r.seg5, &r.seg3 };
	for(int k = 0; k < 2; ++k) {
		oi[2 + 5 * k] = ps[k]->cNode; oi[3 + 5 * k] = ps[k]->pNode; oi[4 + 5 * k] = ps[k]->aNode; oi[5 + 5 * k] = ps[k]->start; oi[6 + 5 * k] = ps[k]->end;
		od[1 + 4 * k] = ps[k]->ratio; od[2 + 4 * k] = ps[k]->wnr; od[3 + 4 * k] = ps[k]->loglik; od[4 + 4 * k] = ps[k]->estLoglik;
	}
	oi[12] = (int) r.n5; oi[13] = (int) r.n3;
	od[0] = r.lod; od[9] = r.alt5.loglik; od[10] = r.alt3.loglik;
	return 1;
}

void orc_tree_evaluate(int nNodes, int csLen, const int* parent, const double* blen, int8_t* seq,
		void* model, int dgK, const double* dgR, double* up, double* down, double* rootMsg, double* height) {
	treeEvaluate(nNodes, csLen, parent, blen, seq, *(Model*) model, dgK, dgR, up, down, rootMsg, height);
}

/* ---------------- whole per-read task, batched with OpenMP (CPU baseline) ----------------
 * reads: concatenated chars, offs[nReads+1]; mates optional (already reverse-complemented);
 * vpaths [nReads][2][6] (invalid rows skipped), mvpaths likewise.  Per read outputs:
 * ai[r][8] (alignment ints as orc_align + status in [7]: 1 ok, 0 invalid, 2 chimera-by-orientation),
 * cost[r], align (optional, [r][L]), best placement bi[r][4], bd[r][8], nCand[r].
 * stageSec[4]: summed wall seconds over threads for align / seed / estimate / place.  */
void orc_pipeline_batch(void* hmm, void* tr, int nReads, const char* reads, const long* offs,
		const char* mates, const long* moffs, const int* vpaths, const int* mvpaths,
		const OrcOpts* o, int nThreads,
		int* ai, double* cost, char* alignOut, int* bi, double* bd, int* nCand, double* stageSec,
		int* candNode /* optional [nReads][64]: candidates in filterPlacements order */, double* candEst /* their estimated logliks */,
		double* candRatio0 /* their estimated ratios */, int* bestPos /* position of the final pick in that order */,
		double* candPlaced /* optional [nReads][64][3]: placed ratio, wnr, height of each candidate, same order */,
		int* candIters /* optional [nReads][64][2]: outer iterations, EM passes */,
		int mode /* 0: the whole task; 1: stop after getSeed (seedCnt / seedIds are outputs); 2: seeds GIVEN (seedCnt / seedIds are inputs, in
		          * the order estimateSeq is to see them): what runs when the tree holds message rows of the seed nodes only (orc_tree_set_rows) */,
		int* seedCnt /* [nReads] */, int* seedIds /* [nReads][64] */,
		int* libIds /* mode 1, optional [nReads][64]: the same scan ordered by the reference's literal std::sort on dist alone (TIE_LIBSTDCXX) */,
		int* tieInfo /* mode 1, optional [nReads][4]: nodes of the whole tree at the stable list's cut-off distance, of them inside the list,
		              * 1 when both lists end at the same distance, 1 when a NaN dist made std::sort undefined (lists equal by fallback) */,
		double* extraSec /* mode 1, optional: thread-seconds spent on the libstdc++ order (not part of the task; subtract from the wall) */) {
	HmmHandle* H = (HmmHandle*) hmm; Tree* t = (Tree*) tr;
	const int L = H->h.L;
	AssignOpts opts = to_opts(o);
	if(nThreads <= 0) nThreads = omp_get_max_threads();
	if((int) H->work.size() < nThreads) H->work.resize(nThreads);
	double acc[4] = {0, 0, 0, 0}, accExtra = 0;
	#pragma omp parallel num_threads(nThreads)
	{
		double loc[4] = {0, 0, 0, 0}, locExtra = 0;
		VitWork& w = H->work[omp_get_thread_num()];
		std::vector<int8_t> dseq(L);
		#pragma omp for schedule(dynamic, 4)
		for(int r = 0; r < nReads; ++r) {
			auto t0 = std::chrono::steady_clock::now();
			int* a8 = ai + 8 * r;
			for(int k = 0; k < 4; ++k) bi[4*r+k] = -1;
			for(int k = 0; k < 8; ++k) bd[8*r+k] = NAN;
			nCand[r] = 0;
			std::string rd(reads + offs[r], offs[r+1] - offs[r]);
			bool bad = false;
			for(char c : rd) if(ABC.encode(c) < 0) bad = true;
			HmmAlignment aln;
			if(!bad) aln = alignSeq(H->h, w, rd, to_vpaths(vpaths + 12 * r, 2));
			int status = (!bad && aln.isValid()) ? 1 : 0;
			if(status == 1 && mates) {
				std::string md(mates + moffs[r], moffs[r+1] - moffs[r]);
				bool mbad = false;
				for(char c : md) if(ABC.encode(c) < 0) mbad = true;
				HmmAlignment ra;
				if(!mbad) ra = alignSeq(H->h, w, md, to_vpaths(mvpaths + 12 * r, 2));
				if(mbad || !ra.isValid()) status = 0;
				else if(!(aln.csStart <= ra.csStart && aln.csEnd <= ra.csEnd)) status = 2;
				else aln.merge(ra);
			}
			double c = aln.cost;
			export_aln(aln, a8, &c, alignOut ? alignOut + (size_t) r * L : nullptr, nullptr, 0);
			a8[7] = status; cost[r] = c;
			auto t1 = std::chrono::steady_clock::now();
			loc[0] += std::chrono::duration<double>(t1 - t0).count();
			if(status != 1) continue;
			int m = orc_digitize(aln.align.data(), L, dseq.data());
			if(m != L) { a8[7] = 0; continue; }
			const int start = aln.csStart - 1, end = aln.csEnd - 1;
			std::vector<PTLoc> seeds, scan;
			bool scanNaN = false;
			if(mode == 2) { /* the given nodes, in the given order; their distance as getSeed measures it */
				for(int k = 0; k < seedCnt[r] && k < 64; ++k) {
					PTLoc l; l.start = start; l.end = end; l.id = seedIds[64 * (size_t) r + k];
					pdist_counts(t->S((int) l.id), dseq.data(), start, end, l.d, l.N);
					l.dist = static_cast<double>(l.d) / l.N;
					seeds.push_back(l);
				}
			}
			else if(mode == 1 && libIds) { /* getSeed in its two halves: the scan is shared with the libstdc++ order below */
				scan = scanSeeds(*t, dseq.data(), start, end, opts.maxHeight, scanNaN);
				seeds = orderSeeds(scan, scanNaN, opts.maxDiff, opts.tieMode, (size_t) opts.maxNSeed);
			}
			else seeds = getSeed(*t, dseq.data(), start, end, opts.maxDiff, opts.maxHeight, opts.tieMode, (size_t) opts.maxNSeed);
			auto t2 = std::chrono::steady_clock::now();
			loc[1] += std::chrono::duration<double>(t2 - t1).count();
			if(mode == 1) {
				seedCnt[r] = (int) std::min<size_t>(seeds.size(), 64);
				for(int k = 0; k < seedCnt[r]; ++k) seedIds[64 * (size_t) r + k] = (int) seeds[k].id;
				if(libIds) { /* the reference's own order of the same scan: src/HmmUFOtu_main.cpp:139, src/hmmufotu.cpp:646-647 */
					std::vector<PTLoc> lib = orderSeeds(scan, scanNaN, opts.maxDiff, TIE_LIBSTDCXX, (size_t) opts.maxNSeed);
					for(size_t k = 0; k < lib.size() && k < 64; ++k) libIds[64 * (size_t) r + k] = (int) lib[k].id;
					if(tieInfo && !seeds.empty()) {
						const double cut = seeds.back().dist;
						int tied = 0, tiedIn = 0;
						for(const PTLoc& l : scan) if(l.dist == cut) tied++;
						for(const PTLoc& l : seeds) if(l.dist == cut) tiedIn++;
						int* ti = tieInfo + 4 * (size_t) r;
						ti[0] = tied; ti[1] = tiedIn; ti[2] = !lib.empty() && lib.back().dist == cut; ti[3] = scanNaN;
					}
					locExtra += std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count();
				}
				continue;
			}
			std::vector<Placement> places;
			for(const PTLoc& l : seeds) places.push_back(estimateSeq(*t, dseq.data(), l, opts.weighted != 0, opts.tieTol));
			filterPlacements(places, opts.maxError);
			if(candNode) for(size_t k = 0; k < places.size() && k < 64; ++k) {
				candNode[64 * (size_t) r + k] = places[k].cNode;
				if(candEst) candEst[64 * (size_t) r + k] = places[k].estLoglik;
				if(candRatio0) candRatio0[64 * (size_t) r + k] = places[k].ratio;
			}
			auto t3 = std::chrono::steady_clock::now();
			loc[2] += std::chrono::duration<double>(t3 - t2).count();
			for(Placement& p : places) placeSeq(*t, dseq.data(), p, opts.maxHeight, opts.fixRootLoglik != 0);
			for(size_t k = 0; k < places.size() && k < 64; ++k) { /* still in filterPlacements order */
				if(candPlaced) { double* c = candPlaced + (64 * (size_t) r + k) * 3; c[0] = places[k].ratio; c[1] = places[k].wnr; c[2] = places[k].height; }
				if(candIters) { int* c = candIters + (64 * (size_t) r + k) * 2; c[0] = places[k].iters; c[1] = places[k].emIters; }
			}
			if(opts.onlyML)
				std::sort(places.rbegin(), places.rend(), [](const Placement& l, const Placement& r) { return l.loglik < r.loglik; });
			else {
				calcQValues(*t, places, opts.prior);
				std::sort(places.rbegin(), places.rend(), [](const Placement& l, const Placement& r) { return l.qPlace < r.qPlace; });
			}
			auto t4 = std::chrono::steady_clock::now();
			loc[3] += std::chrono::duration<double>(t4 - t3).count();
			nCand[r] = (int) places.size();
			if(!places.empty()) export_place(places[0], bi + 4 * r, bd + 8 * r);
			if(bestPos) {
				bestPos[r] = -1;
				if(candNode) for(size_t k = 0; k < places.size() && k < 64; ++k) if(candNode[64 * (size_t) r + k] == places[0].cNode) { bestPos[r] = (int) k; break; }
			}
		}
		#pragma omp critical
		{ for(int k = 0; k < 4; ++k) acc[k] += loc[k]; accExtra += locExtra; }
	}
	if(stageSec) for(int k = 0; k < 4; ++k) stageSec[k] = acc[k];
	if(extraSec) *extraSec = accExtra;
}


/* literal std::sort of n PTLocs (id = index) on dist alone, first k ids: the checker of the product's restatement of libstdc++'s introsort
 * (hu_sort_prefix_libstdcxx, HU_SEED_ORDER_LIBSTDCXX) */
void orc_std_sort_prefix(const double* dist, long n, long k, int* outIdx) {
	std::vector<PTLoc> locs((size_t) n);
	for(long i = 0; i < n; ++i) { locs[i].start = locs[i].end = 0; locs[i].id = i; locs[i].dist = dist[i]; locs[i].d = locs[i].N = 0; }
	std::sort(locs.begin(), locs.end());
	for(long i = 0; i < k && i < n; ++i) outIdx[i] = (int) locs[i].id;
}

/* McIlroy's adversary ("A Killer Adversary for Quicksort", 1999) run against std::sort itself: values are decided while the sort compares
 * them, so that every pivot turns out to be among the smallest of its range.  The values it ends with, given again as plain numbers, walk
 * std::sort through the same comparisons: partitions of depth > 2 lg n, i.e. introsort's heap-sort branch — the input of the test of that
 * branch in the product's restatement. */
void orc_antiqsort(long n, double* out) {
	std::vector<long> val((size_t) n); const long gas = n - 1; long nsolid = 0, candidate = 0;
	for(long i = 0; i < n; ++i) val[i] = gas;
	std::vector<long> ptr((size_t) n);
	for(long i = 0; i < n; ++i) ptr[i] = i;
	auto cmp = [&](long x, long y) {
		if(val[x] == gas && val[y] == gas) { if(x == candidate) val[x] = nsolid++; else val[y] = nsolid++; }
		if(val[x] == gas) candidate = x; else if(val[y] == gas) candidate = y;
		return val[x] < val[y];
	};
	std::sort(ptr.begin(), ptr.end(), cmp);
	for(long i = 0; i < n; ++i) out[i] = (double) val[i] / (double) n;
}

/* literal std::sort(rbegin, rend, less-by-key) of n records (key, index): what filterPlacements and the final sort of the per-read task call
 * (src/HmmUFOtu_main.cpp:164, src/hmmufotu.cpp:726, 730) — the checker of the device routine that replaced those calls (hu_sort_desc64) */
void orc_std_sort_desc(const double* keys, int n, int* order) {
	struct Rec { double key; int idx; };
	std::vector<Rec> v((size_t) n);
	for(int i = 0; i < n; ++i) { v[i].key = keys[i]; v[i].idx = i; }
	std::sort(v.rbegin(), v.rend(), [](const Rec& l, const Rec& r) { return l.key < r.key; });
	for(int i = 0; i < n; ++i) order[i] = v[i].idx;
}

int orc_max_threads(void) { return omp_get_max_threads(); }

} // extern "C"
