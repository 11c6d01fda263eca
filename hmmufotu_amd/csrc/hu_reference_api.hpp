// Per-read adapters with the shape of the reference's free functions (src/HmmUFOtu_main.h:70-113)
// over the batched C ABI: a batch of one.  POD mirrors replace HmmAlignment / PTLoc / PTPlacement
// (node ids instead of shared_ptr<PTUNode>).  Header-only; link with -lhmmufotu_amd.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/hmmufotu_amd.h"

namespace hmmufotu_amd {

struct HmmAlignment { /* src/BandedHMMP7.h:74-130 */
	int K = 0, L = 0, seqStart = 0, seqEnd = 0, hmmStart = 0, hmmEnd = 0, csStart = 0, csEnd = 0;
	double cost = 0;
	std::string align;
	int status = 0;
	bool isValid() const { return status == HU_READ_OK; }
};
struct ViterbiAlignPath { int start, end, from, to, nIns, nDel; }; /* src/BandedHMMP7.h:229-248 */
struct PTLoc { int start, end; long id; double dist; };             /* src/PhyloTreeUnrooted.h:390-405 */
struct PTPlacement {                                                /* src/PhyloTreeUnrooted.h:410-510 */
	int start = 0, end = 0; long cNode = -1, pNode = -1, aNode = -1;
	double wuv = 0, ratio = 0, wnr = 0, loglik = 0, height = 0, qPlace = 0, qTaxon = 0;
};

inline void check(int rc) { if(rc != HU_OK) throw std::runtime_error(hu_last_error()); }

/* One read at a time through one hu_batch: the same call order as the task body of
 * src/hmmufotu.cpp:621-733.  Not the fast path (use hu_assign_batch for throughput). */
class PerRead {
public:
	PerRead(hu_db* db, const hu_opts& o) : db_(db), opts_(o) {
		check(hu_batch_create(db, 1, &b_));
		check(hu_db_info(db, &K_, &L_, nullptr, nullptr, nullptr));
	}
	~PerRead() { hu_batch_destroy(b_); }

	/* alignSeq(hmm, csfm, read, seedLen, seedRegion, mode) after the CSFM lookup */
	HmmAlignment alignSeq(const std::string& read, const std::vector<ViterbiAlignPath>& seeds) {
		int64_t offs[2] = {0, (int64_t) read.size()};
		int32_t vp[12] = {0};
		for(size_t i = 0; i < seeds.size() && i < 2; ++i) { const ViterbiAlignPath& s = seeds[i]; int32_t* r = vp + 6 * i; r[0] = s.start; r[1] = s.end; r[2] = s.from; r[3] = s.to; r[4] = s.nIns; r[5] = s.nDel; }
		check(hu_batch_set_reads(b_, 1, read.data(), offs, vp, nullptr, nullptr, nullptr));
		check(hu_align_batch(b_, &opts_));
		hu_align_rec rec; std::string row(L_, '.');
		check(hu_batch_get_alignments(b_, &rec, &row[0], nullptr, 0));
		HmmAlignment a;
		a.K = K_; a.L = L_; a.seqStart = rec.seq_start; a.seqEnd = rec.seq_end; a.hmmStart = rec.hmm_start; a.hmmEnd = rec.hmm_end;
		a.csStart = rec.cs_start; a.csEnd = rec.cs_end; a.cost = rec.cost; a.align = row; a.status = rec.status;
		return a;
	}
	/* getSeed(ptu, seq, start, end, maxDiff, maxHeight) + truncation to max_nseed */
	std::vector<PTLoc> getSeed() {
		check(hu_seed_batch(b_, &opts_));
		int32_t n = 0, ids[64], d[64], N[64];
		check(hu_batch_get_seeds(b_, &n, ids, d, N));
		hu_align_rec rec; check(hu_batch_get_alignments(b_, &rec, nullptr, nullptr, 0));
		std::vector<PTLoc> locs;
		for(int i = 0; i < n; ++i) locs.push_back(PTLoc{rec.cs_start - 1, rec.cs_end - 1, ids[i], (double) d[i] / N[i]});
		return locs;
	}
	/* estimateSeq + filterPlacements + placeSeq + calcQValues; returns the best placement like
	 * `bestPlace = places[0]` (src/hmmufotu.cpp:733) */
	PTPlacement place() {
		check(hu_estimate_batch(b_, &opts_));
		check(hu_filter_batch(b_, &opts_));
		check(hu_place_batch(b_, &opts_));
		check(hu_finish_batch(b_, &opts_));
		hu_place_rec r; check(hu_batch_get_placements(b_, &r));
		hu_align_rec rec; check(hu_batch_get_alignments(b_, &rec, nullptr, nullptr, 0));
		PTPlacement p;
		p.start = rec.cs_start - 1; p.end = rec.cs_end - 1; p.cNode = r.c_node; p.pNode = r.p_node; p.aNode = r.a_node;
		p.wuv = r.wuv; p.ratio = r.ratio; p.wnr = r.wnr; p.loglik = r.loglik; p.height = r.height; p.qPlace = r.q_place; p.qTaxon = r.q_taxon;
		return p;
	}
private:
	hu_db* db_; hu_batch* b_ = nullptr; hu_opts opts_; int32_t K_ = 0, L_ = 0;
};

} // namespace hmmufotu_amd
