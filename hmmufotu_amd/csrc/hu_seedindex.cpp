// Host seed lookup: what CSFMIndex::locateOne + buildAlignPath give alignSeq
// (src/HmmUFOtu_main.cpp:50-84, src/CSFMIndex.cpp:121-147, 252-273, src/BandedHMMP7.cpp:894-941).
//
// The reference indexes the concatenation of the gap-free MSA rows (one separator per sequence,
// CSFMIndex::buildConcatSeq src/CSFMIndex.cpp:288-330) with an FM-index — suffix array by libdivsufsort, BWT in
// an RRR wavelet tree, SA sampled every SA_SAMPLE_RATE — searches a seed backwards, and takes ONE hit of the
// suffix-array range: a random one (locateOne, rand() under OpenMP tasks: SURVEY F7) or the first
// (locateFirst, :92-119).  A hit is a start position in the concatenation; concat2CS maps it and its
// last base to CS columns, extractCS rebuilds the gapped CS string between them.
//
// Here the same text is indexed by a PREFIX-SORTED position array: every position whose seed_len-mer lies
// inside one sequence, ordered by the 32 symbols that start there (2 bits per base, zero-padded at the end of
// its sequence), then by position — the suffix array of the reference truncated at depth 32 — plus a
// directory over the first 12 bases.  A lookup is one directory read and a binary search that extracts the
// k-mers it compares from the 2-bit text.  The hit taken is the FIRST of the range, i.e. locateFirst's
// choice whenever the candidates differ within 32 symbols (else the lowest text position): deterministic, and
// a member of the reference's own hit set.  Sequences are taken in node-id order of the leaves (the
// reference: MSA row order; the .ptu keeps the map, the engine does not need it).
// Resident: 4 B (position) + 2 B (CS column) + 0.25 B (text) per residue + 64 MiB of directory: 0.95 GB for the
// 1.4 x 10^8 residues of a gg_97-scale MSA (the hash table this replaces held ~5 GB).
#include <algorithm>
#include <atomic>
#include <thread>
#include <cstring>
#include <string>
#include <vector>
#include "hu_common.h"

#define HU_SX_DIRK 12      /* bases of the directory key */

struct hu_seed_index {
	int seedLen = 0, csLen = 0, K = 0;
	int64_t nRes = 0;
	std::vector<int32_t> cs2p;                       /* getProfileLoc */
	std::vector<uint32_t> seqEnd;                    /* per indexed sequence: one past its last residue */
	std::vector<uint16_t> cols;                      /* concat2CS (0-based here): CS column of every residue */
	std::vector<uint64_t> text;                      /* residues, 2 bits each, 32 per word, first base in the top bits */
	std::vector<uint32_t> sa;                        /* positions in (32-symbol prefix, position) order */
	std::vector<uint32_t> dir;                       /* [4^12 + 1] first sa index of every 12-base prefix */
	int64_t distinct = 0;
};

static int8_t sym_code(char c) {
	switch(c) {
	case 'A': case 'M': case 'R': case 'W': case 'V': case 'H': case 'D': case 'N': return 0;
	case 'C': case 'S': case 'Y': case 'B': return 1;
	case 'G': case 'K': return 2;
	case 'T': case 'U': return 3;
	default: return -1;
	}
}

/* the 32 symbols starting at residue p, first base in the top bits (may run past the sequence: the caller masks) */
static inline uint64_t window32(const std::vector<uint64_t>& text, uint64_t p) {
	const uint64_t w = p >> 5; const unsigned s = (unsigned)(p & 31) * 2;
	const uint64_t hi = text[w] << s;
	return s ? hi | (text[w + 1] >> (64 - s)) : hi;
}
static inline uint64_t keep_top(uint64_t v, int nsym) { return nsym >= 32 ? v : nsym <= 0 ? 0 : v & ~((~0ull) >> (2 * nsym)); }

extern "C" int hu_seed_index_create(int32_t n_nodes, int32_t cs_len, const int32_t* parent, const int8_t* seq,
		int32_t K, const int32_t* p2cs, int32_t seed_len, hu_seed_index** out) {
	if(!parent || !seq || !p2cs || !out || n_nodes < 1 || cs_len < 1 || cs_len > 65535 || seed_len < HU_SX_DIRK || seed_len > 31) {
		hu_set_error("hu_seed_index_create: bad argument (seed length must be in %d..31)", HU_SX_DIRK); return HU_ERR_ARG;
	}
	hu_seed_index* ix = new hu_seed_index;
	ix->seedLen = seed_len; ix->csLen = cs_len; ix->K = K;
	ix->cs2p.assign(cs_len + 2, 0);
	for(int k = 1; k <= K; ++k) if(p2cs[k] >= 1 && p2cs[k] <= cs_len) ix->cs2p[p2cs[k]] = k;
	for(int i = p2cs[K] + 1; i <= cs_len; ++i) ix->cs2p[i] = K;   /* extend_index */
	std::vector<char> hasChild(n_nodes, 0);
	for(int i = 0; i < n_nodes; ++i) if(parent[i] >= 0 && parent[i] < n_nodes) hasChild[parent[i]] = 1;
	/* residues of the leaves, concatenated (the CSFM index holds the MSA = leaf sequences only) */
	int64_t total = 0;
	for(int i = 0; i < n_nodes; ++i) if(!hasChild[i]) { const int8_t* s = seq + (size_t) i * cs_len; for(int c = 0; c < cs_len; ++c) total += s[c] >= 0; }
	if(total >= (1ll << 32) - 64) { delete ix; hu_set_error("hu_seed_index_create: more than 2^32 residues"); return HU_ERR_ARG; }
	ix->nRes = total;
	ix->cols.resize((size_t) total);
	ix->text.assign((size_t)(total / 32 + 3), 0);
	int64_t at = 0;
	for(int i = 0; i < n_nodes; ++i) {
		if(hasChild[i]) continue;
		const int8_t* s = seq + (size_t) i * cs_len;
		for(int c = 0; c < cs_len; ++c) {
			if(s[c] < 0) continue;
			ix->cols[(size_t) at] = (uint16_t) c;
			ix->text[(size_t)(at >> 5)] |= (uint64_t)(s[c] & 3) << (62 - 2 * (int)(at & 31));
			++at;
		}
		ix->seqEnd.push_back((uint32_t) at);
	}
	/* positions whose seed lies inside one sequence, with their masked 32-symbol keys */
	struct Ent { uint64_t key; uint32_t pos; };
	std::vector<Ent> ents;
	ents.reserve((size_t) total);
	{
		uint32_t b = 0;
		for(size_t q = 0; q < ix->seqEnd.size(); ++q) {
			const uint32_t e = ix->seqEnd[q];
			for(uint32_t p = b; p + (uint32_t) seed_len <= e; ++p) ents.push_back(Ent{keep_top(window32(ix->text, p), (int)(e - p)), p});
			b = e;
		}
	}
	/* sort by (key, pos): 256 buckets by the first four bases, sorted in parallel */
	{
		std::vector<size_t> cnt(257, 0);
		for(const Ent& e : ents) cnt[(e.key >> 56) + 1]++;
		for(int i = 0; i < 256; ++i) cnt[i + 1] += cnt[i];
		std::vector<Ent> tmp(ents.size());
		{ std::vector<size_t> w(cnt.begin(), cnt.end() - 1); for(const Ent& e : ents) tmp[w[e.key >> 56]++] = e; }
		ents.swap(tmp);
		tmp.clear(); tmp.shrink_to_fit();
		unsigned nt = std::thread::hardware_concurrency(); if(nt > 16) nt = 16; if(nt < 1) nt = 1;
		std::atomic<int> next{0};
		auto work = [&] { for(;;) { const int b = next.fetch_add(1); if(b >= 256) break;
			std::sort(ents.begin() + cnt[b], ents.begin() + cnt[b + 1], [](const Ent& x, const Ent& y) { return x.key != y.key ? x.key < y.key : x.pos < y.pos; }); } };
		std::vector<std::thread> th;
		for(unsigned t = 1; t < nt; ++t) th.emplace_back(work);
		work();
		for(auto& t : th) t.join();
	}
	ix->sa.resize(ents.size());
	ix->dir.assign(((size_t) 1 << (2 * HU_SX_DIRK)) + 1, 0);
	const uint64_t kmask = ~((~0ull) >> (2 * seed_len));
	uint64_t lastK = ~0ull;
	for(size_t i = 0; i < ents.size(); ++i) {
		ix->sa[i] = ents[i].pos;
		ix->dir[(size_t)(ents[i].key >> (64 - 2 * HU_SX_DIRK)) + 1]++;
		const uint64_t km = ents[i].key & kmask;
		if(km != lastK || i == 0) { ix->distinct++; lastK = km; }
	}
	for(size_t i = 1; i < ix->dir.size(); ++i) ix->dir[i] += ix->dir[i - 1];
	*out = ix;
	return HU_OK;
}
extern "C" void hu_seed_index_destroy(hu_seed_index* ix) { delete ix; }
/* number of distinct seed_len-mers indexed */
extern "C" int64_t hu_seed_index_size(const hu_seed_index* ix) { return ix ? ix->distinct : 0; }
/* resident bytes of the index; positions = number of indexed k-mer starts */
extern "C" int64_t hu_seed_index_bytes(const hu_seed_index* ix, int64_t* positions) {
	if(!ix) return 0;
	if(positions) *positions = (int64_t) ix->sa.size();
	return (int64_t)(ix->sa.size() * 4 + ix->dir.size() * 4 + ix->cols.size() * 2 + ix->text.size() * 8 + ix->seqEnd.size() * 4 + ix->cs2p.size() * 4);
}

/* locateFirst + buildAlignPath for the k-mer read[from0 .. from0+seedLen): returns 1 and fills out6 when
 * the k-mer occurs and yields a valid path, else 0 */
static int lookup_one(const hu_seed_index* ix, const char* read, int from0, int32_t* out6) {
	const int k = ix->seedLen;
	uint64_t key = 0;
	for(int i = 0; i < k; ++i) { const int8_t c = sym_code(read[from0 + i]); if(c < 0) return 0; key |= (uint64_t) c << (62 - 2 * i); }
	const size_t b = (size_t)(key >> (64 - 2 * HU_SX_DIRK));
	size_t lo = ix->dir[b], hi = ix->dir[b + 1];
	while(lo < hi) { /* first entry whose k-mer is >= the query (entries are ordered by their 32-symbol prefix, hence by k-mer) */
		const size_t mid = (lo + hi) >> 1;
		if(keep_top(window32(ix->text, ix->sa[mid]), k) < key) lo = mid + 1; else hi = mid;
	}
	if(lo >= ix->dir[b + 1] || keep_top(window32(ix->text, ix->sa[lo]), k) != key) return 0;
	const uint16_t* cols = &ix->cols[ix->sa[lo]];
	/* CSLoc: 1-based start/end, CS string with '-' wherever the hit sequence has no residue (extractCS); walked
	 * exactly like buildAlignPath does (i over the read, j over CS columns) */
	const int csStart = cols[0] + 1, csEnd = cols[k - 1] + 1;
	if(!(csStart > 0 && csStart < csEnd)) return 0; /* CSLoc::isValid */
	int start = 0, end = 0, from = 0, to = 0, nIns = 0, nDel = 0;
	int i = from0 + 1, r = 0;
	for(int j = csStart; j <= csEnd; ++j) {
		const int kk = ix->cs2p[j];
		const bool nonGap = (cols[r] + 1 == j);
		if(from == 0 && nonGap) from = i;
		if(nonGap) to = i;
		if(kk != 0) { if(start == 0) start = kk; end = kk; if(!nonGap) nDel++; }
		else if(nonGap) nIns++;
		if(nonGap) { ++i; ++r; }
	}
	if(!(start > 0 && start <= end && from > 0 && from <= to)) return 0; /* ViterbiAlignPath::isValid */
	out6[0] = start; out6[1] = end; out6[2] = from; out6[3] = to; out6[4] = nIns; out6[5] = nDel;
	return 1;
}

/* every occurrence of one seed, in index order: (sequence, residue offset in it, first CS column 0-based); for tests of
 * the hit semantics.  Returns the number of occurrences (cap entries written). */
extern "C" int64_t hu_seed_index_occurrences(const hu_seed_index* ix, const char* kmer, int32_t* seq_no, int32_t* offset, int32_t* cs_col, int64_t cap) {
	if(!ix || !kmer) return 0;
	const int k = ix->seedLen;
	uint64_t key = 0;
	for(int i = 0; i < k; ++i) { const int8_t c = sym_code(kmer[i]); if(c < 0) return 0; key |= (uint64_t) c << (62 - 2 * i); }
	const size_t b = (size_t)(key >> (64 - 2 * HU_SX_DIRK));
	int64_t n = 0;
	for(size_t i = ix->dir[b]; i < ix->dir[b + 1]; ++i) {
		const uint32_t p = ix->sa[i];
		const uint64_t km = keep_top(window32(ix->text, p), k);
		if(km < key) continue;
		if(km > key) break;
		if(n < cap) {
			const size_t q = std::upper_bound(ix->seqEnd.begin(), ix->seqEnd.end(), p) - ix->seqEnd.begin();
			if(seq_no) seq_no[n] = (int32_t) q;
			if(offset) offset[n] = (int32_t)(p - (q ? ix->seqEnd[q - 1] : 0));
			if(cs_col) cs_col[n] = ix->cols[p];
		}
		++n;
	}
	return n;
}

/* the two seed scans of alignSeq (src/HmmUFOtu_main.cpp:50-84) for n reads; vpaths [n][2][6] */
extern "C" int hu_seed_index_lookup(const hu_seed_index* ix, int n, const char* bases, const int64_t* offs, int seed_region,
		int align_mode, int32_t* vpaths) {
	if(!ix || n < 0 || (n && (!bases || !offs || !vpaths))) { hu_set_error("hu_seed_index_lookup: bad argument"); return HU_ERR_ARG; }
	const int seedLen = ix->seedLen;
	auto one = [&](int r) {
		const char* read = bases + offs[r];
		const int len = (int)(offs[r + 1] - offs[r]);
		int32_t* vp = vpaths + (size_t) r * 12;
		memset(vp, 0, 12 * sizeof(int32_t));
		int k = 0;
		const int regionLen = seed_region < len ? seed_region : len;
		for(int seedFrom = 0; seedFrom + seedLen - 1 < regionLen; ++seedFrom)
			if(lookup_one(ix, read, seedFrom, vp + 6 * k)) { ++k; break; }
		if(align_mode == HU_MODE_GLOBAL && (k == 0 || len >= 2 * regionLen))
			for(int seedTo = len - 1; seedTo - seedLen + 1 >= len - regionLen && seedTo - seedLen + 1 >= 0; --seedTo)
				if(lookup_one(ix, read, seedTo - seedLen + 1, vp + 6 * k)) { ++k; break; }
	};
	unsigned nt = std::thread::hardware_concurrency();
	if(nt > 16) nt = 16;
	if(n < 1024 || nt <= 1) { for(int r = 0; r < n; ++r) one(r); return HU_OK; }
	std::atomic<int> next{0};
	std::vector<std::thread> th;
	for(unsigned t = 0; t < nt; ++t) th.emplace_back([&] { for(;;) { int a = next.fetch_add(256); if(a >= n) break; int e = std::min(n, a + 256); for(int r = a; r < e; ++r) one(r); } });
	for(auto& t : th) t.join();
	return HU_OK;
}
