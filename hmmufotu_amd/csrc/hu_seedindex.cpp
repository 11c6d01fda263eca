// Host seed lookup: what CSFMIndex::locateOne + buildAlignPath give alignSeq
// (src/HmmUFOtu_main.cpp:50-84, src/CSFMIndex.cpp:121-147,262-273, src/BandedHMMP7.cpp:894-941),
// served by a plain hash index over the k-mers of the leaf sequences instead of the reference's
// RRR-wavelet-tree FM-index (libcds).  Two deliberate differences: the k-mer length is fixed when
// the index is built, and among several occurrences the FIRST (lowest leaf id, lowest position)
// is taken where the reference draws one with rand() — deterministic output.
#include <algorithm>
#include <atomic>
#include <thread>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "hu_common.h"

struct hu_seed_index {
	int seedLen = 0, csLen = 0, K = 0;
	std::vector<int32_t> cs2p;                       /* getProfileLoc */
	std::vector<int32_t> leafStart;                  /* per indexed leaf: offset into cols/codes */
	std::vector<uint16_t> cols;                      /* 0-based CS column of every residue, leaves concatenated */
	std::vector<int8_t> codes;
	std::unordered_map<uint64_t, uint64_t> first;    /* packed k-mer -> (leaf << 32 | pos), first occurrence */
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

extern "C" int hu_seed_index_create(int32_t n_nodes, int32_t cs_len, const int32_t* parent, const int8_t* seq,
		int32_t K, const int32_t* p2cs, int32_t seed_len, hu_seed_index** out) {
	if(!parent || !seq || !p2cs || !out || n_nodes < 1 || cs_len < 1 || cs_len > 65535 || seed_len < 8 || seed_len > 31) {
		hu_set_error("hu_seed_index_create: bad argument"); return HU_ERR_ARG;
	}
	hu_seed_index* ix = new hu_seed_index;
	ix->seedLen = seed_len; ix->csLen = cs_len; ix->K = K;
	ix->cs2p.assign(cs_len + 2, 0);
	for(int k = 1; k <= K; ++k) if(p2cs[k] >= 1 && p2cs[k] <= cs_len) ix->cs2p[p2cs[k]] = k;
	for(int i = p2cs[K] + 1; i <= cs_len; ++i) ix->cs2p[i] = K;   /* extend_index */
	std::vector<char> hasChild(n_nodes, 0);
	for(int i = 0; i < n_nodes; ++i) if(parent[i] >= 0 && parent[i] < n_nodes) hasChild[parent[i]] = 1;
	const uint64_t mask = (seed_len == 32) ? ~0ull : ((1ull << (2 * seed_len)) - 1);
	for(int i = 0; i < n_nodes; ++i) {
		if(hasChild[i]) continue; /* the CSFM index holds the MSA (= leaf) sequences only */
		const int8_t* s = seq + (size_t) i * cs_len;
		const int32_t leaf = (int32_t) ix->leafStart.size();
		ix->leafStart.push_back((int32_t) ix->cols.size());
		uint64_t key = 0; int run = 0; int pos = 0;
		for(int c = 0; c < cs_len; ++c) {
			if(s[c] < 0) continue;
			ix->cols.push_back((uint16_t) c); ix->codes.push_back(s[c]);
			key = ((key << 2) | (uint64_t) s[c]) & mask;
			if(++run >= seed_len) ix->first.emplace(key, ((uint64_t) leaf << 32) | (uint32_t)(pos - seed_len + 1));
			++pos;
		}
	}
	ix->leafStart.push_back((int32_t) ix->cols.size());
	*out = ix;
	return HU_OK;
}
extern "C" void hu_seed_index_destroy(hu_seed_index* ix) { delete ix; }
extern "C" int64_t hu_seed_index_size(const hu_seed_index* ix) { return ix ? (int64_t) ix->first.size() : 0; }

/* locateOne + buildAlignPath for the k-mer read[from0 .. from0+seedLen): returns 1 and fills out6 when
 * the k-mer occurs and yields a valid path, else 0 */
static int lookup_one(const hu_seed_index* ix, const char* read, int from0, int32_t* out6) {
	uint64_t key = 0;
	for(int i = 0; i < ix->seedLen; ++i) { const int8_t c = sym_code(read[from0 + i]); if(c < 0) return 0; key = (key << 2) | (uint64_t) c; }
	auto it = ix->first.find(key);
	if(it == ix->first.end()) return 0;
	const int32_t leaf = (int32_t)(it->second >> 32), pos = (int32_t)(it->second & 0xffffffffu);
	const uint16_t* cols = &ix->cols[ix->leafStart[leaf] + pos];
	/* CSLoc: 1-based start/end, CS string with '-' wherever the hit sequence has no residue; walked
	 * exactly like buildAlignPath does (i over the read, j over CS columns) */
	const int csStart = cols[0] + 1, csEnd = cols[ix->seedLen - 1] + 1;
	if(!(csStart > 0 && csStart < csEnd)) return 0; /* CSLoc::isValid */
	int start = 0, end = 0, from = 0, to = 0, nIns = 0, nDel = 0;
	int i = from0 + 1, r = 0;
	for(int j = csStart; j <= csEnd; ++j) {
		const int k = ix->cs2p[j];
		const bool nonGap = (cols[r] + 1 == j);
		if(from == 0 && nonGap) from = i;
		if(nonGap) to = i;
		if(k != 0) { if(start == 0) start = k; end = k; if(!nonGap) nDel++; }
		else if(nonGap) nIns++;
		if(nonGap) { ++i; ++r; }
	}
	if(!(start > 0 && start <= end && from > 0 && from <= to)) return 0; /* ViterbiAlignPath::isValid */
	out6[0] = start; out6[1] = end; out6[2] = from; out6[3] = to; out6[4] = nIns; out6[5] = nDel;
	return 1;
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
