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
#include <functional>
#include <cstdio>
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

/* directory over the first 12 bases and the number of distinct seeds, from `sa` in its final order */
static void finish_index(hu_seed_index* ix) {
	ix->dir.assign(((size_t) 1 << (2 * HU_SX_DIRK)) + 1, 0);
	uint64_t lastK = ~0ull;
	ix->distinct = 0;
	for(size_t i = 0; i < ix->sa.size(); ++i) {
		const uint64_t km = keep_top(window32(ix->text, ix->sa[i]), ix->seedLen);
		ix->dir[(size_t)(km >> (64 - 2 * HU_SX_DIRK)) + 1]++;
		if(km != lastK || i == 0) { ix->distinct++; lastK = km; }
	}
	for(size_t i = 1; i < ix->dir.size(); ++i) ix->dir[i] += ix->dir[i - 1];
}

extern "C" int hu_seed_index_create(int32_t n_nodes, int32_t cs_len, const int32_t* parent, const int8_t* seq,
		int32_t K, const int32_t* p2cs, int32_t seed_len, hu_seed_index** out) try {
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
		hu_run_threads(nt, work);
	}
	ix->sa.resize(ents.size());
	for(size_t i = 0; i < ents.size(); ++i) ix->sa[i] = ents[i].pos;
	finish_index(ix);
	*out = ix;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_seed_index_create"); }
/* ------------------------------------------------------------------------------------------
 * Reading the reference's own index file, <DB>.csfm (CSFMIndex::save / load, src/CSFMIndex.cpp:176-230):
 *     string alphabet name (size_t length + bytes) | char gapCh | uint16 csLen | int32 concatLen | int32 C[256] |
 *     string csSeq | double csIdentity[csLen + 1] | uint16 concat2CS[concatLen + 1] | uint32 saSampled[concatLen / 4] |
 *     saIdx: a libcds BitSequenceRRR | bwt: a libcds WaveletTreeNoptrs over BitSequenceRRR levels with a MapperNone
 * libcds (vendored by the reference under src/libcds, v1.0.x) serialises
 *     BitSequenceRRR     uint32 2 | size_t length | size_t ones | uint32 C_len, C_field_bits, O_len, O_bits_len, sample_rate |
 *                        uint32 C[ceil(C_len * C_field_bits / 32)] | uint32 O[O_len]               (BitSequenceRRR.cpp:379-402)
 *                        blocks of 15 bits: C = popcount of a block, O = its index among the 15-bit words of that popcount in
 *                        the order TableOffsetRRR.cpp:101-134 generates them (positions ascending, lexicographic), in
 *                        bits(binomial(15, c) - 1) bits
 *     WaveletTreeNoptrs  uint32 3 | size_t n | size_t length | uint32 max_v | uint32 height | uint32 2 (MapperNone) |
 *                        height x BitSequenceRRR | uint32 OCC[max_v + 2]                            (WaveletTreeNoptrs.cpp:244-256)
 *                        level l holds bit (height - 1 - l) of the symbols, stably grouped by their higher bits (OCC = first
 *                        position of every symbol's group); n >= length: one padding symbol per value that does not occur.
 * The bit sequences are decoded to plain words once, the BWT symbols read off the wavelet tree, and every sequence is walked backwards
 * from the suffix that starts at its separator (rows 1 .. numSeq of the suffix array: the suffixes that start with the separator) by
 * LF steps on BASE symbols only — the LF step on the separator symbol is not a bijection in this index (separators and the
 * terminator share the symbol 0 and the BWT writes 0 for the row of text position 0), and is never needed.  The walk passes a
 * sampled suffix-array entry (every fourth text position) which anchors the sequence in the text; concat2CS gives the CS columns.
 * The rows met on the way are the ranks of the sequence's suffixes in the reference's own suffix order, so the seeds are indexed in
 * exactly that order and "the first hit" here is locateFirst's (src/CSFMIndex.cpp:92-119).
 * Tested against files written by the real libcds + libdivsufsort (oracle/csfm_ref.cpp, tests/test_csfm.py). */
namespace {
struct PlainBits {
	std::vector<uint64_t> w; std::vector<uint32_t> blk; size_t n = 0;      /* blk[b] = ones before word 8 b */
	bool get(size_t i) const { return (w[i >> 6] >> (i & 63)) & 1; }
	size_t rank1(size_t i) const { /* ones in [0, i] */
		const size_t wi = i >> 6; size_t r = blk[wi >> 3];
		for(size_t k = wi & ~(size_t) 7; k < wi; ++k) r += (size_t) __builtin_popcountll(w[k]);
		return r + (size_t) __builtin_popcountll(w[wi] & (~0ull >> (63 - (i & 63))));
	}
	void index() {
		blk.assign(w.size() / 8 + 2, 0);
		uint32_t run = 0;
		for(size_t k = 0; k < w.size(); ++k) { if((k & 7) == 0) blk[k >> 3] = run; run += (uint32_t) __builtin_popcountll(w[k]); }
		blk[(w.size() + 7) >> 3] = run;
	}
};
struct Reader {      /* every array is checked against what is left of the file BEFORE it is allocated: a damaged length cannot ask for memory */
	FILE* f; bool ok = true; uint64_t left = 0;
	explicit Reader(FILE* fp) : f(fp) { if(fseek(f, 0, SEEK_END) == 0) { const long e = ftell(f); if(e > 0) left = (uint64_t) e; } rewind(f); }
	template<class T> T val() { T v{}; if(left < sizeof(T) || fread(&v, sizeof(T), 1, f) != 1) ok = false; else left -= sizeof(T); return v; }
	template<class T> bool arr(std::vector<T>& v, uint64_t n) {
		if(!ok || n > left / sizeof(T)) return ok = false;
		v.resize((size_t) n); if(n && fread(v.data(), sizeof(T), (size_t) n, f) != n) return ok = false;
		left -= n * sizeof(T); return true;
	}
	bool str(std::string& s) { const uint64_t n = val<uint64_t>(); if(!ok || n > left) return ok = false; s.resize((size_t) n); if(n && fread(&s[0], 1, (size_t) n, f) != n) return ok = false; left -= n; return true; }
};
inline uint32_t bits_of(uint32_t n) { uint32_t b = 0; while(n) { ++b; n >>= 1; } return b; }
inline uint32_t field(const std::vector<uint32_t>& A, size_t ini, uint32_t len) { /* len bits starting at bit ini (libcds get_var_field) */
	if(len == 0) return 0;
	const size_t i = ini >> 5; const uint32_t j = (uint32_t)(ini & 31);
	uint64_t v = A[i]; if(j + len > 32) v |= (uint64_t) A[i + 1] << 32;
	return (uint32_t)((v >> j) & ((1ull << len) - 1));
}
/* the 15-bit words of every popcount in libcds's order + the width of an offset */
struct RrrTables {
	std::vector<uint16_t> word[16]; uint32_t width[16];
	RrrTables() {
		for(int c = 0; c <= 15; ++c) { gen(c, 0, 0, 0); width[c] = bits_of((uint32_t) word[c].size() - 1); }
	}
	void gen(int cls, int placed, int from, uint32_t made) {
		if(placed == cls) { word[cls].push_back((uint16_t) made); return; }
		for(int i = from; i < 15; ++i) gen(cls, placed + 1, i + 1, made | (1u << i));
	}
};
bool read_rrr(Reader& R, PlainBits& out) {
	static const RrrTables T;
	if(R.val<uint32_t>() != 2u) return false;                 /* RRR02_HDR */
	const uint64_t length = R.val<uint64_t>(), ones = R.val<uint64_t>();
	const uint32_t C_len = R.val<uint32_t>(), C_bits = R.val<uint32_t>(), O_len = R.val<uint32_t>(), O_bits_len = R.val<uint32_t>();
	(void) R.val<uint32_t>();                                  /* sample_rate: libcds rebuilds its samples on load, we keep plain words */
	if(!R.ok || C_bits != 4 || length > (1ull << 33) || (uint64_t) C_len != (length + 14) / 15 || (uint64_t) O_bits_len > (uint64_t) O_len * 32) return false;
	std::vector<uint32_t> C, O;
	if(!R.arr(C, ((uint64_t) C_len * C_bits + 31) / 32) || !R.arr(O, O_len)) return false;
	C.push_back(0); O.push_back(0); O.push_back(0);
	out.n = (size_t) length; out.w.assign(out.n / 64 + 2, 0);
	size_t pos = 0, cnt = 0;
	for(size_t k = 0; k < C_len; ++k) {
		const uint32_t c = field(C, k * 4, 4);
		if(c > 15 || pos + T.width[c] > (size_t) O_bits_len) return false;
		const uint32_t off = field(O, pos, T.width[c]); pos += T.width[c];
		if(off >= T.word[c].size()) return false;
		const uint64_t v = T.word[c][off]; cnt += c;
		const size_t b = k * 15;
		out.w[b >> 6] |= v << (b & 63);
		if((b & 63) > 49) out.w[(b >> 6) + 1] |= v >> (64 - (b & 63));
	}
	if(cnt != ones) return false;
	out.index();
	return true;
}
}

extern "C" int hu_seed_index_load_csfm(const char* path, int32_t K, const int32_t* p2cs, int32_t seed_len, hu_seed_index** out) try {
	if(!path || !p2cs || !out || seed_len < HU_SX_DIRK || seed_len > 31) { hu_set_error("hu_seed_index_load_csfm: bad argument (seed length must be in %d..31)", HU_SX_DIRK); return HU_ERR_ARG; }
	FILE* f = fopen(path, "rb");
	if(!f) { hu_set_error("cannot open %s", path); return HU_ERR_IO; }
	Reader R(f);
	auto fail = [&](const char* what) { fclose(f); hu_set_error("%s: %s", path, what); return HU_ERR_IO; };
	std::string abc, csSeq;
	if(!R.str(abc)) return fail("no alphabet name");
	if(abc != "DNA") return fail("not a DNA index");
	(void) R.val<char>();
	const uint16_t csLen = R.val<uint16_t>();
	const int32_t concatLen = R.val<int32_t>();
	std::vector<int32_t> Cc; std::vector<double> ident; std::vector<uint16_t> c2cs; std::vector<uint32_t> saS;
	if(!R.ok || concatLen < 1 || !R.arr(Cc, 256) || !R.str(csSeq) || csSeq.size() != (size_t) csLen + 1 || !R.arr(ident, (size_t) csLen + 1)
		|| !R.arr(c2cs, (size_t) concatLen + 1) || !R.arr(saS, (size_t) concatLen / 4)) return fail("truncated header");
	const size_t N = (size_t) concatLen + 1;
	PlainBits saIdx;
	if(!read_rrr(R, saIdx) || saIdx.n != N) return fail("bad sampled-suffix-array bitmap (BitSequenceRRR)");
	if(R.val<uint32_t>() != 3u) return fail("no WaveletTreeNoptrs");
	const uint64_t wn = R.val<uint64_t>(), wlen = R.val<uint64_t>();
	const uint32_t maxv = R.val<uint32_t>(), height = R.val<uint32_t>();
	if(!R.ok || wlen != N || wn < wlen || maxv > 255 || height != bits_of(maxv) || height < 1 || R.val<uint32_t>() != 2u) return fail("bad wavelet-tree header");
	std::vector<PlainBits> lev(height);
	for(uint32_t l = 0; l < height; ++l) if(!read_rrr(R, lev[l]) || lev[l].n != wn) return fail("bad wavelet-tree level (BitSequenceRRR)");
	std::vector<uint32_t> OCC;
	if(!R.arr(OCC, (size_t) maxv + 2)) return fail("truncated wavelet tree");
	fclose(f);
	unsigned nt = std::thread::hardware_concurrency(); if(nt > 16) nt = 16; if(nt < 1) nt = 1;
	auto par = [&](size_t n, size_t grain, const std::function<void(size_t, size_t)>& body) {
		std::atomic<size_t> next{0};
		auto work = [&] { for(;;) { const size_t a = next.fetch_add(grain); if(a >= n) break; body(a, std::min(n, a + grain)); } };
		hu_run_threads(nt, work);
	};
	/* the BWT symbols (WaveletTreeNoptrs::access, WaveletTreeNoptrs.cpp:301-323) */
	std::vector<uint8_t> L(N);
	std::atomic<int> bad{0};
	par(N, 1 << 16, [&](size_t a, size_t e) {
		for(size_t i = a; i < e; ++i) {
			uint32_t ret = 0; size_t pos = i, start = 0;
			for(uint32_t l = 0; l < height; ++l) {
				if(pos >= wn || start > wn) { bad = 1; ret = 255; break; }      /* a damaged level or OCC table sends the walk out of the level */
				const size_t before = start > 0 ? lev[l].rank1(start - 1) : 0;
				const size_t r1 = lev[l].rank1(pos);
				if(lev[l].get(pos)) { ret |= 1u << (height - l - 1); if(ret >= OCC.size() || r1 < 1 + before) { bad = 1; ret = 255; break; } start = OCC[ret]; pos = r1 - 1 - before + start; }
				else { if(pos + 1 < r1) { bad = 1; ret = 255; break; } pos = (pos + 1 - r1) - 1 + before; }
			}
			L[i] = (uint8_t) ret;
		}
	});
	if(bad) { hu_set_error("%s: the wavelet tree's levels and its OCC table do not fit together", path); return HU_ERR_IO; }
	for(size_t i = 0; i < N; ++i) if(L[i] > 4) { hu_set_error("%s: symbol %d in the BWT", path, (int) L[i]); return HU_ERR_IO; }
	{ /* C[] must be the running totals of the BWT's own symbol counts: then every LF step below lands inside [0, N) */
		uint64_t cnt[5] = {0, 0, 0, 0, 0}, run = 0;
		for(size_t i = 0; i < N; ++i) cnt[L[i]]++;
		for(int c = 0; c <= 4; ++c) { if(Cc[c] < 0 || (uint64_t) Cc[c] != run) { hu_set_error("%s: the symbol counts C[] do not match the BWT", path); return HU_ERR_IO; } run += cnt[c]; }
		if(Cc[5] < 0 || (uint64_t) Cc[5] != N || Cc[1] < 2) { hu_set_error("%s: the symbol counts C[] do not match the BWT", path); return HU_ERR_IO; }
	}
	for(size_t p = 0; p < N; ++p) if(c2cs[p] > csLen) { hu_set_error("%s: concat2CS names column %d of %d", path, (int) c2cs[p], (int) csLen); return HU_ERR_IO; }
	/* occurrences of the four bases before every 64th row */
	const size_t nb = N / 64 + 2;
	std::vector<uint32_t> occ(nb * 4, 0);
	{ uint32_t run[5] = {0, 0, 0, 0, 0}; for(size_t i = 0; i < N; ++i) { if((i & 63) == 0) for(int c = 1; c <= 4; ++c) occ[(i >> 6) * 4 + c - 1] = run[c]; run[L[i]]++; }
	  for(size_t b = (N + 63) / 64; b < nb; ++b) for(int c = 1; c <= 4; ++c) occ[b * 4 + c - 1] = run[c]; }
	auto rankc = [&](int c, size_t i) { /* occurrences of base c in rows [0, i] */
		size_t r = occ[(i >> 6) * 4 + c - 1];
		for(size_t k = i & ~(size_t) 63; k <= i; ++k) r += L[k] == c;
		return r;
	};
	const size_t nSeq = (size_t) Cc[1] - 1;              /* rows 0 .. C[1]-1 start with the symbol 0: the terminator + one separator per sequence */
	std::vector<uint32_t> rowAt(N, 0xffffffffu);          /* rank of the suffix at every text position that holds a base */
	std::vector<int64_t> sepOf(nSeq + 1, -1);             /* text position of the separator whose suffix is row j */
	par(nSeq, 64, [&](size_t a, size_t e) {
		std::vector<uint32_t> rows;
		for(size_t j = a + 1; j <= e; ++j) {
			rows.clear();
			size_t row = j; int64_t s = -1;
			auto sampled = [&](size_t rw, size_t t) { if(s < 0 && saIdx.get(rw)) { const size_t q = saIdx.rank1(rw) - 1; if(q < saS.size()) s = (int64_t) saS[q] + (int64_t) t; } };
			sampled(row, 0);
			for(;;) {
				const int c = L[row];
				if(c == 0) break;
				row = (size_t) Cc[c] + rankc(c, row) - 1;
				rows.push_back((uint32_t) row);
				sampled(row, rows.size());
				if(rows.size() > N) { bad = 1; break; }
			}
			if(s < 0) continue;                      /* fewer than four bases and no sample: cannot hold a seed anyway */
			const size_t len = rows.size();
			if(s >= (int64_t) N || (int64_t) len > s || c2cs[(size_t) s] != 0 || (s - (int64_t) len > 0 && c2cs[(size_t) s - len - 1] != 0)) { bad = 1; continue; }
			sepOf[j] = s;
			for(size_t t = 0; t < len; ++t) { if(c2cs[(size_t) s - 1 - t] == 0) bad = 1; rowAt[(size_t) s - 1 - t] = rows[t]; }
		}
	});
	if(bad) { hu_set_error("%s: the BWT, the sampled suffix array and concat2CS do not describe one text", path); return HU_ERR_IO; }
	/* the sequences in text order */
	hu_seed_index* ix = new hu_seed_index;
	ix->seedLen = seed_len; ix->csLen = csLen; ix->K = K;
	ix->cs2p.assign((size_t) csLen + 2, 0);
	for(int k = 1; k <= K; ++k) if(p2cs[k] >= 1 && p2cs[k] <= csLen) ix->cs2p[p2cs[k]] = k;
	for(int i = std::max(1, K >= 0 ? p2cs[K] + 1 : 1); i <= csLen; ++i) ix->cs2p[i] = K;
	struct Ent { uint32_t row, pos; };
	std::vector<Ent> ents;
	size_t at = 0;
	ix->text.assign((size_t)(N / 32 + 3), 0);
	ix->cols.reserve(N);
	for(size_t p = 0; p < N; ) {
		if(c2cs[p] == 0) { ++p; continue; }
		size_t e = p; while(e < N && c2cs[e] != 0) ++e;                /* [p, e) = one sequence; e = its separator */
		const bool placed = rowAt[p] != 0xffffffffu;                   /* a sequence the walk could not anchor (< 4 bases) is left out */
		if(placed) {
			const size_t b0 = at;
			for(size_t q = p; q < e; ++q) {
				if(rowAt[q] == 0xffffffffu) { delete ix; hu_set_error("%s: a sequence is only partly covered by the BWT walk", path); return HU_ERR_IO; }
				ix->cols.push_back((uint16_t)(c2cs[q] - 1));
				++at;
			}
			for(size_t q = p; q + (size_t) seed_len <= e; ++q) ents.push_back(Ent{rowAt[q], (uint32_t)(b0 + (q - p))});
			ix->seqEnd.push_back((uint32_t) at);
		}
		p = e;
	}
	/* the residues: the base at text position q is the BWT symbol of the row of position q + 1 (its suffix's predecessor) — for the
	 * last base of a sequence that is the row of its separator's suffix */
	{
		std::vector<uint8_t> base(N, 0);
		for(size_t j = 1; j <= nSeq; ++j) if(sepOf[j] > 0) base[(size_t) sepOf[j] - 1] = L[j];
		for(size_t q = 0; q + 1 < N; ++q) if(rowAt[q + 1] != 0xffffffffu && c2cs[q] != 0) base[q] = L[rowAt[q + 1]];
		size_t a2 = 0;
		for(size_t p = 0; p < N; ++p) {
			if(c2cs[p] == 0 || rowAt[p] == 0xffffffffu) continue;
			if(base[p] < 1 || base[p] > 4) { delete ix; hu_set_error("%s: no base at text position %zu", path, p); return HU_ERR_IO; }
			ix->text[a2 >> 5] |= (uint64_t)(base[p] - 1) << (62 - 2 * (int)(a2 & 31));
			++a2;
		}
		if(a2 != at) { delete ix; hu_set_error("%s: internal: residue count", path); return HU_ERR_IO; }
	}
	ix->nRes = (int64_t) at;
	{ /* the rows are distinct ranks in [0, N): order by them with one scatter and one sweep */
		std::vector<uint32_t> byRow(N, 0xffffffffu);
		for(const Ent& e : ents) byRow[e.row] = e.pos;
		ix->sa.clear(); ix->sa.reserve(ents.size());
		for(size_t r = 0; r < N; ++r) if(byRow[r] != 0xffffffffu) ix->sa.push_back(byRow[r]);
		if(ix->sa.size() != ents.size()) { delete ix; hu_set_error("%s: two suffixes share a rank", path); return HU_ERR_IO; }
	}
	finish_index(ix);
	*out = ix;
	return HU_OK;
} catch(...) { return hu_catch_all("hu_seed_index_load_csfm"); }

extern "C" void hu_seed_index_destroy(hu_seed_index* ix) try { delete ix; } catch(...) { (void) hu_catch_all("hu_seed_index_destroy"); }
/* number of distinct seed_len-mers indexed */
extern "C" int64_t hu_seed_index_size(const hu_seed_index* ix) try { return ix ? ix->distinct : 0; } catch(...) { return hu_catch_all("hu_seed_index_size"); }
/* resident bytes of the index; positions = number of indexed k-mer starts */
extern "C" int64_t hu_seed_index_bytes(const hu_seed_index* ix, int64_t* positions) try {
	if(!ix) return 0;
	if(positions) *positions = (int64_t) ix->sa.size();
	return (int64_t)(ix->sa.size() * 4 + ix->dir.size() * 4 + ix->cols.size() * 2 + ix->text.size() * 8 + ix->seqEnd.size() * 4 + ix->cs2p.size() * 4);
} catch(...) { return hu_catch_all("hu_seed_index_bytes"); }

/* locateFirst + buildAlignPath for the k-mer read[from0 .. from0+seedLen): returns 1 and fills out6 when
 * the k-mer occurs and yields a valid path, else 0 */
/* rnd != 0: CSFMIndex::locateOne instead (src/CSFMIndex.cpp:121-147): a member of the seed's hit range drawn with the 64-bit number rnd
 * (start + rnd % count, as the reference draws with rand()).  The reference takes rand() from one global stream that its tasks race on
 * (SURVEY.md F7); here the number is a hash of (user seed, read number, seed position), so a run is reproducible at any thread count. */
static int lookup_one(const hu_seed_index* ix, const char* read, int from0, int32_t* out6, uint64_t rnd = 0) {
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
	if(rnd) { /* the end of the hit range, then one of its members */
		size_t l2 = lo, h2 = ix->dir[b + 1];
		while(l2 < h2) { const size_t mid = (l2 + h2) >> 1; if(keep_top(window32(ix->text, ix->sa[mid]), k) <= key) l2 = mid + 1; else h2 = mid; }
		lo += (size_t)(rnd % (uint64_t)(l2 - lo));
	}
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
extern "C" int64_t hu_seed_index_occurrences(const hu_seed_index* ix, const char* kmer, int32_t* seq_no, int32_t* offset, int32_t* cs_col, int64_t cap) try {
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
} catch(...) { return hu_catch_all("hu_seed_index_occurrences"); }

/* CSFMIndex::locateFirst for one seed of the index's length (src/CSFMIndex.cpp:92-119): 1-based CS columns of its first and last
 * base at the first hit (0, 0 without a hit), and the number of hits (CSFMIndex::count, :43-64).  Returns 1 on a hit. */
extern "C" int hu_seed_index_locate_first(const hu_seed_index* ix, const char* kmer, int32_t* cs_start, int32_t* cs_end, int64_t* count) try {
	if(cs_start) *cs_start = 0;
	if(cs_end) *cs_end = 0;
	if(count) *count = 0;
	if(!ix || !kmer) return 0;
	const int k = ix->seedLen;
	uint64_t key = 0;
	for(int i = 0; i < k; ++i) { const int8_t c = sym_code(kmer[i]); if(c < 0) return 0; key |= (uint64_t) c << (62 - 2 * i); }
	const size_t b = (size_t)(key >> (64 - 2 * HU_SX_DIRK));
	int64_t n = 0; size_t first = 0;
	for(size_t i = ix->dir[b]; i < ix->dir[b + 1]; ++i) {
		const uint64_t km = keep_top(window32(ix->text, ix->sa[i]), k);
		if(km < key) continue;
		if(km > key) break;
		if(n++ == 0) first = i;
	}
	if(count) *count = n;
	if(!n) return 0;
	const uint16_t* cols = &ix->cols[ix->sa[first]];
	if(cs_start) *cs_start = cols[0] + 1;
	if(cs_end) *cs_end = cols[k - 1] + 1;
	return 1;
} catch(...) { return hu_catch_all("hu_seed_index_locate_first"); }

/* the two seed scans of alignSeq (src/HmmUFOtu_main.cpp:50-84) for n reads; vpaths [n][2][6] */
static inline uint64_t hu_mix64(uint64_t x) { x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }
static int lookup_impl(const hu_seed_index* ix, int n, const char* bases, const int64_t* offs, int seed_region, int align_mode, int32_t* vpaths,
		bool random, uint64_t seed, int64_t firstRead);
extern "C" int hu_seed_index_lookup(const hu_seed_index* ix, int n, const char* bases, const int64_t* offs, int seed_region,
		int align_mode, int32_t* vpaths) try {
	return lookup_impl(ix, n, bases, offs, seed_region, align_mode, vpaths, false, 0, 0);
} catch(...) { return hu_catch_all("hu_seed_index_lookup"); }
extern "C" int hu_seed_index_lookup_random(const hu_seed_index* ix, int n, const char* bases, const int64_t* offs, int seed_region,
		int align_mode, uint64_t seed, int64_t first_read, int32_t* vpaths) try {
	return lookup_impl(ix, n, bases, offs, seed_region, align_mode, vpaths, true, seed, first_read);
} catch(...) { return hu_catch_all("hu_seed_index_lookup_random"); }
static int lookup_impl(const hu_seed_index* ix, int n, const char* bases, const int64_t* offs, int seed_region, int align_mode, int32_t* vpaths,
		bool random, uint64_t seed, int64_t firstRead) {
	if(!ix || n < 0 || (n && (!bases || !offs || !vpaths))) { hu_set_error("hu_seed_index_lookup: bad argument"); return HU_ERR_ARG; }
	const int seedLen = ix->seedLen;
	auto rnd = [&](int r, int pos) -> uint64_t { return random ? (hu_mix64(hu_mix64(seed ^ 0x5851f42d4c957f2dull) + (uint64_t)(firstRead + r) * 0x2545f4914f6cdd1dull + (uint64_t) pos) | (1ull << 63)) : 0; };
	auto one = [&](int r) {
		const char* read = bases + offs[r];
		const int len = (int)(offs[r + 1] - offs[r]);
		int32_t* vp = vpaths + (size_t) r * 12;
		memset(vp, 0, 12 * sizeof(int32_t));
		int k = 0;
		const int regionLen = seed_region < len ? seed_region : len;
		for(int seedFrom = 0; seedFrom + seedLen - 1 < regionLen; ++seedFrom)
			if(lookup_one(ix, read, seedFrom, vp + 6 * k, rnd(r, seedFrom))) { ++k; break; }
		if(align_mode == HU_MODE_GLOBAL && (k == 0 || len >= 2 * regionLen))
			for(int seedTo = len - 1; seedTo - seedLen + 1 >= len - regionLen && seedTo - seedLen + 1 >= 0; --seedTo)
				if(lookup_one(ix, read, seedTo - seedLen + 1, vp + 6 * k, rnd(r, seedTo - seedLen + 1))) { ++k; break; }
	};
	unsigned nt = std::thread::hardware_concurrency();
	if(nt > 16) nt = 16;
	if(n < 1024 || nt <= 1) { for(int r = 0; r < n; ++r) one(r); return HU_OK; }
	std::atomic<int> next{0};
	hu_run_threads(nt, [&] { for(;;) { int a = next.fetch_add(256); if(a >= n) break; int e = std::min(n, a + 256); for(int r = a; r < e; ++r) one(r); } });
	return HU_OK;
}
