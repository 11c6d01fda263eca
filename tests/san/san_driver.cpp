// Test infrastructure: the product's HOST-side file readers (hu_host.cpp: .hmm and .ptu; hu_seedindex.cpp: .csfm) built for the CPU
// with AddressSanitizer + UBSan (tests/test_sanitizer.py compiles this file together with those two sources; no HIP code in them) and
// fed damaged copies of valid files: truncations at every scale, byte flips, and length fields overwritten with large values.
// A reader may accept or refuse a damaged file; it may not touch memory it does not own, overflow, or ask for more memory than the
// file could possibly describe (the sanitizer's allocator aborts on those).  Exit code 0 = every trial returned.
//
// usage: san_driver <kind: hmm|ptu|csfm> <valid file> <scratch path> <trials> [<K for csfm> <seed_len>]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "hu_common.h"

static std::vector<char> slurp(const char* p) {
	std::ifstream in(p, std::ios::binary);
	return std::vector<char>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
static void spit(const char* p, const std::vector<char>& v, size_t n) {
	std::ofstream out(p, std::ios::binary | std::ios::trunc);
	out.write(v.data(), (std::streamsize) n);
}

static int K_csfm = 0, seedLen = 20;
static std::vector<int32_t> p2cs;

static int run(const std::string& kind, const char* path) {
	if(kind == "hmm") {
		HuProfileHost prof; std::vector<double> EM, EI, T; std::vector<int32_t> map; int K = 0, L = 0;
		return hu_read_hmm(path, prof, EM, EI, T, map, K, L);
	}
	if(kind == "ptu") { HuTreeHost t; return hu_read_ptu(path, t); }
	hu_seed_index* ix = nullptr;
	int rc = hu_seed_index_load_csfm(path, K_csfm, p2cs.data(), seedLen, &ix);
	if(rc == 0 && ix) {
		int32_t a = 0, b = 0; int64_t c = 0;
		std::string q(seedLen, 'A');
		hu_seed_index_locate_first(ix, q.c_str(), &a, &b, &c);
		hu_seed_index_destroy(ix);
	}
	return rc;
}

int main(int argc, char** argv) {
	if(argc < 5) { fprintf(stderr, "usage: san_driver <hmm|ptu|csfm> <file> <scratch> <trials> [K seed_len]\n"); return 2; }
	const std::string kind = argv[1];
	const std::vector<char> good = slurp(argv[2]);
	const char* scratch = argv[3];
	const int trials = atoi(argv[4]);
	if(good.empty()) { fprintf(stderr, "empty input\n"); return 2; }
	if(kind == "csfm") {
		if(argc < 7) return 2;
		K_csfm = atoi(argv[5]); seedLen = atoi(argv[6]);
		p2cs.resize(K_csfm + 1); for(int k = 0; k <= K_csfm; ++k) p2cs[k] = k;      /* identity map: columns = profile positions */
	}
	if(run(kind, argv[2]) != 0) { fprintf(stderr, "the undamaged file was refused: %s\n", hu_last_error()); return 3; }
	std::mt19937_64 rng(12345);
	int accepted = 0, refused = 0;
	for(int t = 0; t < trials; ++t) {
		std::vector<char> v = good;
		size_t n = v.size();
		const int how = t % 4;
		if(how == 0) {                                  /* truncate: a fifth of the trials near the head, the rest anywhere */
			n = (t % 20 == 0) ? rng() % std::min<size_t>(n, 4096) : rng() % n;
		} else if(how == 1) {                           /* flip 1..8 bytes */
			const int k = 1 + (int)(rng() % 8);
			for(int i = 0; i < k; ++i) v[rng() % n] ^= (char)(1 + rng() % 255);
		} else if(how == 2) {                           /* a large little-endian value over 4 or 8 bytes somewhere in the first 64 KB (where the headers sit) */
			const size_t span = std::min<size_t>(n - 8, 65536);
			const size_t at = rng() % span;
			const uint64_t big = (rng() % 2) ? 0x7fffffffffffff00ull >> (rng() % 40) : 0xffffffffull >> (rng() % 8);
			memcpy(&v[at], &big, (rng() % 2) ? 8 : 4);
		} else {                                        /* a negative 32-bit value, 4-byte aligned or not */
			const size_t at = rng() % (n - 4);
			const int32_t neg = -(int32_t)(1 + rng() % 100000);
			memcpy(&v[at], &neg, 4);
		}
		spit(scratch, v, n);
		if(run(kind, scratch) == 0) accepted++; else refused++;
	}
	printf("%s: %d trials, %d accepted, %d refused\n", kind.c_str(), trials, accepted, refused);
	return 0;
}
