// ORACLE — test infrastructure only.  A C door into the part of the reference that compiles without Eigen3 / Boost:
// its alphabet, DigitalSeq, PrimarySeq and SeqUtils::pDist (the inner function of getSeed, src/HmmUFOtu_main.cpp:127-152) —
// the reference's own sources, compiled from /root/reference/src where they lie (oracle/Makefile, target _ref/libref_seq.so).
// This file holds no reference code: it only calls it.  tests/test_ref_seq.py checks the oracle's restatements against it.
#include <cstdint>
#include <cstring>
#include <string>
#include "AlphabetFactory.h"
#include "DigitalSeq.h"
#include "PrimarySeq.h"
#include "SeqUtils.h"
using namespace EGriceLab::HmmUFOtu;

extern "C" {
/* DigitalSeq(abc, name, str) (src/DigitalSeq.cpp:41-48): codes of the valid characters; returns their number */
int ref_digitize(const char* str, int8_t* out, int cap) {
	const DigitalSeq s(AlphabetFactory::nuclAbc, "s", str);
	const int n = (int) s.length();
	for(int i = 0; i < n && i < cap; ++i) out[i] = s[i];
	return n;
}
/* SeqUtils::pDist(seq1, seq2, start, end) (src/SeqUtils.cpp:37-54) on two code rows of the same length */
double ref_pdist(const int8_t* a, const int8_t* b, int len, int start, int end) {
	DigitalSeq x(AlphabetFactory::nuclAbc, "a"), y(AlphabetFactory::nuclAbc, "b");
	x.assign(a, a + len); y.assign(b, b + len);
	return SeqUtils::pDist(x, y, (size_t) start, (size_t) end);
}
/* PrimarySeq::revcom (src/PrimarySeq.h:229-236) */
int ref_revcom(const char* str, char* out, int cap) {
	PrimarySeq p(AlphabetFactory::nuclAbc, "s", str);
	const std::string r = p.revcom().getSeq();
	if((int) r.size() >= cap) return -1;
	memcpy(out, r.c_str(), r.size() + 1);
	return (int) r.size();
}
/* DegenAlphabet::isValid / encode / isGap for one character (src/DegenAlphabet.cpp:43-64) */
int ref_encode(char c) { return AlphabetFactory::nuclAbc->isValid(c) ? AlphabetFactory::nuclAbc->encode(c) : -100; }
}
