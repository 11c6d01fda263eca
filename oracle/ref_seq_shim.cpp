// ORACLE — test infrastructure only.  A C door into the part of the reference that compiles without Eigen3 / Boost:
// its alphabet, DigitalSeq, PrimarySeq and SeqUtils::pDist (the inner function of getSeed, src/HmmUFOtu_main.cpp:127-152) —
// the reference's own sources, compiled from /root/reference/src where they lie (oracle/Makefile, target _ref/libref_seq.so).
// This file holds no reference code: it only calls it.  tests/test_ref_seq.py checks the oracle's restatements against it.
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include "AlphabetFactory.h"
#include "ProgEnv.h"
#include "StringUtils.h"
#include "DigitalSeq.h"
#include "PrimarySeq.h"
#include "SeqUtils.h"
#include "SeqIO.h"
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

/* ---- the head of a .ptu through the reference's own serialisation code ----
 * hmmufotu-build writes saveProgInfo(out) (src/util/ProgEnv.cpp:24-28: program name without a length + VersionSequence::save) and then
 * PTUnrooted::save (src/PhyloTreeUnrooted.cpp:537-566): size_t nNodes, int csLen, one PTUNode::save per node (:116-129: long id,
 * StringUtils::saveString(name), DigitalSeq::save, saveString(anno), double annoDist), then edges, root, heights, MSA index, models.
 * PhyloTreeUnrooted.cpp itself needs Eigen3 and cannot be compiled here, but everything a node record is made of can: the two functions
 * below call ProgEnv / StringUtils / DigitalSeq as compiled from the reference tree and add only the raw id / annoDist / count fields
 * exactly as the cited lines do. */
/* writes the program header, nNodes, csLen and the node records; returns the number of bytes, -1 when cap is too small */
long ref_ptu_head_write(long nNodes, int csLen, const int8_t* codes /* [nNodes][csLen], a row of -128 = node without a sequence */,
		const char* const* names, const char* const* annos, const double* annoDist, char* out, long cap) {
	std::ostringstream o(std::ios::binary);
	EGriceLab::saveProgInfo(o);
	const size_t n = (size_t) nNodes;
	o.write((const char*) &n, sizeof(size_t));
	o.write((const char*) &csLen, sizeof(int));
	for(long i = 0; i < nNodes; ++i) {
		o.write((const char*) &i, sizeof(long));
		EGriceLab::StringUtils::saveString(std::string(names[i]), o);
		DigitalSeq seq(AlphabetFactory::nuclAbc, names[i]);
		const int8_t* row = codes + (size_t) i * csLen;
		if(row[0] != -128) seq.assign(row, row + csLen);
		seq.save(o);
		EGriceLab::StringUtils::saveString(std::string(annos[i]), o);
		o.write((const char*) &annoDist[i], sizeof(double));
	}
	const std::string b = o.str();
	if((long) b.size() > cap) return -1;
	memcpy(out, b.data(), b.size());
	return (long) b.size();
}
/* reads the same head back from a file (loadProgInfo, src/util/ProgEnv.cpp:30-66; PTUNode::load, src/PhyloTreeUnrooted.cpp:99-114);
 * names / annos come back '\n'-joined.  Returns the file offset at which the edge block starts, -1 on a refused header, -2 when a
 * buffer is too small */
long ref_ptu_head_read(const char* path, long* nNodes, int* csLen, int8_t* codes, long codesCap, char* names, long namesCap,
		char* annos, long annosCap, double* annoDist, long* seqLens) {
	std::ifstream in(path, std::ios::binary);
	if(!in) return -1;
	EGriceLab::loadProgInfo(in);
	if(in.bad() || !in) return -1;
	size_t n = 0; int L = 0;
	in.read((char*) &n, sizeof(size_t));
	in.read((char*) &L, sizeof(int));
	if(!in || (long)(n * (size_t) L) > codesCap) return -2;
	*nNodes = (long) n; *csLen = L;
	std::string allNames, allAnnos;
	for(size_t i = 0; i < n; ++i) {
		long id = -1; std::string name, anno; double ad = 0;
		in.read((char*) &id, sizeof(long));
		EGriceLab::StringUtils::loadString(name, in);
		DigitalSeq seq;
		seq.load(in);
		EGriceLab::StringUtils::loadString(anno, in);
		in.read((char*) &ad, sizeof(double));
		if(!in || id != (long) i || (seq.length() != 0 && (int) seq.length() != L)) return -1;
		seqLens[i] = (long) seq.length();
		for(size_t j = 0; j < seq.length(); ++j) codes[i * (size_t) L + j] = seq[j];
		annoDist[i] = ad;
		allNames += name; allNames += '\n'; allAnnos += anno; allAnnos += '\n';
	}
	if((long) allNames.size() >= namesCap || (long) allAnnos.size() >= annosCap) return -2;
	memcpy(names, allNames.c_str(), allNames.size() + 1); memcpy(annos, allAnnos.c_str(), allAnnos.size() + 1);
	return (long) in.tellg();
}
/* readProgInfo (src/util/ProgEnv.cpp:106-134) on the first line of an assignment file: 1 = accepted, 0 = refused;
 * writeProgInfo (:101-104) into out */
int ref_read_prog_info(const char* text) {
	std::istringstream in(text);
	std::streambuf* keep = std::cerr.rdbuf(nullptr);      /* the reference prints its refusal to stderr */
	int rc;
	try { EGriceLab::readProgInfo(in); rc = in.bad() ? 0 : 1; } catch(...) { rc = -1; }      /* VersionSequence may throw on a malformed version */
	std::cerr.rdbuf(keep);
	return rc;
}
int ref_write_prog_info(const char* info, char* out, int cap) {
	std::ostringstream o;
	EGriceLab::writeProgInfo(o, info);
	const std::string b = o.str();
	if((int) b.size() >= cap) return -1;
	memcpy(out, b.c_str(), b.size() + 1);
	return (int) b.size();
}
/* SeqIO (src/SeqIO.cpp:75-119) over a file: records as id \x1f desc \x1f seq \n into out; returns the number of records, -1 when
 * the reference throws (PrimarySeq refuses a character), -2 when out is too small */
long ref_seqio_read(const char* path, const char* fmt, char* out, long cap) {
	std::ifstream in(path);
	if(!in) return -3;
	std::string all; long n = 0;
	try {
		SeqIO io(&in, AlphabetFactory::nuclAbc, fmt);
		while(io.hasNext()) {
			const PrimarySeq s = io.nextSeq();
			all += s.getId(); all += '\x1f'; all += s.getDesc(); all += '\x1f'; all += s.getSeq(); all += '\n';
			++n;
		}
	} catch(...) { return -1; }
	if((long) all.size() >= cap) return -2;
	memcpy(out, all.c_str(), all.size() + 1);
	return n;
}
}
