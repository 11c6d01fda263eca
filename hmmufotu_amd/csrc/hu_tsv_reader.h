// The READ side of an assignment file as the reference's downstream tools do it:
//   readProgInfo   src/util/ProgEnv.cpp:106-134    first line "# <progName> <version> ...": name HmmUFOtu, version <= v1.5.1
//   TSVScanner     src/util/TSVScanner.cpp:18-43    comment lines skipped up to the header line; fields by NAME, tab separated
// plain files or gzip (zlib; the reference also reads .bz2 through boost::iostreams: not here).  Host only.
#pragma once
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace hu_tsv {

struct LineReader {
	gzFile z = nullptr;
	std::string buf; size_t pos = 0; bool eof = false;
	bool open(const std::string& fn) { z = gzopen(fn.c_str(), "rb"); if(z) gzbuffer(z, 1 << 20); return z != nullptr; }   /* gzopen reads plain files as they are */
	bool line(std::string& out) {
		out.clear();
		for(;;) {
			const size_t nl = buf.find('\n', pos);
			if(nl != std::string::npos) { out.append(buf, pos, nl - pos); pos = nl + 1; return true; }
			out.append(buf, pos, std::string::npos); buf.clear(); pos = 0;
			if(eof) return !out.empty();
			buf.resize(1 << 20);
			const int k = gzread(z, &buf[0], (unsigned) buf.size());
			buf.resize(k > 0 ? (size_t) k : 0);
			if(k <= 0) eof = true;
		}
	}
	~LineReader() { if(z) gzclose(z); }
};

/* readProgInfo: sscanf(header, "# %s %s"), progName == "HmmUFOtu", VersionSequence("v%d.%d.%d") <= 1.5.1 */
inline bool read_prog_info(const std::string& header, std::string& why) {
	char pname[256], ver[256];
	if(sscanf(header.c_str(), "# %255s %255s", pname, ver) != 2) { why = "Unrecognized input file for HmmUFOtu"; return false; }
	if(strcmp(pname, "HmmUFOtu") != 0) { why = "Not an valid input file of HmmUFOtu"; return false; }
	int v[3] = {0, 0, 0};
	sscanf(ver, "v%d.%d.%d", &v[0], &v[1], &v[2]);
	const int mine[3] = {1, 5, 1};
	for(int i = 0; i < 3; ++i) { if(v[i] < mine[i]) return true; if(v[i] > mine[i]) { why = std::string("You are using an old version HmmUFOtu v1.5.1 to read a newer data file that is build by HmmUFOtu ") + ver + ", please download the latest program"; return false; } }
	return true;
}

struct Scanner {
	LineReader in;
	std::vector<std::string> header; std::map<std::string, size_t> col;
	std::vector<std::string> f;
	bool open(const std::string& fn, std::string& why) {
		if(!in.open(fn)) { why = "Unable to open assignment input file '" + fn + "'"; return false; }
		std::string first;
		if(!in.line(first) || !read_prog_info(first, why)) { if(why.empty()) why = "Unrecognized input file for HmmUFOtu"; return false; }
		std::string l;
		while(in.line(l)) { /* TSVScanner(in, hasHeader = true): comments skipped up to the header line */
			if(l.empty() || l[0] == '#') continue;
			split(l, header);
			for(size_t i = 0; i < header.size(); ++i) col[header[i]] = i;
			return true;
		}
		why = "no header line in '" + fn + "'";
		return false;
	}
	static void split(const std::string& l, std::vector<std::string>& out) {
		out.clear();
		size_t a = 0;
		for(;;) { const size_t t = l.find('\t', a); if(t == std::string::npos) { out.push_back(l.substr(a)); break; } out.push_back(l.substr(a, t - a)); a = t + 1; }
	}
	bool next() { std::string l; while(in.line(l)) { if(l.empty()) continue; split(l, f); return true; } return false; }
	const std::string& get(const char* name) const { static const std::string none; auto it = col.find(name); return it == col.end() || it->second >= f.size() ? none : f[it->second]; }
};

/* DegenAlphabet::isSymbol (src/DegenAlphabet.h:89-91: sym_map[c] >= 0) over IUPACNucl (src/IUPACNucl.cpp:33-50): the four bases and the
 * degenerate codes, which map to their first expansion — upper case only (inserts are lower case in an alignment and do not count):
 * what alignIdentity / hmmIdentity count (src/HmmUFOtu_main.cpp:218-239) */
inline bool is_symbol(char c) { return c != '\0' && strchr("ACGTUMRWSYKVHDBN", c) != nullptr; }
inline double align_identity(const std::string& aln, int start, int end) {
	int id = 0;
	for(int i = start; i <= end && i < (int) aln.size(); ++i) id += is_symbol(aln[i]);
	return (double) id / (end - start + 1);
}
inline double hmm_identity(const std::vector<int32_t>& cs2p, const std::string& aln, int start, int end) {
	int id = 0, n = 0;
	for(int i = start; i <= end && i < (int) aln.size(); ++i) if(i + 1 < (int) cs2p.size() && cs2p[i + 1] != 0) { ++n; id += is_symbol(aln[i]); }
	return (double) id / n;
}
/* profile2CSIdx of a .hmm -> cs2ProfileIdx [L + 1] (src/BandedHMMP7.h getProfileLoc) */
inline std::vector<int32_t> cs_to_profile(int K, int L, const std::vector<int32_t>& p2cs) {
	std::vector<int32_t> m((size_t) L + 2, 0);
	for(int k = 1; k <= K; ++k) if(p2cs[k] >= 1 && p2cs[k] <= L) m[p2cs[k]] = k;
	return m;
}

} // namespace hu_tsv
