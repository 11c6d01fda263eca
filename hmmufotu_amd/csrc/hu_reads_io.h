// Reading FASTA / FASTQ read files (plain or gzip) for the hmmufotu-amd CLI: the counterpart of the reference's SeqIO
// (src/SeqIO.cpp:75-119: nextFastaSeq / nextFastqSeq) over zlib.  Header-only so that tests/test_ref_seq.py can drive it on the CPU
// (tests/san/reads_driver.cpp) beside the reference's own SeqIO compiled from /root/reference.
//   id   = the first word after the tag, desc = the rest of the header line without the blanks between them (SeqIO.cpp:83-87)
//   FASTA sequence = the following lines up to the next line that starts with '>', joined; FASTQ = exactly four lines per record
// Deliberate differences: a trailing '\r' (and trailing blanks of a FASTA line) are dropped — the reference's PrimarySeq constructor throws
// on them and the program ends; blank lines BEFORE a record are skipped (the reference stops reading there); bases are upper-cased here
// (the reference keeps the case and upper-cases when it encodes, src/DigitalSeq.cpp:41-48).
#pragma once
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>
#include <dlfcn.h>

struct Read { std::string id, desc, seq; };

/* bzip2 (the reference reads and writes .bz2 through boost::iostreams, src/hmmufotu.cpp:22): the image ships libbz2.so.1 without its header, so the four
 * entry points of its stdio-like interface are declared here as bzlib.h documents them and bound at run time; without the library a .bz2 name fails to open */
struct HuBz2 {
	void* lib = nullptr;
	void* (*bzopen)(const char*, const char*) = nullptr; int (*bzread)(void*, void*, int) = nullptr; int (*bzwrite)(void*, void*, int) = nullptr; void (*bzclose)(void*) = nullptr;
	static const HuBz2& get() {
		static const HuBz2 b = [] {
			HuBz2 x;
			for(const char* nm : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) if((x.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
			if(x.lib) {
				x.bzopen = (void* (*)(const char*, const char*)) dlsym(x.lib, "BZ2_bzopen"); x.bzread = (int (*)(void*, void*, int)) dlsym(x.lib, "BZ2_bzread");
				x.bzwrite = (int (*)(void*, void*, int)) dlsym(x.lib, "BZ2_bzwrite"); x.bzclose = (void (*)(void*)) dlsym(x.lib, "BZ2_bzclose");
			}
			return x;
		}();
		return b;
	}
	bool ok() const { return bzopen && bzread && bzwrite && bzclose; }
	static bool named(const std::string& fn) { return fn.size() > 4 && fn.compare(fn.size() - 4, 4, ".bz2") == 0; }
};

/* line reader over zlib: plain and gzip-compressed inputs alike, bzip2 by name (the reference reads .gz / .bz2 through boost::iostreams) */
struct LineIn {
	gzFile f = nullptr; void* bz = nullptr; std::vector<char> buf; size_t pos = 0, len = 0; bool eof = false;
	bool open(const std::string& fn) {
		if(HuBz2::named(fn)) { if(HuBz2::get().ok()) bz = HuBz2::get().bzopen(fn.c_str(), "rb"); if(bz) buf.resize(1 << 20); return bz != nullptr; }
		f = gzopen(fn.c_str(), "rb"); if(f) { gzbuffer(f, 1 << 20); buf.resize(1 << 20); } return f != nullptr;
	}
	~LineIn() { if(f) gzclose(f); if(bz) HuBz2::get().bzclose(bz); }
	bool fill() { if(eof) return false; const int k = bz ? HuBz2::get().bzread(bz, buf.data(), (int) buf.size()) : gzread(f, buf.data(), (unsigned) buf.size()); pos = 0; len = k > 0 ? (size_t) k : 0; if(k <= 0) eof = true; return k > 0; }
	int peek() { if(pos >= len && !fill()) return EOF; return (unsigned char) buf[pos]; }
	bool getline(std::string& s) {
		s.clear();
		if(pos >= len && !fill()) return false;
		for(;;) {
			const char* b = buf.data() + pos; const char* e = (const char*) memchr(b, '\n', len - pos);
			if(e) { s.append(b, e - b); pos += (size_t)(e - b) + 1; return true; }
			s.append(b, len - pos); pos = len;
			if(!fill()) return true;
		}
	}
};
static bool next_read(LineIn& in, bool fastq, Read& r) {
	std::string line;
	r = Read();
	bool got = false;
	if(fastq) {
		while((got = in.getline(line))) if(!line.empty() && line[0] == '@') break;
		if(!got || line.empty() || line[0] != '@') return false;
		std::string q, plus;
		if(!in.getline(r.seq) || !in.getline(plus) || !in.getline(q)) return false;
	}
	else {
		while((got = in.getline(line))) if(!line.empty() && line[0] == '>') break;
		if(!got || line.empty() || line[0] != '>') return false;
		while(in.peek() != EOF && in.peek() != '>') { std::string s; in.getline(s); while(!s.empty() && (s.back() == '\r' || s.back() == ' ')) s.pop_back(); r.seq += s; }
	}
	while(!line.empty() && line.back() == '\r') line.pop_back();
	/* id = the first word after the tag (operator>> skips blanks before it), desc = the rest of the line after the blanks that follow the id */
	auto blank = [](char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f' || c == '\r'; };
	size_t a = 1; while(a < line.size() && blank(line[a])) ++a;
	size_t e = a; while(e < line.size() && !blank(line[e])) ++e;
	r.id = line.substr(a, e - a);
	while(e < line.size() && blank(line[e])) ++e;
	r.desc = line.substr(e);
	while(!r.seq.empty() && (r.seq.back() == '\r' || r.seq.back() == '\n')) r.seq.pop_back();
	for(char& c : r.seq) c = (char) toupper((unsigned char) c);
	return true;
}
static std::string revcom(const std::string& s) { /* IUPACNucl complements (src/IUPACNucl.cpp:52-71) */
	std::string r(s.rbegin(), s.rend());
	for(char& c : r) switch(c) {
		case 'A': c = 'T'; break; case 'T': c = 'A'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'U': c = 'A'; break;
		case 'Y': c = 'R'; break; case 'R': c = 'Y'; break; case 'K': c = 'M'; break; case 'M': c = 'K'; break;
		case 'B': c = 'V'; break; case 'V': c = 'B'; break; case 'D': c = 'H'; break; case 'H': c = 'D'; break; default: break; }
	return r;
}
