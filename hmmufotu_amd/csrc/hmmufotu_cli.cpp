// hmmufotu-amd — command-line driver with the option surface of the reference's `hmmufotu`
// (src/hmmufotu.cpp:71-110, defaults :37-57) around the batched engine.  Host C++ only: option parsing,
// FASTA/FASTQ reading, the seed scans (hu_seed_index_*), strand auto-detection (:500-542), batching, TSV.
// -C/--chimera* run the segment check of src/hmmufotu.cpp:653-691 in a second batch (hu_chimera_batch).
// -a writes the aligned reads as FASTA (60 columns, description + ";csStart=..;csEnd=..;", :709-715); --align-only stops
// after the alignment.  Inputs may be gzip-compressed, outputs are when their name ends in .gz (zlib; no bz2).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <zlib.h>
#include "../../include/hmmufotu_amd.h"

struct Read { std::string id, desc, seq; };

/* line reader over zlib: plain and gzip-compressed inputs alike (the reference reads .gz / .bz2 through boost::iostreams) */
struct LineIn {
	gzFile f = nullptr; std::vector<char> buf; size_t pos = 0, len = 0; bool eof = false;
	bool open(const std::string& fn) { f = gzopen(fn.c_str(), "rb"); if(f) { gzbuffer(f, 1 << 20); buf.resize(1 << 20); } return f != nullptr; }
	~LineIn() { if(f) gzclose(f); }
	bool fill() { if(eof) return false; const int k = gzread(f, buf.data(), (unsigned) buf.size()); pos = 0; len = k > 0 ? (size_t) k : 0; if(k <= 0) eof = true; return k > 0; }
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
/* writer: gzip when the file name ends in .gz */
struct LineOut {
	gzFile z = nullptr; std::ofstream f; bool on = false;
	bool open(const std::string& fn) {
		on = true;
		if(fn.size() > 3 && fn.compare(fn.size() - 3, 3, ".gz") == 0) { z = gzopen(fn.c_str(), "wb"); return z != nullptr; }
		f.open(fn); return (bool) f;
	}
	void write(const char* p, size_t n) { if(z) { while(n) { const unsigned k = (unsigned) std::min<size_t>(n, 1u << 30); gzwrite(z, p, k); p += k; n -= k; } } else f.write(p, (std::streamsize) n); }
	void write(const std::string& s) { write(s.data(), s.size()); }
	~LineOut() { if(z) gzclose(z); }
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
	const size_t sp = line.find_first_of(" \t");
	r.id = line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
	if(sp != std::string::npos) r.desc = line.substr(sp + 1);
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
static void usage(const char* p) {
	std::cerr << "Usage:    " << p << "  <HmmUFOtu-DB> <READ-FILE1> [READ-FILE2] [options]\n"
		"Options:    -o FILE  -L|--seed-len INT [20]  -R INT [50]  --single  -s|--strand INT [0]  -t|--test INT [100]\n"
		"            -i|--ignore  -N INT [50]  -d|--max-diff DBL  -H|--max-height DBL  -e|--err DBL [20]\n"
		"            -m|--method unweighted|weighted  --ML  --prior uniform|height  --fmt fasta|fastq\n"
		"            -C|--chimera  --num-segment INT [2]  --chimera-err DBL [-e / --num-segment]  --chimera-lod DBL [0]\n"
		"            --chimera-out FILE  --chimera-info  -a FILE  --align-only\n"
		"            --batch INT [8192]  --gpu INT [0]  -v  -h|--help\n";
}
#define CHK(call) do { if((call) != HU_OK) { std::cerr << "Error: " << hu_last_error() << std::endl; return EXIT_FAILURE; } } while(0)

struct Packed { std::string bases; std::vector<int64_t> offs{0}; std::vector<int32_t> vp;
	void add(const std::string& s) { bases += s; offs.push_back((int64_t) bases.size()); }
	void clear() { bases.clear(); offs.assign(1, 0); vp.clear(); } int n() const { return (int) offs.size() - 1; } };

/* one batch of parsed reads on the host.  The main thread parses and runs the seed lookups of batch k + 1 while the
 * worker thread drives batch k through the engine and writes its lines (output stays in read order) */
struct Slot { Packed f, r; std::vector<std::string> ids, descs; };

int main(int argc, char** argv) {
	std::vector<std::string> pos; std::string outFn, fmt, method = "unweighted", prior = "uniform";
	int seedLen = 20, seedRegion = 50, strand = 0, nTest = 100, batch = 8192, gpu = 0, verbose = 0;
	bool single = false, checkChimera = false, chimeraInfo = false, alignOnly = false;
	std::string alnFn;
	int numSeg = 2; double chimeraErr = NAN, chimeraLod = 0; std::string chiOutFn;
	hu_opts o; hu_default_opts(&o);
	std::string cmd;
	for(int i = 0; i < argc; ++i) { cmd += argv[i]; cmd += i + 1 < argc ? " " : ""; }
	for(int i = 1; i < argc; ++i) {
		std::string a = argv[i];
		auto val = [&]() -> const char* { if(i + 1 >= argc) { std::cerr << "Error: option " << a << " needs a value\n"; exit(EXIT_FAILURE); } return argv[++i]; };
		if(a == "-h" || a == "--help") { usage(argv[0]); return EXIT_SUCCESS; }
		else if(a == "-o") outFn = val();
		else if(a == "-L" || a == "--seed-len") seedLen = atoi(val());
		else if(a == "-R") seedRegion = atoi(val());
		else if(a == "--single") single = true;
		else if(a == "-s" || a == "--strand") strand = atoi(val());
		else if(a == "-t" || a == "--test") nTest = atoi(val());
		else if(a == "-i" || a == "--ignore") o.ignore_orient = 1;
		else if(a == "-N") o.max_nseed = atoi(val());
		else if(a == "-d" || a == "--max-diff") o.max_diff = atof(val());
		else if(a == "-H" || a == "--max-height") o.max_height = atof(val());
		else if(a == "-e" || a == "--err") o.max_error = atof(val());
		else if(a == "-m" || a == "--method") method = val();
		else if(a == "--ML") o.only_ml = 1;
		else if(a == "--prior") prior = val();
		else if(a == "--fmt") fmt = val();
		else if(a == "-C" || a == "--chimera") checkChimera = true;
		else if(a == "--num-segment") numSeg = (int) atof(val());
		else if(a == "--chimera-err") chimeraErr = atof(val());
		else if(a == "--chimera-lod") chimeraLod = atof(val());
		else if(a == "--chimera-out") chiOutFn = val();
		else if(a == "--chimera-info") chimeraInfo = true;
		else if(a == "-a") alnFn = val();
		else if(a == "--align-only") alignOnly = true;
		else if(a == "--batch") batch = atoi(val());
		else if(a == "--gpu") gpu = atoi(val());
		else if(a == "-v") verbose++;
		else if(a == "-S" || a == "--seed" || a == "-p" || a == "--process") (void) val(); /* accepted, no effect: lookups are deterministic */
		else if(a[0] == '-' && a.size() > 1) { std::cerr << "Error: unknown option " << a << std::endl; usage(argv[0]); return EXIT_FAILURE; }
		else pos.push_back(a);
	}
	if(pos.size() < 2 || pos.size() > 3) { usage(argv[0]); return EXIT_FAILURE; }
	/* validation as src/hmmufotu.cpp:293-348 */
	if(seedLen < 15 || seedLen > 25) { std::cerr << "-L|--seed-len must be in range [15, 25]" << std::endl; return EXIT_FAILURE; }
	if(seedRegion < seedLen) { std::cerr << "-R cannot be smaller than -L" << std::endl; return EXIT_FAILURE; }
	if(strand < 0 || strand > 2) { std::cerr << "-s|--strand must be 0, 1, or 2" << std::endl; return EXIT_FAILURE; }
	if(o.max_nseed < 1 || o.max_nseed > 64) { std::cerr << "-N must be in range [1, 64]" << std::endl; return EXIT_FAILURE; }
	if(!(o.max_error >= 0)) { std::cerr << "-e|--err must be non-negative" << std::endl; return EXIT_FAILURE; }
	if(method != "unweighted" && method != "weighted") { std::cerr << "-m|--method must be either 'unweighted' or 'weighted'" << std::endl; return EXIT_FAILURE; }
	if(prior != "uniform" && prior != "height") { std::cerr << "--prior must be either 'uniform' or 'height'" << std::endl; return EXIT_FAILURE; }
	/* the chimera options only count with -C (src/hmmufotu.cpp:248-260); checks of :325-340 */
	if(!checkChimera) { chimeraInfo = false; chiOutFn.clear(); numSeg = 2; chimeraErr = NAN; chimeraLod = 0; }
	if(numSeg < 2 || numSeg > 6) { std::cerr << "--num-segment must be in [2, 6]" << std::endl; return EXIT_FAILURE; }
	if(numSeg % 2) { std::cerr << "--num-segment must be an even number" << std::endl; return EXIT_FAILURE; }
	if(std::isnan(chimeraErr)) chimeraErr = o.max_error / numSeg;
	if(!(chimeraErr > 0)) { std::cerr << "--chimera-err must be positive" << std::endl; return EXIT_FAILURE; }
	if(!(chimeraLod >= 0)) { std::cerr << "--chimera-lod must be non-negative" << std::endl; return EXIT_FAILURE; }
	hu_chimera_opts co; co.num_seg = numSeg; co.reserved = 0; co.max_chimera_error = chimeraErr; co.min_chimera_lod = chimeraLod;
	o.weighted = method == "weighted"; o.prior = prior == "height" ? HU_PRIOR_HEIGHT : HU_PRIOR_UNIFORM;
	const bool paired = pos.size() == 3;
	o.align_mode = (paired || !single) ? HU_MODE_GLOBAL : HU_MODE_NGCL;               /* src/hmmufotu.cpp:358 */
	std::string fwdFn = pos[1], revFn = paired ? pos[2] : "";
	auto is_fastq = [&](std::string fn) {
		if(!fmt.empty()) return fmt == "fastq";
		if(fn.size() > 3 && fn.compare(fn.size() - 3, 3, ".gz") == 0) fn.resize(fn.size() - 3);
		return (fn.size() > 6 && fn.compare(fn.size() - 6, 6, ".fastq") == 0) || (fn.size() > 3 && fn.compare(fn.size() - 3, 3, ".fq") == 0);
	};

	hu_db* db = nullptr;
	CHK(hu_db_load((pos[0] + ".hmm").c_str(), (pos[0] + ".ptu").c_str(), gpu, &db));
	int32_t K, L, nNodes, root; int64_t hbm;
	CHK(hu_db_info(db, &K, &L, &nNodes, &root, &hbm));
	if(verbose) std::cerr << "database loaded: K=" << K << " csLen=" << L << " nodes=" << nNodes << " HBM=" << hbm / 1e9 << " GB" << std::endl;
	std::vector<int32_t> parent(nNodes), p2cs(K + 1); std::vector<int8_t> seq((size_t) nNodes * L);
	CHK(hu_db_get_tree(db, parent.data(), nullptr, seq.data(), nullptr));
	CHK(hu_db_get_profile(db, nullptr, nullptr, nullptr, p2cs.data(), nullptr, nullptr));
	hu_seed_index* ix = nullptr;
	CHK(hu_seed_index_create(nNodes, L, parent.data(), seq.data(), K, p2cs.data(), seedLen, &ix));
	if(verbose) std::cerr << "seed index built: " << hu_seed_index_size(ix) << " distinct " << seedLen << "-mers" << std::endl;
	hu_batch* gb = nullptr; hu_batch* wb = nullptr;
	CHK(hu_batch_create(db, batch, &gb));
	if(checkChimera) CHK(hu_batch_create(db, batch, &wb));

	/* strand auto-detection on the first nTest reads by alignment cost (src/hmmufotu.cpp:500-542) */
	if(strand == 0) {
		LineIn tin;
		if(!tin.open(fwdFn)) { std::cerr << "Unable to test forward seq file '" << fwdFn << "'" << std::endl; return EXIT_FAILURE; }
		Packed f, r; Read rd;
		for(int i = 0; i < nTest && i < batch && next_read(tin, is_fastq(fwdFn), rd); ++i) { f.add(rd.seq); r.add(revcom(rd.seq)); }
		double fwdScore = 0, revScore = 0;
		std::vector<hu_align_rec> af(f.n()), ar(f.n());
		if(f.n() > 0) {
			f.vp.resize((size_t) f.n() * 12); r.vp.resize((size_t) f.n() * 12);
			CHK(hu_seed_index_lookup(ix, f.n(), f.bases.data(), f.offs.data(), seedRegion, o.align_mode, f.vp.data()));
			CHK(hu_seed_index_lookup(ix, r.n(), r.bases.data(), r.offs.data(), seedRegion, o.align_mode, r.vp.data()));
			CHK(hu_batch_set_reads(gb, f.n(), f.bases.data(), f.offs.data(), f.vp.data(), nullptr, nullptr, nullptr));
			CHK(hu_align_batch(gb, &o)); CHK(hu_batch_get_alignments(gb, af.data(), nullptr, nullptr, 0));
			CHK(hu_batch_set_reads(gb, r.n(), r.bases.data(), r.offs.data(), r.vp.data(), nullptr, nullptr, nullptr));
			CHK(hu_align_batch(gb, &o)); CHK(hu_batch_get_alignments(gb, ar.data(), nullptr, nullptr, 0));
			for(int i = 0; i < f.n(); ++i) {
				const double cf = af[i].status == HU_READ_OK ? af[i].cost : INFINITY, cr = ar[i].status == HU_READ_OK ? ar[i].cost : INFINITY;
				if(cf < cr) fwdScore++; else revScore++;
			}
		}
		if(fwdScore >= (fwdScore + revScore) * 0.9) strand = 1;
		else if(revScore >= (fwdScore + revScore) * 0.9) strand = 2;
		else { std::cerr << "Failed to determine read strandness. Try larger -t|--test or determine manually" << std::endl; return EXIT_FAILURE; }
		if(verbose) std::cerr << "Read strand determined as " << strand << std::endl;
	}
	if(strand == 2 && paired) std::swap(fwdFn, revFn);
	LineIn fin, rin;
	if(!fin.open(fwdFn)) { std::cerr << "Unable to open forward seq file '" << fwdFn << "'" << std::endl; return EXIT_FAILURE; }
	if(paired) { if(!rin.open(revFn)) { std::cerr << "Unable to open reverse seq file '" << revFn << "'" << std::endl; return EXIT_FAILURE; } }
	LineOut fout; if(!outFn.empty() && !fout.open(outFn)) { std::cerr << "Unable to write to '" << outFn << "'" << std::endl; return EXIT_FAILURE; }
	auto out = [&](const std::string& t) { if(fout.on) fout.write(t); else std::cout.write(t.data(), (std::streamsize) t.size()); };
	const char* header = chimeraInfo ? hu_tsv_header_chimera() : hu_tsv_header();
	const std::string preamble = std::string("# HmmUFOtu v1.5.1 taxonomy assignment generated by ") + argv[0] + "\n# command: " + cmd + "\n" + header + "\n";
	out(preamble);
	LineOut chiOut;
	if(!chiOutFn.empty()) {
		if(!chiOut.open(chiOutFn)) { std::cerr << "Unable to write to '" << chiOutFn << "'" << std::endl; return EXIT_FAILURE; }
		chiOut.write(preamble);
	}
	LineOut alnOut;
	if(!alnFn.empty() && !alnOut.open(alnFn)) { std::cerr << "Unable to write to align file '" << alnFn << "'" << std::endl; return EXIT_FAILURE; }
	std::vector<hu_chimera_rec> chi;
	std::vector<hu_align_rec> alnRecs; std::vector<char> alnRows;
	long flagged = 0;

	std::vector<const char*> annos(nNodes);
	for(int i = 0; i < nNodes; ++i) annos[i] = hu_db_get_annotation(db, i);
	long total = 0, placed = 0;
	auto process = [&](Slot& sl) -> int { /* worker thread: engine + output of one batch */
		Packed& f = sl.f; Packed& r = sl.r; std::vector<std::string>& ids = sl.ids; std::vector<std::string>& descs = sl.descs;
		const int n = f.n();
		int rc;
		if((rc = hu_batch_set_reads(gb, n, f.bases.data(), f.offs.data(), f.vp.data(), paired ? r.bases.data() : nullptr, paired ? r.offs.data() : nullptr,
				paired ? r.vp.data() : nullptr)) != HU_OK) return rc;
		if(checkChimera) { /* common seeds first, the check on them, then the ordinary estimate/filter/place (src/hmmufotu.cpp:643-733) */
			chi.resize((size_t) n);
			if((rc = hu_align_batch(gb, &o)) != HU_OK || (rc = hu_seed_batch(gb, &o)) != HU_OK) return rc;
			if((rc = hu_chimera_batch(gb, wb, &o, &co, chi.data())) != HU_OK) return rc;
			if(!alignOnly && ((rc = hu_estimate_batch(gb, &o)) != HU_OK || (rc = hu_filter_batch(gb, &o)) != HU_OK || (rc = hu_place_batch(gb, &o)) != HU_OK ||
					(rc = hu_finish_batch(gb, &o)) != HU_OK)) return rc;
			for(const hu_chimera_rec& c : chi) flagged += c.is_chimera;
		}
		else if((rc = alignOnly ? hu_align_batch(gb, &o) : hu_assign_batch(gb, &o)) != HU_OK) return rc;
		const int mainKind = alignOnly ? 2 : 0;
		std::vector<const char*> pid(n), pdesc(n);
		for(int i = 0; i < n; ++i) { pid[i] = ids[i].c_str(); pdesc[i] = descs[i].c_str(); }
		const hu_chimera_rec* cp = checkChimera ? chi.data() : nullptr;
		const int64_t need = hu_batch_format_tsv_chimera(gb, pid.data(), pdesc.data(), annos.data(), cp, chimeraInfo, mainKind, nullptr, 0);
		if(need < 0) return (int) need;
		std::string buf((size_t) need, '\0');
		hu_batch_format_tsv_chimera(gb, pid.data(), pdesc.data(), annos.data(), cp, chimeraInfo, mainKind, &buf[0], need);
		out(buf);
		if(alnOut.on) { /* aligned reads that are not chimeras (src/hmmufotu.cpp:709-715; SeqIO::writeFastaSeq, 60 columns) */
			alnRecs.resize((size_t) n); alnRows.resize((size_t) n * L);
			if((rc = hu_batch_get_alignments(gb, alnRecs.data(), alnRows.data(), nullptr, 0)) != HU_OK) return rc;
			for(int i = 0; i < n; ++i) {
				if(alnRecs[i].status != HU_READ_OK || (cp && cp[i].is_chimera)) continue;
				const std::string d = descs[i] + ";csStart=" + std::to_string(alnRecs[i].cs_start) + ";csEnd=" + std::to_string(alnRecs[i].cs_end) + ";";
				alnOut.write(">" + ids[i] + " " + d + "\n");
				for(int c = 0; c < L; c += 60) { alnOut.write(&alnRows[(size_t) i * L + c], (size_t) std::min(60, L - c)); alnOut.write("\n", 1); }
			}
		}
		if(chiOut.on) {
			const int64_t cneed = hu_batch_format_tsv_chimera(gb, pid.data(), pdesc.data(), annos.data(), cp, chimeraInfo, 1, nullptr, 0);
			if(cneed < 0) return (int) cneed;
			std::string cbuf((size_t) cneed, '\0');
			hu_batch_format_tsv_chimera(gb, pid.data(), pdesc.data(), annos.data(), cp, chimeraInfo, 1, &cbuf[0], cneed);
			chiOut.write(cbuf);
		}
		for(char c : buf) if(c == '\n') placed++;
		total += n;
		return HU_OK;
	};
	std::mutex mu; std::condition_variable cv; std::unique_ptr<Slot> pending; bool finished = false; int werr = HU_OK; std::string wmsg;
	std::thread worker([&] {
		for(;;) {
			std::unique_ptr<Slot> sl;
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return pending || finished; }); if(!pending) return; sl = std::move(pending); }
			cv.notify_all();
			bool ok; { std::lock_guard<std::mutex> lk(mu); ok = werr == HU_OK; }
			if(ok) { const int rc = process(*sl); if(rc != HU_OK) { std::lock_guard<std::mutex> lk(mu); werr = rc; wmsg = hu_last_error(); } }
		}
	});
	struct Joiner { std::mutex& mu; std::condition_variable& cv; bool& fin; std::thread& t;
		~Joiner() { { std::lock_guard<std::mutex> lk(mu); fin = true; } cv.notify_all(); if(t.joinable()) t.join(); } } joiner{mu, cv, finished, worker};
	auto submit = [&](std::unique_ptr<Slot>& sl) -> int { /* main thread: seed lookups, then hand the batch to the worker */
		const int n = sl->f.n();
		if(n == 0) return HU_OK;
		int rc;
		sl->f.vp.assign((size_t) n * 12, 0);
		if((rc = hu_seed_index_lookup(ix, n, sl->f.bases.data(), sl->f.offs.data(), seedRegion, o.align_mode, sl->f.vp.data())) != HU_OK) return rc;
		if(paired) { sl->r.vp.assign((size_t) n * 12, 0); if((rc = hu_seed_index_lookup(ix, n, sl->r.bases.data(), sl->r.offs.data(), seedRegion, o.align_mode, sl->r.vp.data())) != HU_OK) return rc; }
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [&] { return !pending; });
		if(werr != HU_OK) { std::cerr << "Error: " << wmsg << std::endl; return werr; }
		pending = std::move(sl);
		lk.unlock(); cv.notify_all();
		sl.reset(new Slot());
		return HU_OK;
	};
	std::unique_ptr<Slot> cur(new Slot());
	Read a, b;
	while(next_read(fin, is_fastq(fwdFn), a) && (!paired || next_read(rin, is_fastq(revFn), b))) {
		std::string s = a.seq;
		if(strand == 2 && !paired) s = revcom(s);          /* wrong strand for single-strand reads (src/hmmufotu.cpp:617-618) */
		cur->f.add(s); cur->ids.push_back(a.id); cur->descs.push_back(a.desc);
		if(paired) cur->r.add(revcom(b.seq));              /* mates are reverse-complemented at read time (:609) */
		if(cur->f.n() == batch) { const int rc = submit(cur); if(rc != HU_OK) { if(rc != werr) std::cerr << "Error: " << hu_last_error() << std::endl; return EXIT_FAILURE; } }
	}
	{ const int rc = submit(cur); if(rc != HU_OK) { if(rc != werr) std::cerr << "Error: " << hu_last_error() << std::endl; return EXIT_FAILURE; } }
	{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !pending; }); finished = true; }
	cv.notify_all();
	worker.join();
	if(werr != HU_OK) { std::cerr << "Error: " << wmsg << std::endl; return EXIT_FAILURE; }
	if(verbose) std::cerr << total << " reads processed, " << placed << " assigned" << (checkChimera ? ", " + std::to_string(flagged) + " flagged as chimera" : std::string()) << std::endl;
	if(wb) hu_batch_destroy(wb);
	hu_batch_destroy(gb); hu_seed_index_destroy(ix); hu_db_destroy(db);
	return EXIT_SUCCESS;
}
