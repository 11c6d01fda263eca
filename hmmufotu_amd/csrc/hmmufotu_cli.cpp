// hmmufotu-amd — command-line driver with the option surface of the reference's `hmmufotu`
// (src/hmmufotu.cpp:71-110, defaults :37-57) around the batched engine.  Host C++ only: option parsing,
// FASTA/FASTQ reading, the seed scans (hu_seed_index_*), strand auto-detection (:500-542), batching, TSV.
// -C/--chimera* run the segment check of src/hmmufotu.cpp:653-691 in a second batch (hu_chimera_batch).
// -a writes the aligned reads as FASTA (60 columns, description + ";csStart=..;csEnd=..;", :709-715); --align-only stops
// after the alignment.  Inputs may be gzip- or bzip2-compressed (.bz2 by name), outputs are when their name ends in .gz / .bz2 (zlib; libbz2 bound at run time).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <zlib.h>
#include "../../include/hmmufotu_amd.h"
#include "hu_reads_io.h"

/* writer: gzip when the file name ends in .gz, bzip2 when it ends in .bz2 */
struct LineOut {
	gzFile z = nullptr; void* bz = nullptr; std::ofstream f; bool on = false;
	bool open(const std::string& fn) {
		on = true;
		if(fn.size() > 3 && fn.compare(fn.size() - 3, 3, ".gz") == 0) { z = gzopen(fn.c_str(), "wb"); return z != nullptr; }
		if(HuBz2::named(fn)) { if(HuBz2::get().ok()) bz = HuBz2::get().bzopen(fn.c_str(), "wb"); return bz != nullptr; }
		f.open(fn); return (bool) f;
	}
	void write(const char* p, size_t n) {
		if(z) { while(n) { const unsigned k = (unsigned) std::min<size_t>(n, 1u << 30); gzwrite(z, p, k); p += k; n -= k; } }
		else if(bz) { while(n) { const int k = (int) std::min<size_t>(n, 1u << 30); HuBz2::get().bzwrite(bz, const_cast<char*>(p), k); p += k; n -= (size_t) k; } }
		else f.write(p, (std::streamsize) n);
	}
	void write(const std::string& s) { write(s.data(), s.size()); }
	~LineOut() { if(z) gzclose(z); if(bz) HuBz2::get().bzclose(bz); }
};

static void usage(const char* p) {
	std::cerr << "Usage:    " << p << "  <HmmUFOtu-DB> <READ-FILE1> [READ-FILE2] [options]\n"
		"Options:    -o FILE  -L|--seed-len INT [20]  -R INT [50]  --single  -s|--strand INT [0]  -t|--test INT [100]\n"
		"            -i|--ignore  -N INT [50]  -d|--max-diff DBL  -H|--max-height DBL  -e|--err DBL [20]\n"
		"            -m|--method unweighted|weighted  --ML  --prior uniform|height  --fmt fasta|fastq\n"
		"            --fix-root-loglik  rank candidates by the intended root log-likelihood (the reference returns a constant)\n"
		"            --no-csfm        build the seed index from the .ptu even when <DB>.csfm exists\n"
		"            -C|--chimera  --num-segment INT [2]  --chimera-err DBL [-e / --num-segment]  --chimera-lod DBL [0]\n"
		"            --chimera-out FILE  --chimera-info  -a FILE  --align-only\n"
		"            --batch INT [8192]  --gpu INT [0] first device  --gpus INT [1] devices, one database replica each\n"
		"            --inflight INT [3] batches in flight per device  -v  --version  -h|--help\n"
		"            --seed-order reference|stable  which of the nodes tying at the cut-off distance become seeds: the reference binary's own choice\n"
		"                             [reference, default] — the first -N of libstdc++'s std::sort on dist alone, reproduced on the device — or\n"
		"                             (dist, node id) [stable]: independent of the sort's tie permutation and ~25 % faster\n"
		"            --col-windows INT [1]  hold the database as INT column windows (messages of a window's CS columns only; window i on device\n"
		"                             --gpu + i mod --gpus) instead of whole replicas: for databases beyond one GPU's memory.  Reads are routed\n"
		"                             by their seeds and re-routed once by their alignment region.  --win-overlap INT [3200] columns shared by\n"
		"                             neighbouring windows: at least the widest alignment region to be expected\n"
		"            -S|--seed INT    a seed hit drawn from all its occurrences (CSFMIndex::locateOne) with this seed; without it the first occurrence\n"
		"                             (locateFirst).  A run repeats at any thread count.  -p|--process INT is accepted and has no effect\n";
}
#define CHK(call) do { if((call) != HU_OK) { std::cerr << "Error: " << hu_last_error() << std::endl; return EXIT_FAILURE; } } while(0)

struct Packed { std::string bases; std::vector<int64_t> offs{0}; std::vector<int32_t> vp;
	void add(const std::string& s) { bases += s; offs.push_back((int64_t) bases.size()); }
	void clear() { bases.clear(); offs.assign(1, 0); vp.clear(); } int n() const { return (int) offs.size() - 1; } };

/* one batch of parsed reads on the host.  The main thread parses and runs the seed lookups of batch k + 1 while the
 * worker thread drives batch k through the engine and writes its lines (output stays in read order) */
struct Slot { Packed f, r; std::vector<std::string> ids, descs; };

/* the seed scans of one packed batch: first occurrence, or (-S) a drawn one; stream 0 = forward reads, 1 = mates, 2 / 3 = the strand test */
static int seed_lookup(const hu_seed_index* ix, Packed& p, int seedRegion, int mode, bool random, uint64_t seed, int stream, int64_t firstRead) {
	p.vp.assign((size_t) p.n() * 12, 0);
	if(!random) return hu_seed_index_lookup(ix, p.n(), p.bases.data(), p.offs.data(), seedRegion, mode, p.vp.data());
	return hu_seed_index_lookup_random(ix, p.n(), p.bases.data(), p.offs.data(), seedRegion, mode, seed + 0x9e3779b97f4a7c15ull * (uint64_t) stream, firstRead, p.vp.data());
}

int main(int argc, char** argv) {
	std::vector<std::string> pos; std::string outFn, fmt, method = "unweighted", prior = "uniform";
	int seedLen = 20, seedRegion = 50, strand = 0, nTest = 100, batch = 8192, gpu = 0, nGpus = 1, inflight = 3, verbose = 0, colWindows = 1; long winOverlap = 3200;
	bool single = false, checkChimera = false, chimeraInfo = false, alignOnly = false, noCsfm = false, randomHits = false;
	uint64_t hitSeed = 0; std::string seedOrder = "reference";
	std::string alnFn;
	int numSeg = 2; double chimeraErr = NAN, chimeraLod = 0; std::string chiOutFn;
	hu_opts o; hu_default_opts(&o);
	std::string cmd;
	for(int i = 0; i < argc; ++i) { cmd += argv[i]; cmd += i + 1 < argc ? " " : ""; }
	for(int i = 1; i < argc; ++i) {
		std::string a = argv[i];
		auto val = [&]() -> const char* { if(i + 1 >= argc) { std::cerr << "Error: option " << a << " needs a value\n"; exit(EXIT_FAILURE); } return argv[++i]; };
		if(a == "-h" || a == "--help") { usage(argv[0]); return EXIT_SUCCESS; }
		else if(a == "--version") { std::cerr << argv[0] << ": v1.5.1\nPackage: HmmUFOtu v1.5.1 (file formats and assignment semantics; hmmufotu_amd engine for gfx950)" << std::endl; return EXIT_SUCCESS; }   /* printVersion, src/util/ProgEnv.cpp:19-22 */
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
		else if(a == "--fix-root-loglik") o.fix_root_loglik = 1;      /* not in the reference: SURVEY.md F4 / H2 */
		else if(a == "--no-csfm") noCsfm = true;                      /* index from the .ptu's leaf rows even when <DB>.csfm exists */
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
		else if(a == "--gpus") nGpus = atoi(val());
		else if(a == "--inflight") inflight = atoi(val());
		else if(a == "--col-windows") colWindows = atoi(val());
		else if(a == "--win-overlap") winOverlap = atol(val());
		else if(a == "-v") verbose++;
		else if(a == "-S" || a == "--seed") { hitSeed = (uint64_t) strtoull(val(), nullptr, 10); randomHits = true; }   /* src/hmmufotu.cpp:262-266 */
		else if(a == "--seed-order") seedOrder = val();
		else if(a == "-p" || a == "--process") (void) val(); /* accepted, no effect: host threads follow the batches in flight */
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
	if(seedOrder != "stable" && seedOrder != "reference") { std::cerr << "--seed-order must be either 'stable' or 'reference'" << std::endl; return EXIT_FAILURE; }
	o.seed_order = seedOrder == "reference" ? HU_SEED_ORDER_LIBSTDCXX : HU_SEED_ORDER_STABLE;
	if(prior != "uniform" && prior != "height") { std::cerr << "--prior must be either 'uniform' or 'height'" << std::endl; return EXIT_FAILURE; }
	/* the chimera options only count with -C (src/hmmufotu.cpp:248-260); checks of :325-340 */
	if(!checkChimera) { chimeraInfo = false; chiOutFn.clear(); numSeg = 2; chimeraErr = NAN; chimeraLod = 0; }
	if(nGpus < 1 || nGpus > 64 || inflight < 1 || inflight > 16 || batch < 1) { std::cerr << "--gpus must be in [1, 64], --inflight in [1, 16], --batch positive" << std::endl; return EXIT_FAILURE; }
	if(colWindows < 1 || colWindows > 64 || winOverlap < 0) { std::cerr << "--col-windows must be in [1, 64], --win-overlap non-negative" << std::endl; return EXIT_FAILURE; }
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
		else if(HuBz2::named(fn)) fn.resize(fn.size() - 4);
		return (fn.size() > 6 && fn.compare(fn.size() - 6, 6, ".fastq") == 0) || (fn.size() > 3 && fn.compare(fn.size() - 3, 3, ".fq") == 0);
	};

	if(hu_device_count() < (getenv("HU_CLI_SHARE_GPU") ? 1 : nGpus) + gpu) { std::cerr << "Error: " << nGpus << " device(s) from index " << gpu << " asked for, " << hu_device_count() << " gfx950 device(s) visible" << std::endl; return EXIT_FAILURE; }
	/* one database replica per device (src/hmmufotu.cpp:457-494 loads the one shared copy), loaded side by side — or, with --col-windows W, W column
	 * windows of ONE database spread over the devices (hu_db_load_window): dbs[k] then holds window k */
	const int nDb = colWindows > 1 ? colWindows : nGpus;
	std::vector<hu_db*> dbs(nDb, nullptr);
	std::vector<hu_window> wins((size_t) colWindows);
	if(colWindows > 1) {
		int32_t K0 = 0, L0 = 0, n0 = 0, r0 = 0;
		CHK(hu_files_parse((pos[0] + ".hmm").c_str(), (pos[0] + ".ptu").c_str(), &K0, &L0, &n0, &r0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0));
		CHK(hu_windows_plan(L0, colWindows, std::min<long>(winOverlap, L0 - 1), wins.data()));
	}
	{
		std::vector<std::thread> lt; std::vector<int> lrc(nDb, HU_OK); std::vector<std::string> lmsg(nDb);
		const bool share = getenv("HU_CLI_SHARE_GPU") != nullptr;    /* rehearsal of --gpus N on a box with one device: every replica on --gpu */
		for(int g = 0; g < nDb; ++g) lt.emplace_back([&, g] {
			const int dev = share ? gpu : gpu + g % nGpus;
			lrc[g] = colWindows > 1 ? hu_db_load_window((pos[0] + ".hmm").c_str(), (pos[0] + ".ptu").c_str(), dev, wins[g].win_start, wins[g].win_len, &dbs[g])
			                        : hu_db_load((pos[0] + ".hmm").c_str(), (pos[0] + ".ptu").c_str(), dev, &dbs[g]);
			if(lrc[g] != HU_OK) lmsg[g] = hu_last_error(); });
		for(auto& t : lt) t.join();
		for(int g = 0; g < nDb; ++g) if(lrc[g] != HU_OK) { std::cerr << "Error: " << lmsg[g] << std::endl; return EXIT_FAILURE; }
	}
	hu_db* db = dbs[0];
	int32_t K, L, nNodes, root; int64_t hbm;
	CHK(hu_db_info(db, &K, &L, &nNodes, &root, &hbm));
	if(verbose) std::cerr << "database loaded on " << nGpus << " device(s): K=" << K << " csLen=" << L << " nodes=" << nNodes << " HBM=" << hbm / 1e9 << " GB each" << std::endl;
	if(verbose && colWindows > 1) { std::cerr << colWindows << " column windows:"; for(const hu_window& w : wins) std::cerr << " [" << w.win_start << ", " << w.win_start + w.win_len << ")"; std::cerr << std::endl; }
	std::vector<int32_t> parent(nNodes), p2cs(K + 1); std::vector<int8_t> seq((size_t) nNodes * L);
	CHK(hu_db_get_tree(db, parent.data(), nullptr, seq.data(), nullptr));
	CHK(hu_db_get_profile(db, nullptr, nullptr, nullptr, p2cs.data(), nullptr, nullptr));
	hu_seed_index* ix = nullptr;
	/* the database's own index file when it is there (<DB>.csfm, as the reference requires it): seeds in the file's suffix order,
	 * first hit = CSFMIndex::locateFirst; else the same index from the leaf rows of the .ptu */
	bool fromCsfm = false;
	if(!noCsfm) { FILE* t = fopen((pos[0] + ".csfm").c_str(), "rb"); if(t) { fclose(t); fromCsfm = true; } }
	if(fromCsfm) CHK(hu_seed_index_load_csfm((pos[0] + ".csfm").c_str(), K, p2cs.data(), seedLen, &ix));
	else CHK(hu_seed_index_create(nNodes, L, parent.data(), seq.data(), K, p2cs.data(), seedLen, &ix));
	{ std::vector<int8_t>().swap(seq); }
	if(verbose) { int64_t np = 0; const int64_t by = hu_seed_index_bytes(ix, &np); std::cerr << (fromCsfm ? "seed index read from the .csfm: " : "seed index built: ") << hu_seed_index_size(ix) << " distinct " << seedLen << "-mers at " << np << " positions, " << by / 1e6 << " MB" << std::endl; }
	hu_batch* gb = nullptr;
	CHK(hu_batch_create(db, std::min(batch, std::max(nTest, 1)), &gb));       /* strand detection only */

	/* strand auto-detection on the first nTest reads by alignment cost (src/hmmufotu.cpp:500-542) */
	if(strand == 0) {
		LineIn tin;
		if(!tin.open(fwdFn)) { std::cerr << "Unable to test forward seq file '" << fwdFn << "'" << std::endl; return EXIT_FAILURE; }
		Packed f, r; Read rd;
		for(int i = 0; i < nTest && i < batch && next_read(tin, is_fastq(fwdFn), rd); ++i) { f.add(rd.seq); r.add(revcom(rd.seq)); }
		double fwdScore = 0, revScore = 0;
		std::vector<hu_align_rec> af(f.n()), ar(f.n());
		if(f.n() > 0) {
			CHK(seed_lookup(ix, f, seedRegion, o.align_mode, randomHits, hitSeed, 2, 0));
			CHK(seed_lookup(ix, r, seedRegion, o.align_mode, randomHits, hitSeed, 3, 0));
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
	hu_batch_destroy(gb); gb = nullptr;
	std::vector<const char*> annos(nNodes);
	for(int i = 0; i < nNodes; ++i) annos[i] = hu_db_get_annotation(db, i);

	/* Four stages, every one on threads of its own, batches numbered in read order:
	 *   reader (this thread)   parses FASTA / FASTQ into batches
	 *   seeder                 the seed scans of alignSeq for a batch (hu_seed_index_lookup, itself multi-threaded)
	 *   engine workers         nGpus x inflight of them, each with its own hu_batch (and work batch for -C) on device w % nGpus:
	 *                          the batch through the engine, then its lines formatted into strings (65 MB per 8,192 reads: the
	 *                          formatting of several batches runs side by side)
	 *   writer                 writes the finished batches in read order
	 * The reference: one OpenMP task per read, lines in completion order under a critical section (src/hmmufotu.cpp:603-751). */
	struct Done { std::string mainTxt, chiTxt, alnTxt; long n = 0, placed = 0, flagged = 0; };
	struct Pipe {
		std::mutex mu; std::condition_variable cv;
		std::deque<std::pair<long, std::unique_ptr<Slot>>> parsed, seeded;
		std::map<long, Done> done;
		bool readerDone = false, seederDone = false; int workersLeft = 0;
		int err = HU_OK; std::string msg;
		void fail(int rc, const std::string& m) { std::lock_guard<std::mutex> lk(mu); if(err == HU_OK) { err = rc; msg = m; } cv.notify_all(); }
	} P;
	const auto tLoop = std::chrono::steady_clock::now();
	/* busy seconds per stage (reported with -v): which stage bounds the pipeline */
	std::atomic<long long> usParse{0}, usSeed{0}, usEngine{0}, usFormat{0}, usWrite{0};
	auto usSince = [](std::chrono::steady_clock::time_point t) { return (long long) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count(); };
	/* ordinary mode: worker w drives one batch object on replica w % nGpus.  Column-window mode: worker w is a LANE with one batch object per window
	 * (batch object w * colWindows + k on window k): it splits a batch of reads by window, runs the parts, re-routes, and puts the lines back in read order */
	const int nWorkers = (colWindows > 1 ? 1 : nGpus) * inflight;
	const size_t depth = (size_t) nWorkers + 2;
	P.workersLeft = nWorkers;
	const int perWorker = colWindows > 1 ? colWindows : 1;
	std::vector<hu_batch*> wgb((size_t) nWorkers * perWorker, nullptr), wwb((size_t) nWorkers * perWorker, nullptr);
	for(int w = 0; w < nWorkers * perWorker; ++w) {
		hu_db* d = colWindows > 1 ? dbs[w % colWindows] : dbs[w % nGpus];
		CHK(hu_batch_create(d, batch, &wgb[w])); if(checkChimera) CHK(hu_batch_create(d, batch, &wwb[w]));
	}

	struct PerRead { std::vector<int64_t> mainLen, chiLen, alnLen; std::vector<hu_align_rec> recs; };      /* column-window mode: where each read's lines are in the part's text */
	auto process = [&](int w, Slot& sl, Done& dn, PerRead* pr = nullptr) -> int { /* engine worker: one batch through the engine, then its text */
		hu_batch* gb = wgb[w]; hu_batch* wb = wwb[w];
		Packed& f = sl.f; Packed& r = sl.r; std::vector<std::string>& ids = sl.ids; std::vector<std::string>& descs = sl.descs;
		const int n = f.n();
		int rc;
		std::vector<hu_chimera_rec> chi;
		auto tE = std::chrono::steady_clock::now();
		if((rc = hu_batch_set_reads(gb, n, f.bases.data(), f.offs.data(), f.vp.data(), paired ? r.bases.data() : nullptr, paired ? r.offs.data() : nullptr,
				paired ? r.vp.data() : nullptr)) != HU_OK) return rc;
		if(checkChimera) { /* common seeds first, the check on them, then the ordinary estimate/filter/place (src/hmmufotu.cpp:643-733) */
			chi.resize((size_t) n);
			if((rc = hu_align_batch(gb, &o)) != HU_OK || (rc = hu_seed_batch(gb, &o)) != HU_OK) return rc;
			if((rc = hu_chimera_batch(gb, wb, &o, &co, chi.data())) != HU_OK) return rc;
			if(!alignOnly && ((rc = hu_estimate_batch(gb, &o)) != HU_OK || (rc = hu_filter_batch(gb, &o)) != HU_OK || (rc = hu_place_batch(gb, &o)) != HU_OK ||
					(rc = hu_finish_batch(gb, &o)) != HU_OK)) return rc;
			for(const hu_chimera_rec& c : chi) dn.flagged += c.is_chimera;
		}
		else if((rc = alignOnly ? hu_align_batch(gb, &o) : hu_assign_batch(gb, &o)) != HU_OK) return rc;
		usEngine += usSince(tE);
		auto tF = std::chrono::steady_clock::now();
		const int mainKind = alignOnly ? 2 : 0;
		std::vector<const char*> pid(n), pdesc(n);
		for(int i = 0; i < n; ++i) { pid[i] = ids[i].c_str(); pdesc[i] = descs[i].c_str(); }
		const hu_chimera_rec* cp = checkChimera ? chi.data() : nullptr;
		const char* txt = nullptr;
		const int64_t need = hu_batch_format_tsv_ptr(gb, pid.data(), pdesc.data(), annos.data(), cp, chimeraInfo, mainKind, &txt);
		if(need < 0) return (int) need;
		dn.mainTxt.assign(txt, (size_t) need);
		if(pr) {
			pr->mainLen.assign((size_t) n, 0); pr->recs.resize((size_t) n);
			if((rc = hu_batch_tsv_line_lengths(gb, pr->mainLen.data())) != HU_OK || (rc = hu_batch_get_alignments(gb, pr->recs.data(), nullptr, nullptr, 0)) != HU_OK) return rc;
			pr->alnLen.assign((size_t) n, 0); pr->chiLen.assign((size_t) n, 0);
		}
		if(alnOut.on) { /* aligned reads that are not chimeras (src/hmmufotu.cpp:709-715; SeqIO::writeFastaSeq, 60 columns) */
			std::vector<hu_align_rec> alnRecs((size_t) n); std::vector<char> alnRows((size_t) n * L);
			if((rc = hu_batch_get_alignments(gb, alnRecs.data(), alnRows.data(), nullptr, 0)) != HU_OK) return rc;
			for(int i = 0; i < n; ++i) {
				if(alnRecs[i].status != HU_READ_OK || (cp && cp[i].is_chimera)) continue;
				const size_t at0 = dn.alnTxt.size();
				dn.alnTxt += ">" + ids[i] + " " + descs[i] + ";csStart=" + std::to_string(alnRecs[i].cs_start) + ";csEnd=" + std::to_string(alnRecs[i].cs_end) + ";\n";
				for(int c = 0; c < L; c += 60) { dn.alnTxt.append(&alnRows[(size_t) i * L + c], (size_t) std::min(60, L - c)); dn.alnTxt += '\n'; }
				if(pr) pr->alnLen[(size_t) i] = (int64_t)(dn.alnTxt.size() - at0);
			}
		}
		if(chiOut.on) {
			const char* ctxt = nullptr;
			const int64_t cneed = hu_batch_format_tsv_ptr(gb, pid.data(), pdesc.data(), annos.data(), cp, chimeraInfo, 1, &ctxt);
			if(cneed < 0) return (int) cneed;
			dn.chiTxt.assign(ctxt, (size_t) cneed);
			if(pr && (rc = hu_batch_tsv_line_lengths(gb, pr->chiLen.data())) != HU_OK) return rc;
		}
		for(char c : dn.mainTxt) if(c == '\n') dn.placed++;
		dn.n = n;
		usFormat += usSince(tF);
		return HU_OK;
	};
	/* column-window mode: one batch of reads over the windows.  Route by the seeds (hu_route_by_seeds), one part per window through `process`, the reads
	 * that come back HU_READ_OUT_OF_WINDOW routed once more by their region (hu_route_by_region), every read's lines put back in read order */
	std::atomic<long> nRerouted{0}, nUnplaceable{0};
	auto process_windows = [&](int lane, Slot& sl, Done& dn) -> int {
		const int n = sl.f.n();
		std::vector<int32_t> lens((size_t) n), mlens((size_t) n), first((size_t) n);
		for(int i = 0; i < n; ++i) { lens[i] = (int32_t)(sl.f.offs[i + 1] - sl.f.offs[i]); if(paired) mlens[i] = (int32_t)(sl.r.offs[i + 1] - sl.r.offs[i]); }
		int rc = hu_route_by_seeds(db, colWindows, wins.data(), n, lens.data(), sl.f.vp.data(), paired ? mlens.data() : nullptr, paired ? sl.r.vp.data() : nullptr, first.data());
		if(rc != HU_OK) return rc;
		std::vector<std::string> mainL((size_t) n), chiL((size_t) n), alnL((size_t) n);
		std::vector<hu_align_rec> recs((size_t) n);
		std::vector<int32_t> win(first);
		auto run_part = [&](int k, const std::vector<int>& idx) -> int {
			Slot part; Done pd; PerRead pr;
			for(int i : idx) {
				part.f.add(sl.f.bases.substr((size_t) sl.f.offs[i], (size_t)(sl.f.offs[i + 1] - sl.f.offs[i]))); part.f.vp.insert(part.f.vp.end(), sl.f.vp.begin() + (size_t) i * 12, sl.f.vp.begin() + (size_t) i * 12 + 12);
				if(paired) { part.r.add(sl.r.bases.substr((size_t) sl.r.offs[i], (size_t)(sl.r.offs[i + 1] - sl.r.offs[i]))); part.r.vp.insert(part.r.vp.end(), sl.r.vp.begin() + (size_t) i * 12, sl.r.vp.begin() + (size_t) i * 12 + 12); }
				part.ids.push_back(sl.ids[i]); part.descs.push_back(sl.descs[i]);
			}
			const int prc = process(lane * colWindows + k, part, pd, &pr);
			if(prc != HU_OK) return prc;
			size_t am = 0, ac = 0, aa = 0;
			for(size_t j = 0; j < idx.size(); ++j) {
				const int i = idx[j];
				mainL[i].assign(pd.mainTxt, am, (size_t) pr.mainLen[j]); am += (size_t) pr.mainLen[j];
				chiL[i].assign(pd.chiTxt, ac, (size_t) pr.chiLen[j]); ac += (size_t) pr.chiLen[j];
				alnL[i].assign(pd.alnTxt, aa, (size_t) pr.alnLen[j]); aa += (size_t) pr.alnLen[j];
				recs[i] = pr.recs[j];
			}
			dn.flagged += pd.flagged;
			return HU_OK;
		};
		for(int k = 0; k < colWindows; ++k) {
			std::vector<int> idx;
			for(int i = 0; i < n; ++i) if(win[i] == k) idx.push_back(i);
			if(!idx.empty() && (rc = run_part(k, idx)) != HU_OK) return rc;
		}
		std::vector<int> out;
		for(int i = 0; i < n; ++i) if(recs[i].status == HU_READ_OUT_OF_WINDOW) out.push_back(i);
		if(!out.empty()) {
			std::vector<int32_t> cs((size_t) out.size()), ce((size_t) out.size()), w2((size_t) out.size());
			for(size_t j = 0; j < out.size(); ++j) { cs[j] = recs[out[j]].cs_start; ce[j] = recs[out[j]].cs_end; }
			if((rc = hu_route_by_region(colWindows, wins.data(), (int) out.size(), cs.data(), ce.data(), w2.data())) != HU_OK) return rc;
			for(int k = 0; k < colWindows; ++k) {
				std::vector<int> idx;
				for(size_t j = 0; j < out.size(); ++j) if(w2[j] == k && win[out[j]] != k) idx.push_back(out[j]);
				if(idx.empty()) continue;
				if((rc = run_part(k, idx)) != HU_OK) return rc;
				nRerouted += (long) idx.size();
			}
			for(int i : out) if(recs[i].status == HU_READ_OUT_OF_WINDOW) ++nUnplaceable;
		}
		for(int i = 0; i < n; ++i) { dn.mainTxt += mainL[i]; dn.chiTxt += chiL[i]; dn.alnTxt += alnL[i]; }
		for(char c : dn.mainTxt) if(c == '\n') dn.placed++;
		dn.n = n;
		return HU_OK;
	};
	std::thread seeder([&] {
		for(;;) {
			std::pair<long, std::unique_ptr<Slot>> job;
			{ std::unique_lock<std::mutex> lk(P.mu); P.cv.wait(lk, [&] { return P.err != HU_OK || !P.parsed.empty() || P.readerDone; });
			  if(P.err != HU_OK || P.parsed.empty()) { P.seederDone = true; P.cv.notify_all(); return; }
			  job = std::move(P.parsed.front()); P.parsed.pop_front(); }
			P.cv.notify_all();
			Slot& sl = *job.second; const int n = sl.f.n();
			auto tS = std::chrono::steady_clock::now();
			(void) n;
			int rc = seed_lookup(ix, sl.f, seedRegion, o.align_mode, randomHits, hitSeed, 0, (int64_t) job.first * batch);
			if(rc == HU_OK && paired) rc = seed_lookup(ix, sl.r, seedRegion, o.align_mode, randomHits, hitSeed, 1, (int64_t) job.first * batch);
			usSeed += usSince(tS);
			if(rc != HU_OK) { P.fail(rc, hu_last_error()); continue; }
			{ std::unique_lock<std::mutex> lk(P.mu); P.cv.wait(lk, [&] { return P.err != HU_OK || P.seeded.size() < depth; }); P.seeded.push_back(std::move(job)); }
			P.cv.notify_all();
		}
	});
	std::vector<std::thread> workers;
	for(int w = 0; w < nWorkers; ++w) workers.emplace_back([&, w] {
		for(;;) {
			std::pair<long, std::unique_ptr<Slot>> job;
			{ std::unique_lock<std::mutex> lk(P.mu); P.cv.wait(lk, [&] { return P.err != HU_OK || !P.seeded.empty() || P.seederDone; });
			  if(P.err != HU_OK || P.seeded.empty()) { P.workersLeft--; P.cv.notify_all(); return; }
			  job = std::move(P.seeded.front()); P.seeded.pop_front(); }
			P.cv.notify_all();
			Done dn;
			const int rc = colWindows > 1 ? process_windows(w, *job.second, dn) : process(w, *job.second, dn);
			if(rc != HU_OK) { P.fail(rc, hu_last_error()); continue; }
			{ std::unique_lock<std::mutex> lk(P.mu); P.cv.wait(lk, [&] { return P.err != HU_OK || P.done.size() < depth + (size_t) nWorkers || P.done.empty() || job.first < P.done.begin()->first; });
			  P.done.emplace(job.first, std::move(dn)); }
			P.cv.notify_all();
		}
	});
	long total = 0, placed = 0, flagged = 0;
	std::thread writer([&] {
		long nextSeq = 0;
		for(;;) {
			Done dn;
			{ std::unique_lock<std::mutex> lk(P.mu);
			  P.cv.wait(lk, [&] { return P.err != HU_OK || (!P.done.empty() && P.done.begin()->first == nextSeq) || P.workersLeft == 0; });
			  if(P.err != HU_OK) return;
			  if(P.done.empty() || P.done.begin()->first != nextSeq) return;      /* every worker has left and nothing is due */
			  dn = std::move(P.done.begin()->second); P.done.erase(P.done.begin()); }
			P.cv.notify_all();
			auto tW = std::chrono::steady_clock::now();
			out(dn.mainTxt);
			if(alnOut.on) alnOut.write(dn.alnTxt);
			if(chiOut.on) chiOut.write(dn.chiTxt);
			usWrite += usSince(tW);
			total += dn.n; placed += dn.placed; flagged += dn.flagged;
			++nextSeq;
		}
	});
	{
		long seqNo = 0;
		std::unique_ptr<Slot> cur(new Slot());
		auto submit = [&]() {
			if(cur->f.n() == 0) return;
			{ std::unique_lock<std::mutex> lk(P.mu); P.cv.wait(lk, [&] { return P.err != HU_OK || P.parsed.size() < depth; }); P.parsed.emplace_back(seqNo++, std::move(cur)); }
			P.cv.notify_all();
			cur.reset(new Slot());
		};
		Read a, b;
		const bool fq1 = is_fastq(fwdFn), fq2 = paired && is_fastq(revFn);
		auto tP = std::chrono::steady_clock::now();
		while(next_read(fin, fq1, a) && (!paired || next_read(rin, fq2, b))) {
			if(strand == 2 && !paired) cur->f.add(revcom(a.seq));          /* wrong strand for single-strand reads (src/hmmufotu.cpp:617-618) */
			else cur->f.add(a.seq);
			cur->ids.push_back(std::move(a.id)); cur->descs.push_back(std::move(a.desc));
			if(paired) cur->r.add(revcom(b.seq));              /* mates are reverse-complemented at read time (:609) */
			if(cur->f.n() == batch) { usParse += usSince(tP); submit(); tP = std::chrono::steady_clock::now(); { std::lock_guard<std::mutex> lk(P.mu); if(P.err != HU_OK) break; } }
		}
		usParse += usSince(tP);
		submit();
		{ std::lock_guard<std::mutex> lk(P.mu); P.readerDone = true; }
		P.cv.notify_all();
	}
	/* hu_last_error() is per thread; the library leaves a failing call's message in the CALLING thread's slot (a host-pool worker that throws is
	 * carried to the caller: hu_catch_all), and every stage above reads it on the thread that made the call, right after the call.
	 * The -v counters (reads in total, placed, flagged) are plain variables of the writer thread: read only below, after writer.join(). */
	seeder.join();
	for(auto& t : workers) t.join();
	writer.join();
	for(size_t w = 0; w < wgb.size(); ++w) { if(wwb[w]) hu_batch_destroy(wwb[w]); hu_batch_destroy(wgb[w]); }
	if(P.err != HU_OK) { std::cerr << "Error: " << P.msg << std::endl; return EXIT_FAILURE; }
	if(verbose) {
		const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - tLoop).count();
		std::cerr << "read loop: " << total << (paired ? " pairs" : " reads") << " in " << sec << " s = " << (sec > 0 ? total / sec : 0) << " per second (parse + seed lookup + engine + TSV, "
		          << nGpus << " device(s) x " << inflight << " batches in flight)" << std::endl;
		std::cerr << "stage busy seconds: parse " << usParse / 1e6 << " (1 thread), seed lookup " << usSeed / 1e6 << " (1 thread driving <= 16), engine " << usEngine / 1e6
		          << " + format " << usFormat / 1e6 << " (summed over " << nWorkers << " workers), write " << usWrite / 1e6 << " (1 thread)" << std::endl;
	}
	if(verbose && colWindows > 1) std::cerr << "column windows: " << nRerouted << " reads re-routed by their region, " << nUnplaceable << " whose region no window holds (widen --win-overlap)" << std::endl;
	if(verbose) std::cerr << total << " reads processed, " << placed << " assigned" << (checkChimera ? ", " + std::to_string(flagged) + " flagged as chimera" : std::string()) << std::endl;
	hu_seed_index_destroy(ix);
	for(hu_db* d : dbs) hu_db_destroy(d);
	return EXIT_SUCCESS;
}
