// hmmufotu-amd-sum: the OTU table of assignment files — the consumer side of the wire contract as
// hmmufotu-sum implements it (src/hmmufotu-sum.cpp:140-470): per input file (sample) the reads are counted by the node of their
// taxon_id column when Q_taxon >= -q and the identity filters pass; OTUs are written in node order with their taxonomy.
//   hmmufotu-amd-sum <HmmUFOtu-DB> <INFILE [INFILE2 ...]> -o OTU-OUT [-r FILE] [-l FILE] [--use-dbname] [-q DBL] [--aln-iden DBL]
//                    [--hmm-iden DBL] [-n INT] [-s INT] [-v]
//                    [-t OTU-TREE]
// -t: the OTU tree (PTUnrooted::convertToNewickTree(getAncestors(otuSeen), prefix), src/PhyloTreeUnrooted.cpp:426-447, 1127-1133; NewickTree::write,
// src/NewickTree.cpp:61-77): the tree cut down to the paths from the OTUs to the root — a node's children are written, all of them, when one
// of them lies on such a path.
// Not here: -c (consensus sequences of the OTUs: Dirichlet-density inference, training-side code of the reference) and --pseudo-tree (its new
// node ids follow the iteration order of a hash set of pointers: not reproducible); asking for them is an error, not a silent skip.
// Host only: the tree's annotations come from <DB>.ptu through hu_tree_info_* (messages read past), the profile map from <DB>.hmm.
#include <algorithm>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <sstream>
#include "hu_tsv_reader.h"
#include "../../include/hmmufotu_amd.h"

static void usage(const char* p) {
	std::cerr << "Generate phylogeny-based OTUTable with taxonomy information\n"
		"Usage:    " << p << "  <HmmUFOtu-DB> <(INFILE [INFILE2 ...]> <-o OTU-OUT> [options]\n"
		"INFILE          FILE           : assignment file(s) from hmmufotu / hmmufotu-amd, plain or .gz\n"
		"Options:    -o  FILE           : OTU summary output, required\n"
		"            -r  FILE           : output the read IDs for each OTU\n"
		"            -t  FILE           : write the OTU tree into FILE\n"
		"            -l  FILE           : sample name list, with 1st field sample-name and 2nd field assignment filename\n"
		"            --use-dbname  FLAG : use DBNAME as prefix for OTUs\n"
		"            -q  DBL            : minimum qTaxon score required [0]\n"
		"            --aln-iden  DBL    : minimum alignment identity required [0]\n"
		"            --hmm-iden  DBL    : minimum profile-HMM identity required [0]\n"
		"            -n  INT            : minimum number of observed reads required to define an OTU across all samples [0]\n"
		"            -s  INT            : minimum number of observed samples required to define an OTU [0]\n"
		"            -v  FLAG           : verbose\n"
		"            (-c and --pseudo-tree of hmmufotu-sum are not provided)\n";
}

/* a count as Eigen's IOFormat(FullPrecision) prints a double holding an integer (src/OTUTable.cpp:26, 161) */
static std::string num(double v) { std::ostringstream o; o.precision(15); o << v; return o.str(); }

int main(int argc, char** argv) {
	std::vector<std::string> pos; std::string otuFn, readFn, listFn, treeFn;
	double minQ = 0, minAln = 0, minHmm = 0; int minRead = 0, minSample = 0, verbose = 0; bool useDb = false;
	for(int i = 1; i < argc; ++i) {
		std::string a = argv[i];
		auto val = [&]() -> const char* { if(i + 1 >= argc) { std::cerr << "Error: option " << a << " needs a value\n"; exit(EXIT_FAILURE); } return argv[++i]; };
		if(a == "-h" || a == "--help") { usage(argv[0]); return EXIT_SUCCESS; }
		else if(a == "--version") { std::cerr << argv[0] << ": v1.5.1\nPackage: HmmUFOtu v1.5.1" << std::endl; return EXIT_SUCCESS; }
		else if(a == "-o") otuFn = val(); else if(a == "-r") readFn = val(); else if(a == "-l") listFn = val();
		else if(a == "--use-dbname") useDb = true;
		else if(a == "-q") minQ = atof(val()); else if(a == "--aln-iden") minAln = atof(val()); else if(a == "--hmm-iden") minHmm = atof(val());
		else if(a == "-n") minRead = atoi(val()); else if(a == "-s") minSample = atoi(val());
		else if(a == "-e" || a == "--effN") (void) val();
		else if(a == "-t") treeFn = val();
		else if(a == "-c" || a == "--pseudo-tree") { std::cerr << "Error: " << a << " (OTU consensus sequences / trees) is not provided by hmmufotu-amd-sum" << std::endl; return EXIT_FAILURE; }
		else if(a == "--no-gap") { }
		else if(a.compare(0, 2, "-v") == 0) verbose += (int) a.size() - 1;
		else if(a[0] == '-' && a.size() > 1) { std::cerr << "Error: unknown option " << a << std::endl; usage(argv[0]); return EXIT_FAILURE; }
		else pos.push_back(a);
	}
	if(pos.size() < 2) { std::cerr << "Error:" << std::endl; usage(argv[0]); return EXIT_FAILURE; }
	if(otuFn.empty()) { std::cerr << "-o must be specified" << std::endl; return EXIT_FAILURE; }
	if(!(minRead >= 0)) { std::cerr << "-n must be non-negative integer" << std::endl; return EXIT_FAILURE; }
	if(!(minSample >= 0)) { std::cerr << "-s must be non-negative integer" << std::endl; return EXIT_FAILURE; }
	const std::string dbName = pos[0];
	std::vector<std::string> inFiles(pos.begin() + 1, pos.end());
	std::map<std::string, std::string> fn2name;
	for(const std::string& f : inFiles) fn2name[f] = f;                  /* the file name is the sample name by default */
	if(!listFn.empty()) { /* src/hmmufotu-sum.cpp:213-240 */
		std::ifstream li(listFn);
		if(!li) { std::cerr << "Unable to open sample list '" << listFn << "'" << std::endl; return EXIT_FAILURE; }
		inFiles.clear();
		std::string line;
		while(std::getline(li, line)) {
			if(!line.empty() && line[0] == '#') continue;
			std::vector<std::string> f; hu_tsv::Scanner::split(line, f);
			if(f.size() >= 2 && fn2name.count(f[1])) { inFiles.push_back(f[1]); fn2name[f[1]] = f[0]; }
		}
	}
	/* database: the nodes' annotations, the profile's column map */
	hu_tree_info* ti = nullptr;
	if(hu_tree_info_load((dbName + ".ptu").c_str(), &ti) != HU_OK) { std::cerr << "Unable to load Phylogenetic tree data '" << dbName << ".ptu': " << hu_last_error() << std::endl; return EXIT_FAILURE; }
	int32_t N = 0, L = 0, K = 0, Lh = 0;
	hu_tree_info_get(ti, &N, &L, nullptr, nullptr);
	std::vector<int32_t> cs2p;
	if(minHmm != 0) {
		if(hu_files_parse((dbName + ".hmm").c_str(), nullptr, &K, &Lh, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0) != HU_OK) {
			std::cerr << "Unable to read HMM profile '" << dbName << ".hmm': " << hu_last_error() << std::endl; return EXIT_FAILURE; }
		std::vector<int32_t> p2cs((size_t) K + 1);
		hu_files_parse((dbName + ".hmm").c_str(), nullptr, &K, &Lh, nullptr, nullptr, nullptr, nullptr, nullptr, p2cs.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1);
		cs2p = hu_tsv::cs_to_profile(K, Lh, p2cs);
	}
	std::ofstream otuOut(otuFn);
	if(!otuOut) { std::cerr << "Unable to write to '" << otuFn << "'" << std::endl; return EXIT_FAILURE; }
	std::ofstream readOut;
	if(!readFn.empty()) { readOut.open(readFn); if(!readOut) { std::cerr << "Unable to write to '" << readFn << "'" << std::endl; return EXIT_FAILURE; } }
	const size_t S = inFiles.size();
	const std::string prefix = useDb ? dbName + "_" : "";
	std::map<int32_t, std::vector<long>> count;                        /* node -> reads per sample */
	std::map<int32_t, std::vector<std::string>> reads;
	std::vector<std::string> sampleNames;
	for(size_t s = 0; s < S; ++s) {
		if(verbose) std::cerr << "Processing sample " << fn2name[inFiles[s]] << " ..." << std::endl;
		hu_tsv::Scanner sc; std::string why;
		if(!sc.open(inFiles[s], why)) { std::cerr << why << std::endl; return EXIT_FAILURE; }
		sampleNames.push_back(fn2name[inFiles[s]]);
		while(sc.next()) { /* src/hmmufotu-sum.cpp:369-383 */
			const int csStart = atoi(sc.get("CS_start").c_str()), csEnd = atoi(sc.get("CS_end").c_str());
			const std::string& aln = sc.get("alignment");
			const long taxon = atol(sc.get("taxon_id").c_str());
			const double qTaxon = atof(sc.get("Q_taxon").c_str());
			if(taxon >= 0 && qTaxon >= minQ && (minAln == 0 || hu_tsv::align_identity(aln, csStart - 1, csEnd - 1) >= minAln)
					&& (minHmm == 0 || hu_tsv::hmm_identity(cs2p, aln, csStart - 1, csEnd - 1) >= minHmm)) {
				if(taxon >= N) { std::cerr << "taxon_id " << taxon << " of read '" << sc.get("id") << "' is not a node of " << dbName << std::endl; return EXIT_FAILURE; }
				std::vector<long>& c = count[(int32_t) taxon];
				if(c.empty()) c.assign(S, 0);
				c[s]++;
				if(readOut.is_open()) reads[(int32_t) taxon].push_back(sc.get("id"));
			}
		}
	}
	/* the table, OTUs in node order (src/hmmufotu-sum.cpp:405-425; OTUTable::saveTable src/OTUTable.cpp:154-164) */
	otuOut << "# HmmUFOtu v1.5.1 OTU table generated by " << argv[0] << std::endl;       /* writeProgInfo(out, " OTU table generated by " + argv[0]) */
	otuOut << "otuID";
	for(const std::string& nm : sampleNames) otuOut << "\t" << nm;
	otuOut << "\ttaxonomy" << std::endl;
	std::vector<int32_t> kept;
	for(auto& kv : count) {
		long tot = 0, ns = 0;
		for(long c : kv.second) { tot += c; ns += c > 0; }
		if(tot >= minRead && ns >= minSample) kept.push_back(kv.first);
	}
	for(int32_t u : kept) {
		const char* anno = "";
		hu_tree_info_node(ti, u, nullptr, nullptr, nullptr, nullptr, nullptr, &anno);
		otuOut << prefix << u;
		for(long c : count[u]) otuOut << "\t" << num((double) c);
		otuOut << "\t" << anno << std::endl;                             /* PTUNode::getTaxon(maxDist = inf) == the annotation */
	}
	if(readOut.is_open()) { /* src/hmmufotu-sum.cpp:433-440; the info string starts without a blank there, too */
		readOut << "# HmmUFOtu v1.5.1" << "OTU read info generated by " << argv[0] << std::endl;
		for(int32_t u : kept) { readOut << prefix << u << "\t"; const std::vector<std::string>& r = reads[u]; for(size_t i = 0; i < r.size(); ++i) readOut << (i ? " " : "") << r[i]; readOut << std::endl; }
	}
	if(!treeFn.empty()) { /* src/hmmufotu-sum.cpp:462-466 */
		std::ofstream treeOut(treeFn);
		if(!treeOut) { std::cerr << "Unable to write to '" << treeFn << "'" << std::endl; return EXIT_FAILURE; }
		int32_t root = 0;
		hu_tree_info_get(ti, nullptr, nullptr, &root, nullptr);
		std::vector<char> onPath((size_t) N, 0);                          /* getAncestors(otuSeen): the OTUs and everything above them */
		for(int32_t u : kept) for(int32_t v = u; v >= 0 && !onPath[v]; ) { onPath[v] = 1; int32_t par = -1; hu_tree_info_node(ti, v, &par, nullptr, nullptr, nullptr, nullptr, nullptr); v = par; }
		std::function<void(int32_t)> write = [&](int32_t u) {
			const int32_t* ch = nullptr; const int nc = hu_tree_info_children(ti, u, &ch);
			bool flag = false;
			for(int i = 0; i < nc; ++i) flag |= onPath[ch[i]] != 0;
			if(flag) { treeOut << '('; for(int i = 0; i < nc; ++i) { if(i) treeOut << ","; write(ch[i]); } treeOut << ')'; }
			int32_t par = -1; double len = 0;
			hu_tree_info_node(ti, u, &par, &len, nullptr, nullptr, nullptr, nullptr);
			treeOut << prefix << u << ':' << (par < 0 ? 0.0 : len);         /* NewickTree::write: the length whenever it is >= 0, at ostream's default precision */
		};
		write(root);
		treeOut << ';';
	}
	if(verbose) std::cerr << kept.size() << " OTUs over " << S << " sample(s)" << std::endl;
	hu_tree_info_free(ti);
	return EXIT_SUCCESS;
}
