// hmmufotu-amd-jplace: jplace (JSON phylogenetic placement, version 3) from assignment files — the consumer side of the wire contract
//  as hmmufotu-jplace implements it (src/hmmufotu-jplace.cpp:180-320): the tree of the database with edge
// numbers (PTUnrooted::toJPlaceTreeStr, src/PhyloTreeUnrooted.cpp:1135-1157: an edge is numbered by its child node), and one
// placement per read that passes the filters — [edge_num, likelihood, like_weight_ratio, distal_length, proximal_length,
// pendant_length] (JPlace, src/HmmUFOtu_main.cpp:241-247).
//   hmmufotu-amd-jplace <HmmUFOtu-DB> <INFILE [INFILE2 ...]> [-o FILE] [-q DBL] [--aln-iden DBL] [--hmm-iden DBL] [-sm] [-V|--var] [-a|--anno]
// The reference writes through jsoncpp's styled writer; this writes the same JSON document (same members, alphabetical member order
// as jsoncpp keeps it, numbers with 17 significant digits) with its own line breaks.  Host only.
#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>
#include "hu_tsv_reader.h"
#include "../../include/hmmufotu_amd.h"

static std::string dbl(double v) { /* boost::lexical_cast<string>(double) / jsoncpp: 17 significant digits */
	if(std::isnan(v) || std::isinf(v)) return "null";
	char t[40]; snprintf(t, sizeof(t), "%.17g", v); return t;
}
static std::string jstr(const std::string& s) {
	std::string o = "\"";
	for(char c : s) { if(c == '"' || c == '\\') { o += '\\'; o += c; } else if(c == '\n') o += "\\n"; else if(c == '\t') o += "\\t"; else if((unsigned char) c < 0x20) { char t[8]; snprintf(t, sizeof(t), "\\u%04x", c); o += t; } else o += c; }
	return o + "\"";
}
static void tree_str(const hu_tree_info* ti, int32_t node, std::string& out) {
	const int32_t* ch = nullptr;
	const int nc = hu_tree_info_children(ti, node, &ch);
	if(nc > 0) { out += "("; for(int i = 0; i < nc; ++i) { if(i) out += ","; tree_str(ti, ch[i], out); } out += ")"; }
	int32_t parent = -1; double len = 0;
	hu_tree_info_node(ti, node, &parent, &len, nullptr, nullptr, nullptr, nullptr);
	out += std::to_string(node);
	if(parent >= 0 && len > 0) out += ":" + dbl(len);
	if(parent >= 0) out += "{" + std::to_string(node) + "}";          /* getEdgeID(node, parent) = the child's id */
}

int main(int argc, char** argv) {
	std::vector<std::string> pos; std::string outFn; double minQ = 0, minAln = 0, minHmm = 0; bool showSm = false, showVar = false, showAnno = false;
	std::string cmd;
	for(int i = 0; i < argc; ++i) { cmd += argv[i]; cmd += i + 1 < argc ? " " : ""; }
	for(int i = 1; i < argc; ++i) {
		std::string a = argv[i];
		auto val = [&]() -> const char* { if(i + 1 >= argc) { std::cerr << "Error: option " << a << " needs a value\n"; exit(EXIT_FAILURE); } return argv[++i]; };
		if(a == "-h" || a == "--help") { std::cerr << "Usage:    " << argv[0] << "  <HmmUFOtu-DB> <(INFILE [INFILE2 ...]> [-o FILE] [-q DBL] [--aln-iden DBL] [--hmm-iden DBL] [-sm] [-V|--var] [-a|--anno]" << std::endl; return EXIT_SUCCESS; }
		else if(a == "--version") { std::cerr << argv[0] << ": v1.5.1\nPackage: HmmUFOtu v1.5.1" << std::endl; return EXIT_SUCCESS; }
		else if(a == "-o") outFn = val(); else if(a == "-q") minQ = atof(val());
		else if(a == "--aln-iden") minAln = atof(val()); else if(a == "--hmm-iden") minHmm = atof(val());
		else if(a == "-sm") showSm = true; else if(a == "-V" || a == "--var") showVar = true; else if(a == "-a" || a == "--anno") showAnno = true;
		else if(a.compare(0, 2, "-v") == 0) { }
		else if(a[0] == '-' && a.size() > 1) { std::cerr << "Error: unknown option " << a << std::endl; return EXIT_FAILURE; }
		else pos.push_back(a);
	}
	if(pos.size() < 2) { std::cerr << "Error: <HmmUFOtu-DB> and at least one assignment file are needed" << std::endl; return EXIT_FAILURE; }
	(void) minAln; (void) minHmm;    /* parsed like the reference parses them; its filter tests the identities for non-zero only (below) */
	const std::string dbName = pos[0];
	hu_tree_info* ti = nullptr;
	if(hu_tree_info_load((dbName + ".ptu").c_str(), &ti) != HU_OK) { std::cerr << "Unable to load Phylogenetic tree data '" << dbName << ".ptu': " << hu_last_error() << std::endl; return EXIT_FAILURE; }
	int32_t N = 0, L = 0, root = 0, K = 0, Lh = 0; hu_model_desc md;
	hu_tree_info_get(ti, &N, &L, &root, &md);
	if(hu_files_parse((dbName + ".hmm").c_str(), nullptr, &K, &Lh, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0) != HU_OK) {
		std::cerr << "Unable to read HMM profile '" << dbName << ".hmm': " << hu_last_error() << std::endl; return EXIT_FAILURE; }
	std::vector<int32_t> p2cs((size_t) K + 1);
	hu_files_parse((dbName + ".hmm").c_str(), nullptr, &K, &Lh, nullptr, nullptr, nullptr, nullptr, nullptr, p2cs.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1);
	const std::vector<int32_t> cs2p = hu_tsv::cs_to_profile(K, Lh, p2cs);
	std::ofstream of; if(!outFn.empty()) { of.open(outFn); if(!of) { std::cerr << "Unable to write to '" << outFn << "'" << std::endl; return EXIT_FAILURE; } }
	std::ostream& out = of.is_open() ? of : std::cout;
	std::string placements;
	long nPlaced = 0;
	for(size_t f = 1; f < pos.size(); ++f) {
		hu_tsv::Scanner sc; std::string why;
		if(!sc.open(pos[f], why)) { std::cerr << why << std::endl; return EXIT_FAILURE; }
		while(sc.next()) { /* src/hmmufotu-jplace.cpp:229-271 */
			const int csStart = atoi(sc.get("CS_start").c_str()), csEnd = atoi(sc.get("CS_end").c_str());
			const std::string& aln = sc.get("alignment");
			const double ratio = atof(sc.get("branch_ratio").c_str());
			const long taxon = atol(sc.get("taxon_id").c_str());
			const double annoDist = atof(sc.get("anno_dist").c_str()), loglik = atof(sc.get("loglik").c_str()), q = atof(sc.get("Q_placement").c_str());
			/* the reference tests the two identities for being non-zero, not against --aln-iden / --hmm-iden (src/hmmufotu-jplace.cpp:241-243) */
			if(!(taxon >= 0 && q >= minQ && hu_tsv::align_identity(aln, csStart - 1, csEnd - 1) && hu_tsv::hmm_identity(cs2p, aln, csStart - 1, csEnd - 1))) continue;
			int c = 0, p = 0;
			sscanf(sc.get("branch_id").c_str(), "%d->%d", &c, &p);
			if(c < 0 || c >= N || p < 0 || p >= N) { std::cerr << "branch_id '" << sc.get("branch_id") << "' of read '" << sc.get("id") << "' names no node of " << dbName << std::endl; return EXIT_FAILURE; }
			int32_t cp = -1, pp = -1; double lenC = 0, lenP = 0;
			hu_tree_info_node(ti, c, &cp, &lenC, nullptr, nullptr, nullptr, nullptr); hu_tree_info_node(ti, p, &pp, &lenP, nullptr, nullptr, nullptr, nullptr);
			const long edge = cp == p ? c : pp == c ? p : -1;              /* getEdgeID */
			const double edgeLen = cp == p ? lenC : pp == c ? lenP : NAN;  /* getBranchLength(cNode, pNode) */
			/* JPlace (src/HmmUFOtu_main.cpp:241-247) */
			const double distal = edgeLen * ratio, proximal = edgeLen * (1.0 - ratio);
			const double pendant = ratio <= 0.5 ? annoDist - distal : annoDist - proximal;
			const double likeRatio = q >= 250 ? 1 : std::exp(-q / 10 * std::log(10.0));      /* q2p, src/math/Stats.h:244-246 */
			placements += (nPlaced++ ? ",\n\t\t" : "\n\t\t");
			placements += "{ \"n\" : [ " + jstr(sc.get("id")) + " ], \"p\" : [ [ " + std::to_string(edge) + ", " + dbl(loglik) + ", " + dbl(likeRatio) + ", " + dbl(distal) + ", " +
				dbl(proximal) + ", " + dbl(pendant) + " ] ] }";
		}
	}
	std::string tree; tree_str(ti, root, tree); tree += ";";
	static const char* mnames[] = {"GTR", "TN93", "HKY85", "F81", "K80", "JC69"};
	out << "{\n\t\"fields\" : [ \"edge_num\", \"likelihood\", \"like_weight_ratio\", \"distal_length\", \"proximal_length\", \"pendant_length\" ],\n";
	out << "\t\"metadata\" : {\n";
	if(showVar) out << "\t\t\"among_site_rate_variation\" : " << jstr(md.dg_k > 0 ? "Discrete Gamma model" : "none") << ",\n";
	out << "\t\t\"invocation\" : " << jstr(cmd);
	if(showAnno) {
		out << ",\n\t\t\"node_taxonomy_annotations\" : {";
		for(int32_t i = 0; i < N; ++i) { const char* an = ""; hu_tree_info_node(ti, i, nullptr, nullptr, nullptr, nullptr, nullptr, &an); out << (i ? ", " : " ") << jstr(std::to_string(i)) << " : " << jstr(an); }
		out << " }";
	}
	if(showSm) out << ",\n\t\t\"substitution_model\" : " << jstr(mnames[md.type]);
	out << "\n\t},\n";
	out << "\t\"placements\" : [" << placements << (nPlaced ? "\n\t" : "") << "],\n";
	out << "\t\"tree\" : " << jstr(tree) << ",\n\t\"version\" : 3\n}" << std::endl;
	hu_tree_info_free(ti);
	return EXIT_SUCCESS;
}
