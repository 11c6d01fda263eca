// Test driver of the reference-shaped per-read adapters (hu_reference_api.hpp) and of the in-memory database route of
// INTEGRATION.md §2B.  Built by the Makefile into ../bin/hu_adapter_test, run by tests/test_adapters_gpu.py, which compares
// every printed record with the CPU oracle.
//
//   hu_adapter_test <db.hmm> <db.ptu> <reads.txt> [files|text]
//
// reads.txt: one read per line: <bases> then 12 integers (two ViterbiAlignPath rows start end from to nIns nDel; an unused
// row is all zero).  "files": hu_db_load on the two files.  "text": the profile and the model go through
// hu_profile_parse_text / hu_model_parse_text — the text a maintainer gets from `os << hmm` and `model->write(os)` on loaded
// reference objects — and the tree arrays through hu_db_create, as they would come from PTUnrooted's public getters.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include "hu_reference_api.hpp"

using namespace hmmufotu_amd;

static std::string slurp(const char* path) { std::ifstream f(path, std::ios::binary); std::ostringstream o; o << f.rdbuf(); return o.str(); }

int main(int argc, char** argv) {
	if(argc < 4) { fprintf(stderr, "usage: %s <db.hmm> <db.ptu> <reads.txt> [files|text]\n", argv[0]); return 2; }
	const bool viaText = argc > 4 && strcmp(argv[4], "text") == 0;
	hu_db* db = nullptr;
	try {
		if(!viaText) check(hu_db_load(argv[1], argv[2], 0, &db));
		else {
			const std::string hmmText = slurp(argv[1]);                       // stands for: std::ostringstream os; os << hmm;
			int32_t K = 0, L = 0, n = 0, root = 0;
			check(hu_profile_parse_text(hmmText.data(), (int64_t) hmmText.size(), &K, &L, nullptr, nullptr, nullptr, nullptr));
			std::vector<double> EM(4 * (K + 1)), EI(4 * (K + 1)), T(7 * (K + 1)); std::vector<int32_t> p2cs(K + 1);
			check(hu_profile_parse_text(hmmText.data(), (int64_t) hmmText.size(), &K, &L, EM.data(), EI.data(), T.data(), p2cs.data()));
			hu_model_desc md;
			check(hu_files_parse(nullptr, argv[2], nullptr, &L, &n, &root, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
					nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &md, 0));
			std::vector<int32_t> parent(n); std::vector<double> blen(n), height(n), up((size_t) n * L * 4), down((size_t) n * L * 4);
			std::vector<int8_t> seq((size_t) n * L);
			check(hu_files_parse(nullptr, argv[2], nullptr, &L, &n, &root, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
					parent.data(), blen.data(), seq.data(), height.data(), up.data(), down.data(), &md, 1));
			/* the model once more through its text form (what DNASubModel::write emits inside the .ptu): GTR needs pi + R */
			std::ostringstream mt;
			static const char* names[] = {"GTR", "TN93", "HKY85", "F81", "K80", "JC69"};
			mt.precision(17);
			mt << names[md.type] << "\n# DNA Substitution Model\nType: " << names[md.type] << "\n";
			if(md.type != HU_K80 && md.type != HU_JC69) mt << "pi: " << md.pi[0] << " " << md.pi[1] << " " << md.pi[2] << " " << md.pi[3] << "\n";
			if(md.type == HU_GTR) {
				mt << "R:\n";
				for(int i = 0; i < 4; ++i) mt << md.par[4 * i] << " " << md.par[4 * i + 1] << " " << md.par[4 * i + 2] << " " << md.par[4 * i + 3] << "\n";
				mt << "Q:\n0 0 0 0\n0 0 0 0\n0 0 0 0\n0 0 0 0\n";
			}
			else if(md.type == HU_TN93) mt << "kr: " << md.par[0] << "\nky: " << md.par[1] << "\nbeta: " << md.par[2] << "\n";
			else if(md.type == HU_HKY85) mt << "kappa: " << md.par[0] << "\nbeta: " << md.par[1] << "\n";
			else if(md.type == HU_F81) mt << "beta: " << md.par[0] << "\n";
			else if(md.type == HU_K80) mt << "kappa: " << md.par[0] << "\n";
			hu_model_desc md2;
			const std::string ms = mt.str();
			check(hu_model_parse_text(ms.data(), (int64_t) ms.size(), &md2));
			md2.dg_k = md.dg_k; memcpy(md2.dg_rate, md.dg_rate, sizeof(md.dg_rate));
			hu_profile_desc pd{K, L, EM.data(), EI.data(), T.data(), p2cs.data()};
			hu_tree_desc td; memset(&td, 0, sizeof(td));
			td.n_nodes = n; td.cs_len = L; td.parent = parent.data(); td.blen = blen.data(); td.seq = seq.data();
			td.up = up.data(); td.down = down.data(); td.height = height.data();
			check(hu_db_create(&pd, &td, &md2, 0, &db));
		}
		hu_opts o; hu_default_opts(&o);
		Engine ptu(db, o);
		const size_t maxNSeed = (size_t) o.max_nseed;
		std::ifstream in(argv[3]);
		std::string line;
		int r = 0;
		while(std::getline(in, line)) {
			if(line.empty()) continue;
			std::istringstream ls(line);
			std::string bases; ls >> bases;
			std::vector<ViterbiAlignPath> vpaths;
			for(int k = 0; k < 2; ++k) { ViterbiAlignPath v; ls >> v.start >> v.end >> v.from >> v.to >> v.nIns >> v.nDel; if(v.start > 0) vpaths.push_back(v); }
			/* the task body of src/hmmufotu.cpp:621-733, call by call */
			HmmAlignment aln = ptu.alignSeq(bases, vpaths);
			printf("ALN %d %d %d %d %d %d %d %d %.17g %s\n", r, aln.status, aln.seqStart, aln.seqEnd, aln.hmmStart, aln.hmmEnd, aln.csStart, aln.csEnd, aln.cost, aln.isValid() ? aln.align.c_str() : "-");
			if(aln.isValid()) {
				DigitalSeq seq(aln.align);                                                                        /* :641 */
				std::vector<PTLoc> seeds = getSeed(ptu, seq, aln.csStart - 1, aln.csEnd - 1, o.max_diff, o.max_height);   /* :645 */
				if(r < 3) { /* the reference's WHOLE vector on request: every eligible node, sorted by the caller's own std::sort — its head is the device's list */
					const std::vector<PTLoc> whole = getSeed(ptu, seq, aln.csStart - 1, aln.csEnd - 1, o.max_diff, o.max_height, true);
					bool head = whole.size() >= seeds.size(), sorted = true;
					for(size_t i = 0; head && i < seeds.size(); ++i) head = whole[i].id == seeds[i].id && whole[i].dist == seeds[i].dist;
					for(size_t i = 1; i < whole.size(); ++i) sorted = sorted && !(whole[i].dist < whole[i - 1].dist);
					printf("WHOLE %d %zu %d %d\n", r, whole.size(), (int) head, (int) sorted);
				}
				if(seeds.size() > maxNSeed) seeds.erase(seeds.end() - (seeds.size() - maxNSeed), seeds.end());      /* :646-647 */
				printf("SEED %d %zu", r, seeds.size());
				for(const PTLoc& l : seeds) printf(" %ld:%.17g", l.id, l.dist);
				printf("\n");
				std::vector<PTPlacement> places = estimateSeq(ptu, seq, seeds, "unweighted");                       /* :720 */
				printf("EST %d %zu", r, places.size());
				for(const PTPlacement& p : places) printf(" %ld:%.17g:%.17g:%.17g", p.cNode, p.ratio, p.wnr, p.loglik);
				printf("\n");
				{ /* the member form on one location gives the same record */
					const PTPlacement one = ptu.estimateSeq(seq, seeds[0], "unweighted");
					printf("MEMBER %d %d\n", r, (int)(one.cNode == places[0].cNode && one.ratio == places[0].ratio && one.wnr == places[0].wnr && one.loglik == places[0].loglik));
				}
				filterPlacements(places, o.max_error);                                                           /* :722 */
				printf("FILT %d %zu", r, places.size());
				for(const PTPlacement& p : places) printf(" %ld", p.cNode);
				printf("\n");
				placeSeq(ptu, seq, places, o.max_height);                                                         /* :724 */
				printf("PLACED %d %zu", r, places.size());
				for(const PTPlacement& p : places) printf(" %ld:%ld:%.17g:%.17g:%.17g:%.17g", p.cNode, p.aNode, p.ratio, p.wnr, p.loglik, p.height);
				printf("\n");
				calcQValues(ptu, places, UNIFORM);                                                                /* :729 */
				printf("Q %d %zu", r, places.size());
				for(const PTPlacement& p : places) printf(" %ld:%.17g:%.17g", p.cNode, p.qPlace, p.qTaxon);
				printf("\n");
				std::sort(places.rbegin(), places.rend(), compareByQPlace);                                       /* :730 */
				const PTPlacement& p = places[0];                                                                 /* :733 */
				printf("PLACE %d %ld %ld %ld %d %d %.17g %.17g %.17g %.17g %.17g %.17g\n", r, p.cNode, p.pNode, p.aNode, p.start, p.end, p.ratio, p.wnr, p.loglik, p.height, p.qPlace, p.qTaxon);
				if(r == 0 && places.size() > 1) { /* a caller that drops a placement between the stages: the later stages see the vector as it is */
					std::vector<PTPlacement> fewer(places.begin() + 1, places.end());
					calcQValues(ptu, fewer, UNIFORM);
					printf("QDROP %d %zu", r, fewer.size());
					for(const PTPlacement& q : fewer) printf(" %ld:%.17g", q.cNode, q.qPlace);
					printf("\n");
				}
			}
			++r;
		}
	}
	catch(const std::exception& e) { fprintf(stderr, "hu_adapter_test: %s\n", e.what()); if(db) hu_db_destroy(db); return 1; }
	hu_db_destroy(db);
	return 0;
}
