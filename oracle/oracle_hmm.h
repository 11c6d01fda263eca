// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_models.h header).  PARITY UNPINNED.
//
// Banded plan-7 profile HMM in cost (-ln p) space: profile preparation, banded/full
// Viterbi, traceback, CS-coordinate alignment.  Follows
//   src/BandedHMMP7.cpp:100-109 (post-load chain), :561-583 (setSequenceMode),
//   :701-705 (extend_index), :721-746, :748-892 (DP), :894-941 (buildAlignPath),
//   :943-1006 (trace), :1008-1081 (global align), :1083-1120 (wingRetract),
//   :1137-1186 (getPaddingSeq), :1188-1213 (merge); src/BandedHMMP7.h:668-788;
//   src/BandedHMMP7Bg.cpp:33-35; src/HmmUFOtu_main.cpp:39-105 (alignSeq, minus the
//   CSFM lookup which stays outside the boundary: VPaths are inputs).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <limits>
#include <algorithm>

namespace orc {

static const double INF = std::numeric_limits<double>::infinity();

/* IUPACNucl alphabet (src/IUPACNucl.cpp:33-50, src/DegenAlphabet.cpp:43-64):
 * ACGT -> 0..3, degenerate -> first expansion, "-._" -> -2, everything else -1 */
struct Alphabet {
	int8_t map[128];
	Alphabet() {
		for(int i = 0; i < 128; ++i) map[i] = -1;
		map[(int)'A'] = 0; map[(int)'C'] = 1; map[(int)'G'] = 2; map[(int)'T'] = 3;
		map[(int)'U'] = 3; map[(int)'M'] = 0; map[(int)'R'] = 0; map[(int)'W'] = 0;
		map[(int)'S'] = 1; map[(int)'Y'] = 1; map[(int)'K'] = 2; map[(int)'V'] = 0;
		map[(int)'H'] = 0; map[(int)'D'] = 0; map[(int)'B'] = 1; map[(int)'N'] = 0;
		map[(int)'-'] = -2; map[(int)'.'] = -2; map[(int)'_'] = -2;
	}
	int8_t encode(char c) const { return (c >= 0) ? map[(int)c] : -1; }
	bool isSymbol(char c) const { return encode(c) >= 0; }
};
static const Alphabet ABC;

enum AlignMode { GLOBAL = 0, LOCAL = 1, NGCL = 2, CGNL = 3 };
enum { tMM = 0, tMI = 1, tMD = 2, tIM = 3, tII = 4, tDM = 5, tDD = 6 };

struct VPath { int start, end, from, to, nIns, nDel;
	bool isValid() const { return start > 0 && start <= end && from > 0 && from <= to && nIns >= 0 && nDel >= 0; } };

struct HmmAlignment {
	int K = 0, L = 0;
	int seqStart = 0, seqEnd = 0, hmmStart = 0, hmmEnd = 0, csStart = 0, csEnd = 0;
	double cost = 0;
	std::string align;
	std::string trace;
	bool usedFull = false;
	bool isValid() const {
		return 0 < seqStart && seqStart <= seqEnd && 0 < hmmStart && hmmStart <= hmmEnd && hmmEnd <= K &&
				0 < csStart && csStart <= csEnd && csEnd <= L && cost >= 0 && cost != INF && L == (int) align.length();
	}
	void merge(const HmmAlignment& o) { /* src/BandedHMMP7.cpp:1188-1213 */
		if(!(K == o.K && L == o.L)) return;
		if(o.seqStart < seqStart) seqStart = o.seqStart;
		if(o.seqEnd > seqEnd) seqEnd = o.seqEnd;
		if(o.hmmStart < hmmStart) hmmStart = o.hmmStart;
		if(o.hmmEnd > hmmEnd) hmmEnd = o.hmmEnd;
		if(o.csStart < csStart) csStart = o.csStart;
		if(o.csEnd > csEnd) csEnd = o.csEnd;
		cost += o.cost;
		for(int i = 0; i < L; ++i)
			if(align[i] == '.' && o.align[i] != '.') align[i] = o.align[i];
	}
};

struct Hmm {
	int K = 0, L = 0;
	std::vector<double> EM, EI;   // cost, [k*4 + b], k = 0..K
	std::vector<double> T;        // cost, [k*7 + t]
	std::vector<double> entry, exit_; // probabilities, 0..K
	std::vector<double> entryC, exitC; // costs
	std::vector<int> cs2p, p2cs;  // 1-based maps, index 0 dummy
	double T_NN = INF, T_NB = INF, T_EC = INF, T_CC = INF; // costs

	/* post-load chain of operator>> (src/BandedHMMP7.cpp:104-109) */
	void init(int K_, int L_, const double* em, const double* ei, const double* t, const int* p2cs_) {
		K = K_; L = L_;
		EM.assign(em, em + 4 * (K + 1));
		EI.assign(ei, ei + 4 * (K + 1));
		T.assign(t, t + 7 * (K + 1));
		p2cs.assign(p2cs_, p2cs_ + K + 1);
		int maxIdx = std::max(L, p2cs[K]) + 2;
		cs2p.assign(maxIdx, 0);
		for(int k = 1; k <= K; ++k) cs2p[p2cs[k]] = k;
		for(int i = p2cs[K] + 1; i <= L && i < 65536; ++i) cs2p[i] = K; /* extend_index */
		/* adjustProfileLocalMode: entry/exit from Tmat[0](M,M), Tmat[K](M,M) with Tmat = exp(-cost) */
		entry.assign(K + 1, 0.0); exit_.assign(K + 1, 0.0);
		double t0 = std::exp(-T[0 * 7 + tMM]), tK = std::exp(-T[K * 7 + tMM]);
		for(int k = 1; k <= K; ++k) { entry[k] = t0; exit_[k] = tK; }
		/* wingRetract (src/BandedHMMP7.cpp:1083-1120) */
		for(int j = 2; j <= K; ++j) {
			double cost = 0;
			cost += T[0 * 7 + tMD];
			for(int i = 1; i < j - 1; ++i) cost += T[i * 7 + tDD];
			cost += T[(j - 1) * 7 + tDM];
			entry[j] += std::exp(-cost);
			if(entry[j] > 1) entry[j] = 1;
		}
		for(int i = 1; i <= K - 1; ++i) {
			double cost = 0;
			cost += T[i * 7 + tMD];
			for(int j = i + 1; j < K; ++j) cost += T[j * 7 + tDD];
			cost += T[K * 7 + tDM];
			exit_[i] += std::exp(-cost);
			if(exit_[i] > 1) exit_[i] = 1;
		}
		entryC.resize(K + 1); exitC.resize(K + 1);
		for(int k = 0; k <= K; ++k) { entryC[k] = -std::log(entry[k]); exitC[k] = -std::log(exit_[k]); }
	}

	/* setSequenceMode (src/BandedHMMP7.cpp:561-583); p1 per src/BandedHMMP7Bg.cpp:33-35 */
	void setMode(int mode) {
		const int MIN_BG_K = 350;
		double p1 = K >= MIN_BG_K ? K / (K + 1.0) : MIN_BG_K / (MIN_BG_K + 1.0);
		double term = 1 - p1;
		double nn = 0, cc = 0;
		switch(mode) {
		case GLOBAL: nn = cc = 0; break;
		case LOCAL: nn = cc = term; break;
		case NGCL: nn = 0; cc = term; break;
		case CGNL: nn = term; cc = 0; break;
		}
		double nb = 1.0 - nn, ec = 1.0;
		T_NN = -std::log(nn); T_NB = -std::log(nb); T_EC = -std::log(ec); T_CC = -std::log(cc);
	}

	int profileLoc(int idx) const { return idx >= 0 && idx < (int) cs2p.size() ? cs2p[idx] : 0; }

	/* buildAlignPath (src/BandedHMMP7.cpp:894-941) */
	VPath buildAlignPath(int locStart, int locEnd, const std::string& CS, int csFrom, int csTo) const {
		(void) locEnd; (void) csTo;
		int start = 0, end = 0, from = 0, to = 0, nIns = 0, nDel = 0;
		int i = csFrom, j = locStart;
		for(char c : CS) {
			int k = profileLoc(j);
			bool nonGap = ABC.isSymbol(c);
			if(from == 0 && nonGap) from = i;
			if(nonGap) to = i;
			if(k != 0) {
				if(start == 0) start = k;
				end = k;
				if(!nonGap) nDel++;
			}
			else if(nonGap) nIns++;
			j++;
			if(nonGap) i++;
		}
		return VPath{start, end, from, to, nIns, nDel};
	}
};

/* dense DP workspace with lazy reset of touched cells */
struct VitWork {
	int K = 0, cap = 0;
	std::vector<double> M, I, D;      // (cap+1) x (K+1), index i*(K+1)+j
	std::vector<uint8_t> mark;
	std::vector<int> touched;
	void ensure(int K_, int L) {
		if(K_ != K || L > cap) {
			K = K_; cap = std::max(L, cap);
			size_t n = (size_t)(cap + 1) * (K + 1);
			M.assign(n, INF); I.assign(n, INF); D.assign(n, INF); mark.assign(n, 0);
			touched.clear();
		}
	}
	void touch(size_t c) { if(!mark[c]) { mark[c] = 1; touched.push_back((int) c); } }
	void reset() {
		for(int c : touched) { M[c] = I[c] = D[c] = INF; mark[c] = 0; }
		touched.clear();
	}
};

inline double min3(double a, double b, double c) { return std::min(a, std::min(b, c)); }
inline double min4(double a, double b, double c, double d) { return std::min(a, min3(b, c, d)); }

struct Viterbi {
	const Hmm& h;
	VitWork& w;
	const int8_t* x; // encoded read, 0-based
	int L;           // read length
	int W;           // K+1
	Viterbi(const Hmm& h, VitWork& w, const int8_t* x, int L) : h(h), w(w), x(x), L(L), W(h.K + 1) { w.ensure(h.K, L); }

	size_t at(int i, int j) const { return (size_t) i * W + j; }

	void prepare() { /* prepareViterbiScores (src/BandedHMMP7.cpp:735-746) */
		for(int i = 1; i <= L; ++i) {
			size_t c = at(i, 0);
			w.touch(c);
			w.M[c] = (i == 1 ? 0 : h.T_NN * (i - 1));
			w.M[c] += h.T_NB;
			w.I[c] = w.M[c];
		}
	}
	inline void cell(int i, int j, bool withB) {
		const int K = h.K;
		size_t c = at(i, j), cd = at(i - 1, j - 1), cu = at(i - 1, j), cl = at(i, j - 1);
		w.touch(c);
		int b = x[i - 1];
		const double* tp = &h.T[(size_t)(j - 1) * 7];
		const double* tj = &h.T[(size_t) j * 7];
		double vm = w.M[cd] + tp[tMM], vi = w.I[cd] + tp[tIM], vd = w.D[cd] + tp[tDM];
		double best = withB ? min4(w.M[at(i, 0)] + h.entryC[j], vm, vi, vd) : min3(vm, vi, vd);
		w.M[c] = h.EM[(size_t) j * 4 + b] + best;
		w.I[c] = h.EI[(size_t) j * 4 + b] + std::min(w.M[cu] + tj[tMI], w.I[cu] + tj[tII]);
		if(j > 1 && j < K)
			w.D[c] = std::min(w.M[cl] + tp[tMD], w.D[cl] + tp[tDD]);
	}
	void full() { /* src/BandedHMMP7.cpp:748-771 */
		prepare();
		for(int j = 1; j <= h.K; ++j)
			for(int i = 1; i <= L; ++i)
				cell(i, j, true);
	}
	void banded(const std::vector<VPath>& vp) { /* src/BandedHMMP7.cpp:782-881 */
		prepare();
		const int K = h.K;
		for(size_t p = 0; p < vp.size(); ++p) {
			const VPath& v = vp[p];
			int upQLen = p == 0 ? v.from - 1 : v.from - vp[p - 1].to;
			if(upQLen < 0) upQLen = 0;
			int up_start = p == 0 ? (int)(v.start - upQLen * (1 + 0.2)) : vp[p - 1].end;
			if(up_start < 1) up_start = 1;
			int up_from = p == 0 ? (int)(v.from - upQLen * (1 + 0.2)) : vp[p - 1].to;
			if(up_from < 1) up_from = 1;
			for(int j = up_start; j <= v.start; ++j)
				for(int i = up_from; i <= v.from; ++i)
					cell(i, j, true);
			for(int j = v.start; j <= v.end; ++j)
				for(int i = v.from; i <= v.to; ++i) {
					int dist = (i - v.from) - (j - v.start);
					if(!(dist <= v.nIns && dist >= -v.nDel)) continue;
					cell(i, j, true);
				}
		}
		const VPath& last = vp.back();
		int downQLen = L - last.to;
		int down_end = (int)(last.end + downQLen * (1 + 0.2));
		int down_to = (int)(last.to + downQLen * (1 + 0.2));
		if(down_end > K) down_end = K;
		if(down_to > L) down_to = L;
		for(int j = last.end; j <= down_end; ++j)
			for(int i = last.to; i <= down_to; ++i)
				cell(i, j, false);
	}
	/* S value of cell (i, col), col in 0..K+1 (src/BandedHMMP7.cpp:883-891) */
	double S(int i, int col) const {
		double s;
		if(col <= h.K) { s = w.M[at(i, col)]; s += h.exitC[col]; }
		else { s = w.I[at(i, h.K)]; s += h.T[(size_t) h.K * 7 + tIM]; }
		s += h.T_EC;
		if(i >= 1 && i < L) s += h.T_CC * (L - i);
		return s;
	}
	static char whichMin4(double pB, double pM, double pI, double pD) {
		int idx = 0; double mn = INF;
		if(pB < mn) { idx = 0; mn = pB; }
		if(pM < mn) { idx = 1; mn = pM; }
		if(pI < mn) { idx = 2; mn = pI; }
		if(pD < mn) { idx = 3; mn = pD; }
		return "BMID"[idx];
	}
	static char whichMin2(double p0, double p1, const char* st) {
		int idx = 0; double mn = INF;
		if(p0 < mn) { idx = 0; mn = p0; }
		if(p1 < mn) { idx = 1; mn = p1; }
		return st[idx];
	}
	/* buildViterbiTrace (src/BandedHMMP7.cpp:943-1006); returns false if all-inf */
	bool trace(double& minScore, int& alnStart, int& alnEnd, int& alnFrom, int& alnTo, std::string& tr) const {
		const int K = h.K;
		/* minCoeff over the (L+1)x(K+2) matrix, first strict minimum in column-major order.
		 * Only touched cells (and column K+1 via I(.,K)) can be finite. */
		minScore = INF; int minRow = 0, minCol = 0;
		bool any = false;
		for(int c : w.touched) {
			int i = c / W, j = c % W;
			double s = S(i, j);
			if(!(s < INF)) continue; /* inf and NaN never win Eigen's minCoeff visitor */
			if(!any || s < minScore || (s == minScore && (j < minCol || (j == minCol && i < minRow)))) {
				any = true; minScore = s; minRow = i; minCol = j;
			}
		}
		for(int i = 0; i <= L; ++i) {
			if(!w.mark[at(i, K)]) continue;
			double s = S(i, K + 1);
			if(s < minScore) { minScore = s; minRow = i; minCol = K + 1; }
		}
		if(minScore == INF) return false;
		char s = minCol <= K ? 'M' : 'I';
		int i = minRow, j = minCol <= K ? minCol : K;
		alnEnd = j; alnTo = minRow;
		tr.clear();
		tr.push_back('E');
		while(i >= 1 && j >= 0) {
			tr.push_back(s);
			if(s == 'M') {
				const double* tp = &h.T[(size_t)(j - 1) * 7];
				size_t cd = at(i - 1, j - 1);
				s = j > 1 ? whichMin4(w.M[at(i, 0)] + h.entryC[j], w.M[cd] + tp[tMM], w.I[cd] + tp[tIM], w.D[cd] + tp[tDM])
						: whichMin2(w.M[at(i, 0)] + h.entryC[j], w.I[cd] + tp[tIM], "BI");
				i--; j--;
			}
			else if(s == 'I') {
				const double* tj = &h.T[(size_t) j * 7];
				size_t cu = at(i - 1, j);
				s = j > 0 ? whichMin2(w.M[cu] + tj[tMI], w.I[cu] + tj[tII], "MI")
						: whichMin2(w.M[at(i, 0)] + h.T[0 * 7 + tMI], w.I[cu] + tj[tII], "BI");
				i--;
			}
			else if(s == 'D') {
				const double* tp = &h.T[(size_t)(j - 1) * 7];
				size_t cl = at(i, j - 1);
				s = whichMin2(w.M[cl] + tp[tMD], w.D[cl] + tp[tDD], "MD");
				j--;
			}
			else break;
		}
		alnStart = j + 1; alnFrom = i + 1;
		if(tr.back() != 'B') tr.push_back('B');
		std::reverse(tr.begin(), tr.end());
		return true;
	}
};

/* getPaddingSeq (src/BandedHMMP7.cpp:1137-1186), modes used by buildGlobalAlign */
enum PadMode { PAD_LEFT, PAD_RIGHT, PAD_JUSTIFIED };
inline std::string paddingSeq(int L, const std::string& insert, char padCh, PadMode mode) {
	if(insert.empty()) return std::string(L > 0 ? L : 0, padCh);
	std::string pad;
	int n = (int) insert.length();
	switch(mode) {
	case PAD_LEFT:
		if(n >= L) pad.append(insert.substr(0, L));
		else { pad.append(insert); pad.append(L - n, padCh); }
		break;
	case PAD_RIGHT:
		if(n >= L) pad.append(insert.substr(n - L, L));
		else { pad.append(L - n, padCh); pad.append(insert); }
		break;
	case PAD_JUSTIFIED:
		if(n >= L) {
			pad.append(insert.substr(0, (int) std::floor(L / 2.0)));
			pad.append(insert.substr(n - (int) std::ceil(L / 2.0), (int) std::ceil(L / 2.0)));
		}
		else { /* reference quirk (:1173-1177): the second half repeats the FIRST ceil(n/2) chars */
			pad.append(insert.substr(0, (int) std::floor(n / 2.0)));
			pad.append(L - n, padCh);
			pad.append(insert.substr(0, (int) std::ceil(n / 2.0)));
		}
		break;
	}
	return pad;
}

/* buildGlobalAlign (src/BandedHMMP7.cpp:1008-1081) */
inline HmmAlignment buildGlobalAlign(const Hmm& h, const std::string& seq, double minScore,
		int alnStart, int alnEnd, int alnFrom, int alnTo, const std::string& tr) {
	HmmAlignment aln;
	const int L = h.L;
	std::string seqN = seq.substr(0, alnFrom - 1);
	std::string seqC = (size_t) alnTo <= seq.size() ? seq.substr(alnTo, L - alnTo) : std::string();
	int csStart = h.p2cs[alnStart], csEnd = h.p2cs[alnEnd];
	int j = 0, k = 0;
	std::string insert;
	for(size_t s = 0; s < tr.size(); ++s) {
		switch(tr[s]) {
		case 'B':
			aln.align.append(paddingSeq(csStart - 1, seqN, '.', PAD_RIGHT));
			j = alnFrom; k = alnStart;
			break;
		case 'M':
			if(k > 1 && s > 1 && h.p2cs[k] - h.p2cs[k - 1] > 1)
				aln.align.append(paddingSeq(h.p2cs[k] - h.p2cs[k - 1] - 1, insert, '-', PAD_JUSTIFIED));
			insert.clear();
			aln.align.push_back(seq.at(j - 1));
			j++; k++;
			break;
		case 'I':
			insert.clear();
			while(s < tr.size() && tr[s] == 'I') {
				insert.push_back((char) ::tolower(seq.at(j - 1)));
				j++; s++;
			}
			s--;
			break;
		case 'D':
			if(k > 1 && h.p2cs[k] - h.p2cs[k - 1] > 1)
				aln.align.append(h.p2cs[k] - h.p2cs[k - 1] - 1, '-');
			aln.align.push_back('-');
			k++;
			break;
		case 'E':
			aln.align.append(paddingSeq(L - csEnd, seqC, '.', PAD_LEFT));
			break;
		}
	}
	aln.K = h.K; aln.L = L;
	aln.seqStart = alnFrom; aln.seqEnd = alnTo;
	aln.hmmStart = alnStart; aln.hmmEnd = alnEnd;
	aln.csStart = csStart; aln.csEnd = csEnd;
	aln.cost = minScore;
	aln.trace = tr;
	return aln;
}

/* alignSeq after the seed lookup (src/HmmUFOtu_main.cpp:86-104) */
inline HmmAlignment alignSeq(const Hmm& h, VitWork& w, const std::string& read, const std::vector<VPath>& vpaths) {
	const int N = (int) read.size();
	std::vector<int8_t> x(N);
	for(int i = 0; i < N; ++i) x[i] = ABC.encode(read[i]);
	Viterbi v(h, w, x.data(), N);
	double minScore; int aS, aE, aF, aT; std::string tr;
	bool usedFull = false, ok = false;
	if(!vpaths.empty()) {
		v.banded(vpaths);
		ok = v.trace(minScore, aS, aE, aF, aT, tr);
		if(!ok) { w.reset(); v.full(); usedFull = true; ok = v.trace(minScore, aS, aE, aF, aT, tr); }
	}
	else { v.full(); usedFull = true; ok = v.trace(minScore, aS, aE, aF, aT, tr); }
	HmmAlignment aln;
	if(ok) aln = buildGlobalAlign(h, read, minScore, aS, aE, aF, aT, tr);
	else { aln.K = h.K; aln.L = h.L; aln.cost = INF; }
	aln.usedFull = usedFull;
	w.reset();
	return aln;
}

} // namespace orc
