// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
//
// CPU restatement of the HmmUFOtu per-read assignment path (reference v1.5.1).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  The product (hmmufotu_amd/) never links this code.
//
// PARITY UNPINNED: the reference cannot be compiled here (Eigen3/Boost absent)
// and its own tests hold no numeric vectors for this path.  The
// only reference-produced numbers available are the trained model files under
// data/*.sm (Q matrix of the GTR model, beta of TN93/HKY85/F81); the oracle is
// pinned against those plus mathematical invariants (tests/test_oracle_*.py).
//
// This header: DNA substitution models  P(t)  and the discrete-Gamma rates.
//   GTR   : src/GTR.h:116-121, src/GTR.cpp:124-145, src/DNASubModel.cpp:123-126
//   TN93  : src/TN93.h:113-154      HKY85 : src/HKY85.h:111-153
//   F81   : src/F81.h:110-118       K80   : src/K80.h:98-118
//   JC69  : src/JC69.h:97-101
#pragma once
#include <cmath>
#include <cstring>
#include <algorithm>

namespace orc {

enum ModelType { M_GTR = 0, M_TN93 = 1, M_HKY85 = 2, M_F81 = 3, M_K80 = 4, M_JC69 = 5 };

// 4x4 matrices are row-major: P[i*4+j] = P(i -> j)
struct Model {
	int type;
	double pi[4];
	double kr, ky, kappa, beta;
	double R[16], Q[16];          // GTR only
	double U[16], U1[16], lam[4]; // GTR eigen system: Q = U diag(lam) U1

	void Pr(double v, double* P) const;
};

// cyclic Jacobi for a symmetric 4x4; A is destroyed, V gets eigenvectors in columns
inline void jacobi4(double A[16], double V[16], double w[4]) {
	for(int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
	for(int sweep = 0; sweep < 100; ++sweep) {
		double off = 0;
		for(int p = 0; p < 4; ++p) for(int q = p + 1; q < 4; ++q) off += A[p*4+q] * A[p*4+q];
		if(off < 1e-300) break;
		for(int p = 0; p < 4; ++p) for(int q = p + 1; q < 4; ++q) {
			double apq = A[p*4+q];
			if(apq == 0) continue;
			double theta = (A[q*4+q] - A[p*4+p]) / (2 * apq);
			double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
			double c = 1 / std::sqrt(t * t + 1), s = t * c;
			for(int k = 0; k < 4; ++k) { // A <- A J
				double akp = A[k*4+p], akq = A[k*4+q];
				A[k*4+p] = c * akp - s * akq;
				A[k*4+q] = s * akp + c * akq;
			}
			for(int k = 0; k < 4; ++k) { // A <- J^T A
				double apk = A[p*4+k], aqk = A[q*4+k];
				A[p*4+k] = c * apk - s * aqk;
				A[q*4+k] = s * apk + c * aqk;
			}
			for(int k = 0; k < 4; ++k) {
				double vkp = V[k*4+p], vkq = V[k*4+q];
				V[k*4+p] = c * vkp - s * vkq;
				V[k*4+q] = s * vkp + c * vkq;
			}
		}
	}
	for(int i = 0; i < 4; ++i) w[i] = A[i*4+i];
}

/* GTR::setQfromParams (src/GTR.cpp:124-145): Q(i,j) = R(i,j)*pi(j), diagonal = -rowsum,
 * rescaled so that -sum_i pi_i Q_ii = 1 (DNASubModel::scale with pi = Ones() !! note the
 * reference passes the DEFAULT pi = Vector4d::Ones(), src/DNASubModel.h:154, so
 * beta = sum_i Q_ii, not the pi-weighted rate).  The reference then eigen-decomposes the
 * non-symmetric Q with Eigen::EigenSolver; the spectral form is unique, so we get the same
 * U diag(lam) U^-1 through the similarity transform S = D^1/2 Q D^-1/2 (symmetric). */
inline void gtr_setup(Model& m) {
	double* Q = m.Q;
	for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j) Q[i*4+j] = m.R[i*4+j] * m.pi[j];
	for(int i = 0; i < 4; ++i) {
		Q[i*4+i] = 0;
		double s = 0;
		for(int j = 0; j < 4; ++j) if(j != i) s += Q[i*4+j];
		Q[i*4+i] = -s;
	}
	double beta = 0;
	for(int i = 0; i < 4; ++i) beta += 1.0 * Q[i*4+i]; /* pi = Ones() */
	for(int i = 0; i < 16; ++i) Q[i] = Q[i] / -beta * 1.0;
	double S[16], V[16];
	double sq[4], isq[4];
	for(int i = 0; i < 4; ++i) { sq[i] = std::sqrt(m.pi[i]); isq[i] = 1 / sq[i]; }
	for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j) S[i*4+j] = sq[i] * Q[i*4+j] * isq[j];
	for(int i = 0; i < 4; ++i) for(int j = i + 1; j < 4; ++j) { // enforce symmetry
		double a = 0.5 * (S[i*4+j] + S[j*4+i]);
		S[i*4+j] = S[j*4+i] = a;
	}
	jacobi4(S, V, m.lam);
	for(int i = 0; i < 4; ++i) for(int k = 0; k < 4; ++k) {
		m.U[i*4+k] = isq[i] * V[i*4+k];
		m.U1[k*4+i] = V[i*4+k] * sq[i];
	}
}

inline void Model::Pr(double v, double* P) const {
	const int A = 0, C = 1, G = 2, T = 3;
	switch(type) {
	case M_GTR: {
		if(v == 0) { for(int i = 0; i < 16; ++i) P[i] = (i % 5 == 0) ? 1.0 : 0.0; return; }
		double e[4];
		for(int k = 0; k < 4; ++k) e[k] = std::exp(lam[k] * v);
		for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j) {
			double s = 0;
			for(int k = 0; k < 4; ++k) s += (U[i*4+k] * e[k]) * U1[k*4+j];
			P[i*4+j] = s;
		}
		return;
	}
	case M_TN93: case M_HKY85: {
		double a = pi[A], c = pi[C], g = pi[G], t = pi[T];
		double kR = type == M_TN93 ? kr : kappa;
		double kY = type == M_TN93 ? ky : kappa;
		double e = std::exp(-beta * v);
		double eR = std::exp(-(1 + (a + g) * (kR - 1)) * beta * v);
		double eY = std::exp(-(1 + (c + t) * (kY - 1)) * beta * v);
		P[A*4+A] = (a * (a + g + (c + t) * e) + g * eR) / (a + g);
		P[A*4+C] = c * (1 - e);
		P[A*4+G] = (g * (a + g + (c + t) * e) - g * eR) / (a + g);
		P[A*4+T] = t * (1 - e);
		P[C*4+A] = a * (1 - e);
		P[C*4+C] = (c * (c + t + (a + g) * e) + t * eY) / (c + t);
		P[C*4+G] = g * (1 - e);
		P[C*4+T] = (t * (c + t + (a + g) * e) - t * eY) / (c + t);
		P[G*4+A] = (a * (a + g + (c + t) * e) - a * eR) / (a + g);
		P[G*4+C] = c * (1 - e);
		P[G*4+G] = (g * (a + g + (c + t) * e) + a * eR) / (a + g);
		P[G*4+T] = t * (1 - e);
		P[T*4+A] = a * (1 - e);
		P[T*4+C] = (c * (c + t + (a + g) * e) - c * eY) / (c + t);
		P[T*4+G] = g * (1 - e);
		P[T*4+T] = (t * (c + t + (a + g) * e) + c * eY) / (c + t);
		if(P[A*4+G] < 0) P[A*4+G] = 0;
		if(P[C*4+T] < 0) P[C*4+T] = 0;
		if(P[G*4+A] < 0) P[G*4+A] = 0;
		if(P[T*4+C] < 0) P[T*4+C] = 0;
		return;
	}
	case M_F81: {
		double e = std::exp(-beta * v);
		for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j)
			P[i*4+j] = i == j ? e + pi[j] * (1 - e) : pi[j] * (1 - e);
		return;
	}
	case M_K80: {
		double e = std::exp(-4 * beta * v);
		double eV = std::exp(-2 * (1 + kappa) * beta * v);
		for(int i = 0; i < 16; ++i) P[i] = (1.0 - e) / 4;
		for(int i = 0; i < 4; ++i) P[i*4+i] = (1.0 + e + 2 * eV) / 4;
		P[A*4+G] = P[G*4+A] = P[C*4+T] = P[T*4+C] = (1.0 + e - 2 * eV) / 4;
		return;
	}
	default: { /* JC69 */
		double off = (1 - std::exp(-4 * v / 3)) / 4;
		double dia = (1 + 3 * std::exp(-4 * v / 3)) / 4;
		for(int i = 0; i < 16; ++i) P[i] = off;
		for(int i = 0; i < 4; ++i) P[i*4+i] = dia;
		return;
	}
	}
}

/* par: GTR -> R[16] row-major; TN93 -> kr,ky,beta; HKY85 -> kappa,beta; F81 -> beta;
 * K80 -> kappa (beta = 1/(2 kappa), src/K80.h:98-100); JC69 -> none.  pi ignored for K80/JC69. */
inline Model make_model(int type, const double* pi, const double* par) {
	Model m;
	std::memset(&m, 0, sizeof(m));
	m.type = type;
	for(int i = 0; i < 4; ++i) m.pi[i] = (type == M_K80 || type == M_JC69) ? 0.25 : pi[i];
	switch(type) {
	case M_GTR: std::memcpy(m.R, par, sizeof(double) * 16); gtr_setup(m); break;
	case M_TN93: m.kr = par[0]; m.ky = par[1]; m.beta = par[2]; break;
	case M_HKY85: m.kappa = par[0]; m.beta = par[1]; break;
	case M_F81: m.beta = par[0]; break;
	case M_K80: m.kappa = par[0]; m.beta = 1 / (2 * m.kappa); break;
	default: break;
	}
	return m;
}

} // namespace orc
