"""Pins of the oracle's substitution models (CPU).

Reference-produced numbers available for this path: the trained model files the reference ships
under data/ (copied to tests/golden/ref_data): the GTR file stores the rate matrix Q the
reference computed from (pi, R) — a known-answer vector for GTR::setQfromParams — and the
TN93/HKY85/F81 files store beta as the reference's setBeta() derived it from (pi, kappa...).
Everything else is pinned by invariants of a reversible Markov chain.
"""
import numpy as np
import pytest

from hmmufotu_amd import synth
from oracle import oracle_py as O

MODELS = ["GTR", "TN93", "HKY85", "F81", "K80", "JC69"]


def _om(name):
    m = synth.load_model(name)
    return m, O.Model(m.type_id, m.pi, m.par)


def test_gtr_Q_matches_reference_file():
    m, om = _om("GTR")
    assert np.abs(om.Q() - m.Q_file).max() < 1e-15          # Q printed by the reference itself
    assert abs(np.trace(om.Q()) + 1) < 1e-14                 # DNASubModel::scale with pi = Ones()


def test_reference_betas():
    a, c, g, t = synth.load_model("TN93").pi
    kr, ky, beta = synth.load_model("TN93").par
    assert abs(beta - 1 / (2 * (a * c + a * t + c * g + g * t + kr * a * g + ky * c * t))) < 1e-12       # src/TN93.h:100-103
    kappa, beta = synth.load_model("HKY85").par
    assert abs(beta - 1 / (2 * (a + g) * (c + t) + 2 * kappa * (a * g + c * t))) < 1e-12                 # src/HKY85.h:100-102
    beta, = synth.load_model("F81").par
    assert abs(beta - 1 / (1 - (a * a + c * c + g * g + t * t))) < 1e-12                               # src/F81.h:100-102


@pytest.mark.parametrize("name", MODELS)
def test_markov_invariants(name):
    m, om = _om(name)
    pi = np.full(4, 0.25) if name in ("K80", "JC69") else m.pi
    assert np.abs(om.P(0.0) - np.eye(4)).max() < 1e-15
    for s, t in ((0.01, 0.02), (0.3, 0.7), (1e-5, 2.0)):
        Ps, Pt, Pst = om.P(s), om.P(t), om.P(s + t)
        assert np.abs(Ps.sum(1) - 1).max() < 1e-13
        assert np.abs(Ps @ Pt - Pst).max() < 1e-13                       # Chapman-Kolmogorov
        assert np.abs(pi[:, None] * Ps - (pi[:, None] * Ps).T).max() < 1e-13   # detailed balance
        assert (Ps >= 0).all()
    assert np.abs(om.P(500.0) - pi[None, :]).max() < 1e-9               # stationary limit


def test_gtr_reduces_to_hky85():
    """GTR with R(i,j) = kappa on transitions, 1 on transversions is HKY85 (up to the time scale)."""
    hk = synth.load_model("HKY85")
    kappa, beta = hk.par
    R = np.ones((4, 4)); np.fill_diagonal(R, 0)
    for i, j in ((0, 2), (2, 0), (1, 3), (3, 1)):
        R[i, j] = kappa
    g = O.Model(0, hk.pi, R.ravel())
    h = O.Model(2, hk.pi, hk.par)
    Q = g.Q()
    rate_gtr = -(Q * np.eye(4)).sum() and Q[0, 1] / hk.pi[1]          # A->C rate per unit pi
    scale = (beta) / rate_gtr                                          # HKY85: Q(A,C) = beta * pi_C
    for t in (0.01, 0.2, 1.0):
        assert np.abs(g.P(t * scale) - h.P(t)).max() < 1e-12


@pytest.mark.parametrize("name", MODELS)
def test_generator_and_engine_spectral_forms_agree_with_oracle(name):
    """three independent implementations of P(t): oracle closed forms (reference formulas),
    numpy generator, and the engine's host-side spectral decomposition (no GPU needed)."""
    from hmmufotu_amd import engine as E
    m, om = _om(name)
    U, lam, U1 = E.model_spectral(E.model_desc(m.type_id, m.pi, m.par))
    for t in (0.0, 1e-6, 0.003, 0.05, 0.4, 3.0):
        Po = om.P(t)
        assert np.abs(synth.model_P(m, t) - Po).max() < 1e-13
        assert np.abs((U * np.exp(lam * t)[None, :]) @ U1 - Po).max() < 1e-13


@pytest.mark.parametrize("name", MODELS)
def test_spectral_form_invariants_the_placement_kernel_relies_on(name):
    """k_place_blk / k_estimate_prod assume: eigenvalue 0 first with U(:,0) = 1 and U^-1(0,:) = pi (component 0 of a
    packed message is pi . e), and U = Pi^-1/2 V with V orthonormal, i.e. sum_i pi_i U_im U_in = delta_mn."""
    from hmmufotu_amd import engine as E
    m, _ = _om(name)
    U, lam, U1 = E.model_spectral(E.model_desc(m.type_id, m.pi, m.par))
    pi = np.full(4, 0.25) if name in ("K80", "JC69") else np.asarray(m.pi, float)
    assert lam[0] == 0.0 and (lam[1:] < 0).all()
    assert (U[:, 0] == 1.0).all() and np.array_equal(U1[0], pi)
    assert np.abs(U @ U1 - np.eye(4)).max() < 1e-13
    assert np.abs((U * pi[:, None]).T @ U - np.eye(4)).max() < 1e-13


def test_dgamma_rates_sum_to_one():
    b, r = synth.dgamma(4, 0.5)
    assert abs(r.sum() - 1) < 1e-12 and (np.diff(r) > 0).all() and b[0] == 0 and np.isinf(b[-1])   # F6: not multiplied by K
