"""GPU parity tests (-m gpu) of the gradient-enhanced surrogate (reference:
src/surrogates/GradientGP.jl, src/acquisition_functions/gradNormUCB.jl) through the C-ABI, against the mpmath
golden vectors (incl. the reference's own closed-form test, test/test_surrogates.jl:291-348) and the CPU oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth
from oracle import gp_oracle as O
from oracle import grad_oracle as G

from tests.parity_record import check
from tests.test_gpu_parity import FAMS

GOLD = os.path.join(os.path.dirname(__file__), "golden")
GRAD = json.load(open(os.path.join(GOLD, "grad_small.json")))


def make_grad(family, ell, sf2, noise, p, mean_c=None, **kw):
    mean = abo.gradConstMean(mean_c) if mean_c is not None else None
    return abo.GradientGP(sf2 * abo.with_lengthscale(FAMS[family](), ell), p, noise, mean=mean, **kw)


@pytest.mark.parametrize("i", range(len(GRAD)))
def test_gradient_gp_golden(i):
    c = GRAD[i]
    p = len(c["X"][0]) + 1
    m = abo.update(make_grad(c["family"], c["ell"], c["sigma_f2"], c["noise_var"], p, c["mean_c"]), c["X"], c["Ys"])
    Z = np.array(c["Z"])
    M = len(Z)
    gm = abo.posterior_grad_mean(m, Z).reshape(p, M).T          # by outputs → (M, p)
    gv = abo.posterior_grad_var(m, Z).reshape(p, M).T
    mu_pm, cov, _ = abo.posterior_grad_cov(m, Z, return_all=True)
    for j in range(M):
        np.testing.assert_allclose(gm[j], c["mu"][j], rtol=0, atol=1e-11)
        np.testing.assert_allclose(mu_pm[j], c["mu"][j], rtol=0, atol=1e-11)
        np.testing.assert_allclose(cov[j], c["cov"][j], rtol=0, atol=1e-11)
        np.testing.assert_allclose(gv[j], np.diag(np.array(c["cov"][j])), rtol=0, atol=1e-11)
    np.testing.assert_allclose(abo.posterior_mean(m, Z), [v[0] for v in c["mu"]], rtol=0, atol=1e-11)
    np.testing.assert_allclose(abo.posterior_var(m, Z), [np.array(v)[0, 0] for v in c["cov"]], rtol=0, atol=1e-11)
    assert abs(abo.nlml_fitted(m) - c["nlml"]) < 1e-10
    one = abo.posterior_grad_cov(m, [c["Z"][0]])
    assert one.shape == (p, p)


def test_reference_closed_form_case_shapes_and_copy():
    # test/test_surrogates.jl:289-348 and :399-414
    gp = abo.GradientGP(abo.SqExponentialKernel(), 3, 0.1)
    xs = [[0.0, 0.0], [0.5, 0.5], [1.0, 1.0]]
    ys = [[1.0, 0.1, 0.1], [0.5, 0.0, 0.0], [0.0, -0.1, -0.1]]
    m = abo.update(gp, xs, ys)
    assert m.noise_var == 0.1 and m.p == 3 and m.gpx is not None and gp.gpx is None
    tx = [[0.25, 0.25]]
    assert abo.posterior_var(m, tx)[0] >= 0.0
    assert len(abo.posterior_grad_mean(m, tx)) == 3 and len(abo.posterior_grad_var(m, tx)) == 3
    c = abo.copy(m)
    assert c.noise_var == m.noise_var and c.p == m.p and c.gp == m.gp and c.gpx is not m.gpx
    with pytest.raises(abo.DimensionMismatch):
        abo.update(gp, xs, [[1.0, 0.1], [0.5, 0.0], [0.0, -0.1]])
    with pytest.raises(abo.DimensionMismatch):
        abo.update(abo.GradientGP(abo.SqExponentialKernel(), 4, 0.1), xs, [[1.0, 0.1, 0.1, 0.0]] * 3)
    with pytest.raises(abo.DimensionMismatch):
        abo.append(m, [0.1, 0.2], 0.0)              # an observation of a gradient-enhanced model has p values


@pytest.mark.parametrize("family,d,N,M,ell,noise", [(O.SE, 2, 60, 500, 0.6, 1e-3), (O.MATERN52, 3, 100, 700, 0.8, 1e-3),
                                                    (O.MATERN72, 4, 90, 300, 1.1, 1e-2), (O.MATERN52, 8, 150, 400, 1.5, 1e-2),
                                                    # round 6: beyond 32 inputs (GradientGP.jl:617-639 has no limit) — the slab
                                                    # generator kgen_grad_wide_kernel; 41·40 = 1640 factor rows: the AUTO engine is the
                                                    # int8 one, fed by the quantiser pass (the slab generator writes fp64 only)
                                                    (O.SE, 40, 20, 60, 2.5, 1e-2), (O.MATERN52, 70, 12, 40, 4.0, 1e-2),
                                                    (O.MATERN52, 40, 40, 64, 3.0, 1e-2), (O.MATERN72, 128, 5, 20, 6.0, 1e-2)])
def test_gradient_gp_against_oracle(family, d, N, M, ell, noise):
    p = d + 1
    X = synth.points(1, N, d)
    f = np.sin(2 * np.pi * X).sum(axis=1) / np.sqrt(d)
    gF = 2 * np.pi * np.cos(2 * np.pi * X) / np.sqrt(d)
    Ys = np.column_stack([f + 0.4, gF])
    Z = synth.points(2, M, d)
    mean_c = np.concatenate([[0.4], np.zeros(d)])
    st = G.fit(family, ell, 1.2, noise, mean_c, X, Ys)
    m = abo.update(make_grad(family, ell, 1.2, noise, p, mean_c), X, Ys)
    L, alpha, Linv = abo.get_factor(m)
    assert L.shape == (p * N, p * N)
    # the library keeps the (d+1)N-row system point-major (row i·p + q) where the reference orders by outputs (q·N + i):
    # the Cholesky factor of a permuted matrix is a different matrix, so compare what both factorise; α crosses the ABI
    # by outputs
    idx = np.array([q * N + i for i in range(N) for q in range(p)])
    K_o = st.L @ st.L.T
    assert np.max(np.abs(L @ L.T - K_o[np.ix_(idx, idx)])) < 1e-9
    assert np.max(np.abs(Linv @ L - np.eye(p * N))) < 1e-8
    assert np.max(np.abs(alpha - st.alpha)) < 1e-6 * max(1, np.max(np.abs(st.alpha)))
    mu_o, var_o = G.predict_grad(st, Z)
    assert np.max(np.abs(abo.posterior_grad_mean(m, Z) - mu_o)) < 1e-8
    assert np.max(np.abs(abo.posterior_grad_var(m, Z) - var_o)) < 1e-8 * max(1.0, 1.2 / ell ** 2)
    mf, vf = G.predict(st, Z)
    mu, var = abo.mean_and_var(m, Z)
    assert np.max(np.abs(mu - mf)) < 1e-8 and np.max(np.abs(var - vf)) < 1e-8
    # the standard acquisition functions run on the function output of the gradient-enhanced model
    best = float(f.min() + 0.4)
    ei = abo.ExpectedImprovement(0.01, best)(m, Z)
    np.testing.assert_allclose(ei, O.expected_improvement(mu, var, best, 0.01), rtol=1e-9, atol=1e-13)
    # GradientNormUCB (gradNormUCB.jl:43-51)
    s = abo.GradientNormUCB(2.0)(m, Z[:64])
    np.testing.assert_allclose(s, G.grad_norm_ucb(st, Z[:64], 2.0), rtol=1e-7, atol=1e-8)


def test_gradient_gp_standardisation_helpers():
    # test/test_surrogates.jl:363-397
    gp = abo.GradientGP(abo.SqExponentialKernel(), 3, 0.1)
    y_train = [[1.0, 0.1, 0.1], [2.0, 0.2, 0.2], [3.0, 0.3, 0.3]]
    mu, sd = abo.get_mean_std(gp, y_train, "mean_scale")
    assert mu[0] == pytest.approx(2.0) and mu[1] == 0.0 and mu[2] == 0.0 and sd[0] > 0 and sd[1] == sd[0] == sd[2]
    ys = abo.std_y(gp, y_train, mu, sd)
    for yo, y_s in zip(y_train, ys):
        for q in range(3):
            assert y_s[q] == pytest.approx((yo[q] - mu[q]) / sd[q], abs=1e-8)
    r = abo.rescale_model(gp, sd)
    assert abo.get_scale(r)[0] == pytest.approx(1.0 / sd[0] ** 2) and r.noise_var == pytest.approx(0.1 / sd[0] ** 2)
    assert abo._get_minimum(gp, y_train) == 1.0
    g2 = abo.update(gp, [[0.0, 0.0], [1.0, 1.0], [0.3, 0.8]], ys)
    mu_u, var_u = abo.unstandardized_mean_and_var(g2, [[0.2, 0.2]], (mu, sd))
    assert mu_u.shape == (1, 3) and var_u.shape == (1, 3)
    acq = abo.GradientNormUCB(1.5)
    assert abo.update(acq, y_train, g2) is acq and abo.copy(acq) == acq


def test_gradient_gp_nlml_and_hyperparameter_mle():
    # nlml(model::GradientGP, params, xs, ys) (GradientGP.jl:684-698) and the generic optimize_hyperparameters
    # (bayesian_opt.jl:196-328) on the gradient-enhanced model
    d, N = 2, 40
    X = synth.points(1, N, d)
    f = np.sin(2 * np.pi * X).sum(axis=1) / np.sqrt(d)
    gF = 2 * np.pi * np.cos(2 * np.pi * X) / np.sqrt(d)
    Ys = np.column_stack([f, gF])
    gp = make_grad(O.MATERN52, 0.9, 1.0, 1e-3, d + 1)
    for le, ls in [(np.log(0.9), 0.0), (np.log(0.4), np.log(2.5))]:
        st = G.fit(O.MATERN52, float(np.exp(le)), float(np.exp(ls)), 1e-3, np.zeros(d + 1), X, Ys)
        want = G.nlml(st)
        assert abs(abo.nlml(gp, [le, ls], X, Ys) - want) <= 1e-9 * max(1.0, abs(want))
        assert abo.nlml_ls(gp, le, ls, X, Ys) == abo.nlml(gp, [le, ls], X, Ys)
    v0, g0 = abo.gradient_gp.nlml_and_grad(gp, [np.log(0.9), 0.0], X, Ys)          # analytic (abo_nlml_grad)
    v1, g1 = abo.gradient_gp.nlml_and_grad_fd(gp, [np.log(0.9), 0.0], X, Ys)       # differences of the device NLML
    assert v0 == v1 and np.max(np.abs(g0 - g1)) <= 1e-7 * max(1.0, np.max(np.abs(g1)))
    for fam, ell in ((O.SE, 0.7), (O.MATERN72, 1.1)):                              # every family's derivative triple
        gq = make_grad(fam, ell, 1.3, 1e-3, d + 1)
        pq = [np.log(ell), np.log(1.3)]
        ga, gf = abo.gradient_gp.nlml_and_grad(gq, pq, X, Ys)[1], abo.gradient_gp.nlml_and_grad_fd(gq, pq, X, Ys)[1]
        assert np.max(np.abs(ga - gf)) <= 1e-7 * max(1.0, np.max(np.abs(gf))), (fam, ga, gf)
    h = 1e-5
    for c in range(2):
        e = np.zeros(2); e[c] = h
        pp, pm = np.array([np.log(0.9), 0.0]) + e, np.array([np.log(0.9), 0.0]) - e
        fd = (G.nlml(G.fit(O.MATERN52, float(np.exp(pp[0])), float(np.exp(pp[1])), 1e-3, np.zeros(d + 1), X, Ys))
              - G.nlml(G.fit(O.MATERN52, float(np.exp(pm[0])), float(np.exp(pm[1])), 1e-3, np.zeros(d + 1), X, Ys))) / (2 * h)
        assert abs(g0[c] - fd) <= 1e-5 * max(1.0, abs(fd))
    # the same beyond 32 inputs (the slab generator's ∂/∂log ℓ build)
    dw, Nw = 40, 12
    Xw = synth.points(3, Nw, dw)
    Yw = np.column_stack([np.sin(2 * np.pi * Xw).sum(axis=1) / np.sqrt(dw), 2 * np.pi * np.cos(2 * np.pi * Xw) / np.sqrt(dw)])
    gw = make_grad(O.MATERN52, 2.5, 1.1, 1e-2, dw + 1)
    pw = [np.log(2.5), np.log(1.1)]
    stw = G.fit(O.MATERN52, 2.5, 1.1, 1e-2, np.zeros(dw + 1), Xw, Yw)
    assert abs(abo.nlml(gw, pw, Xw, Yw) - G.nlml(stw)) <= 1e-9 * max(1.0, abs(G.nlml(stw)))
    ga, gf = abo.gradient_gp.nlml_and_grad(gw, pw, Xw, Yw)[1], abo.gradient_gp.nlml_and_grad_fd(gw, pw, Xw, Yw)[1]
    assert np.max(np.abs(ga - gf)) <= 1e-7 * max(1.0, np.max(np.abs(gf))), (ga, gf)
    dom = abo.ContinuousDomain(np.zeros(d), np.ones(d))
    new = abo.optimize_hyperparameters(gp, X, Ys, [np.log(0.9), 0.0], domain=dom, rng=np.random.default_rng(0))
    assert isinstance(new, abo.GradientGP) and new.gpx is None and new.p == d + 1
    p1 = [np.log(abo.get_lengthscale(new)[0]), np.log(abo.get_scale(new)[0])]
    assert abo.nlml(new, p1, X, Ys) < v0 - 1e-3


@pytest.mark.parametrize("family,d,N0,n_app,ell,noise,n_max", [
    (O.SE, 2, 20, 6, 0.6, 1e-3, 64),
    (O.MATERN52, 3, 40, 9, 0.8, 1e-3, 64),          # 160 rows → 196: crosses the 128-row padding boundary twice
    (O.MATERN72, 4, 30, 5, 1.1, 1e-2, 0),           # no spare capacity: the first append refits with room to grow
    (O.MATERN52, 36, 8, 3, 2.5, 1e-2, 16),          # beyond 32 inputs: the appended kernel rows come from the slab generator
])
def test_gradient_gp_append_matches_full_refit(family, d, N0, n_app, ell, noise, n_max):
    """abo_append_grad: p bordered row-appends per observation on the point-major factor.  The reference refits per step
    (update(::GradientGP), GradientGP.jl:659-668), so parity is the from-scratch path on the N + j points: this library's own
    refit and the CPU oracle (oracle/grad_oracle.py)."""
    p = d + 1
    N = N0 + n_app
    X = synth.points(1, N, d)
    f = np.sin(2 * np.pi * X).sum(axis=1) / np.sqrt(d)
    gF = 2 * np.pi * np.cos(2 * np.pi * X) / np.sqrt(d)
    Ys = np.column_stack([f + 0.3, gF])
    mean_c = np.concatenate([[0.3], np.zeros(d)])
    Z = synth.points(2, 300, d)
    m = abo.update(make_grad(family, ell, 1.2, noise, p, mean_c, n_max=n_max), X[:N0], Ys[:N0])
    models = [m]
    for j in range(n_app):
        m = abo.append(m, X[N0 + j], Ys[N0 + j])
        models.append(m)
    ref = abo.update(make_grad(family, ell, 1.2, noise, p, mean_c), X, Ys)
    st = G.fit(family, ell, 1.2, noise, mean_c, X, Ys)
    mu_o, var_o = G.predict_grad(st, Z)
    case = f"grad_append/fam{family}_d{d}_N{N0}+{n_app}"
    check(case, "grad_mu", np.max(np.abs(abo.posterior_grad_mean(m, Z) - mu_o)), 1e-7)
    check(case, "grad_var", np.max(np.abs(abo.posterior_grad_var(m, Z) - var_o)) / max(1.0, 1.2 / ell ** 2), 1e-7)
    check(case, "grad_mu_vs_own_refit", np.max(np.abs(abo.posterior_grad_mean(m, Z) - abo.posterior_grad_mean(ref, Z))), 1e-7)
    check(case, "nlml_rel", abs(abo.nlml_fitted(m) - G.nlml(st)) / max(1.0, abs(G.nlml(st))), 1e-8)
    L, alpha, Linv = abo.get_factor(m)
    Lr, alpha_r, _ = abo.get_factor(ref)
    check(case, "L_vs_own_refit", np.max(np.abs(L - Lr)), 1e-8)
    check(case, "alpha_rel", np.max(np.abs(alpha - st.alpha)) / max(1.0, np.max(np.abs(st.alpha))), 1e-6)
    assert np.max(np.abs(Linv @ L - np.eye(p * N))) < 1e-7
    Xb, Yb = abo.training_data(m)
    np.testing.assert_array_equal(Xb, X)
    np.testing.assert_array_equal(Yb, Ys)
    # every intermediate model is still valid and unchanged (free rollback)
    k = n_app // 2
    st_k = G.fit(family, ell, 1.2, noise, mean_c, X[:N0 + k], Ys[:N0 + k])
    mu_k, var_k = G.predict(st_k, Z[:50])
    mu_g, var_g = abo.mean_and_var(models[k], Z[:50])
    assert np.max(np.abs(mu_g - mu_k)) < 1e-7 and np.max(np.abs(var_g - var_k)) < 1e-7
    # the NLML gradient needs a freshly fitted model; an appended view says so
    with pytest.raises(ValueError):
        abo._lib.check(abo._lib.lib().abo_nlml_grad(m._require(), None, None, None))


def test_gradient_gp_resident_grid_downdates(monkeypatch):
    """A function-value candidate grid resident with a gradient-enhanced model: after abo_append_grad its posterior is
    down-dated by p rank-1 steps (one per appended row) — through the resident K_ZX and by re-evaluating the kernel — and
    equals the oracle's posterior on the N + j points; EI epilogue and top-k run on the stored posterior."""
    d, N0, n_app = 3, 50, 4
    p = d + 1
    X = synth.points(1, N0 + n_app, d)
    f = np.sin(2 * np.pi * X).sum(axis=1) / np.sqrt(d)
    gF = 2 * np.pi * np.cos(2 * np.pi * X) / np.sqrt(d)
    Ys = np.column_stack([f, gF])
    Z = synth.points(2, 2000, d)
    out = {}
    for mode in ("64", "0"):
        monkeypatch.setenv("ABO_CAND_KZX_GIB", mode)
        m = abo.update(make_grad(O.MATERN52, 0.8, 1.0, 1e-3, p, np.zeros(p), n_max=N0 + 16), X[:N0], Ys[:N0])
        cands = abo.ResidentCandidates(m, Z)
        for j in range(n_app):
            m = abo.append(m, X[N0 + j], Ys[N0 + j])
            cands.downdate(m)
        out[mode] = cands.mean_and_var()
        if mode == "64":
            acq = abo.ExpectedImprovement(0.01, float(f.min()))
            s_r, tv_r, ti_r = cands.evaluate(acq, k=10, return_scores=True)
            s_f, tv_f, ti_f = abo.evaluate(acq, m, Z, k=10)
            np.testing.assert_allclose(s_r, s_f, rtol=0, atol=1e-9)
    st = G.fit(O.MATERN52, 0.8, 1.0, 1e-3, np.zeros(p), X, Ys)
    mu_o, var_o = G.predict(st, Z)
    for mode in ("64", "0"):
        assert np.max(np.abs(out[mode][0] - mu_o)) < 1e-8 and np.max(np.abs(out[mode][1] - var_o)) < 1e-8
