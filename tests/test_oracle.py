"""CPU tests: both oracle restatements against the golden fixtures (tests/golden/*.json, generated
in 60-digit mpmath from the closed forms the reference's tests assert) and against each other."""
import json
import os

import numpy as np
import pytest

from oracle import gp_oracle as O
from tests import oracle_c

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


KAT = _load("kat.json")
RANDOM = _load("random_small.json")


def _run_py(c):
    st = O.fit(c["family"], c["ell"], c["sigma_f2"], c["noise_var"], c["mean_c"], np.array(c["X"]), c["y"])
    mu, var = O.predict(st, np.array(c["Z"]))
    return st, mu, var


def _check_case(c, mu, var, alpha, nlml, tol):
    sf2 = c["sigma_f2"]
    np.testing.assert_allclose(mu, c["mu"], rtol=0, atol=tol * max(1.0, np.max(np.abs(c["mu"]))))
    np.testing.assert_allclose(var, c["var"], rtol=0, atol=tol * sf2)
    np.testing.assert_allclose(alpha, c["alpha"], rtol=0, atol=tol * max(1.0, np.max(np.abs(c["alpha"]))) * 1e3)
    assert abs(nlml - c["nlml"]) <= tol * max(1.0, abs(c["nlml"])) * 10


@pytest.mark.parametrize("name", ["kat1", "kat3", "kat4", "kat5"])
def test_kat_python_oracle(name):
    # reference tolerance for these identities is atol=1e-10 (test/test_surrogates.jl:103-104)
    c = KAT[name]
    st, mu, var = _run_py(c)
    _check_case(c, mu, var, st.alpha, O.nlml(st), 1e-13)
    if "ei" in c:
        np.testing.assert_allclose(O.expected_improvement(mu, var, c["best_y"], c["xi"]), c["ei"], rtol=1e-9, atol=1e-16)
        np.testing.assert_allclose(O.upper_confidence_bound(mu, var, c["beta"]), c["ucb"], rtol=1e-12)
        np.testing.assert_allclose(O.probability_improvement(mu, var, c["best_y"], c["xi"]), c["pi"], rtol=1e-9, atol=1e-16)


def test_kat_published_values():
    # the digits quoted in SURVEY.md §8(c) / BASELINE.md §5
    assert abs(KAT["kat1"]["mu"][0] - 0.1771247751991296) < 1e-15
    assert abs(KAT["kat1"]["var"][0] - 0.050320225208722924) < 1e-15
    assert abs(KAT["kat1"]["nlml"] - 2.6769327097262567) < 1e-14
    assert abs(KAT["kat3"]["ucb"][0] - (-1.0186125700256665)) < 1e-14
    assert abs(KAT["kat4"]["ucb"][0] - (-0.20223654076594783)) < 1e-14
    v = KAT["kat5"]["var"]
    assert v[0] < v[1] < v[2]          # test/test_bayesian_opt.jl:484-485


def test_kat6_must_fail():
    c = KAT["kat6"]
    with pytest.raises(O.NotPositiveDefinite) as e:
        O.fit(c["family"], c["ell"], c["sigma_f2"], c["noise_var"], c["mean_c"], np.array(c["X"]), c["y"])
    assert e.value.info == 3
    lib = oracle_c.load()
    info, _, _ = oracle_c.fit(lib, c["family"], c["ell"], c["sigma_f2"], c["noise_var"], c["mean_c"], np.array(c["X"]), c["y"])
    assert info == 3


@pytest.mark.parametrize("i", range(len(RANDOM)))
def test_random_small_both_oracles(i):
    c = RANDOM[i]
    st, mu, var = _run_py(c)
    cond_slack = 1e-11 / min(1.0, c["noise_var"] / c["sigma_f2"]) * 1e-2
    tol = max(1e-12, cond_slack)
    _check_case(c, mu, var, st.alpha, O.nlml(st), tol)
    np.testing.assert_allclose(O.expected_improvement(mu, var, c["best_y"], c["xi"]), c["ei"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(O.upper_confidence_bound(mu, var, c["beta"]), c["ucb"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(O.probability_improvement(mu, var, c["best_y"], c["xi"]), c["pi"], rtol=1e-6, atol=1e-12)

    lib = oracle_c.load()
    X = np.array(c["X"]); Z = np.array(c["Z"]); y = np.array(c["y"])
    info, L, alpha = oracle_c.fit(lib, c["family"], c["ell"], c["sigma_f2"], c["noise_var"], c["mean_c"], X, y)
    assert info == 0
    mu_c, var_c = oracle_c.predict(lib, c["family"], c["ell"], c["sigma_f2"], c["mean_c"], X, L, alpha, Z)
    _check_case(c, mu_c, var_c, alpha, lib.oracle_nlml(L, alpha, y, c["mean_c"], len(y)), tol)
    np.testing.assert_allclose(L, st.L, rtol=0, atol=1e-12)
    for kind, p0 in ((O.ACQ_EI, c["xi"]), (O.ACQ_UCB, c["beta"]), (O.ACQ_PI, c["xi"])):
        np.testing.assert_allclose(oracle_c.acq(lib, kind, mu_c, var_c, p0, c["best_y"]),
                                   O.acquisition(kind, mu_c, var_c, p0, c["best_y"]), rtol=1e-9, atol=1e-30)


def test_topk_order_matches_stable_reverse_sortperm():
    # sortperm(scores; rev=true): ties keep the lowest index, NaN sorts first (acq_utils.jl:51-52)
    s = np.array([1.0, 3.0, 3.0, -np.inf, np.nan, 2.0, 3.0, np.inf, -0.0, 0.0])
    vals, idx = O.top_k(s, 10)
    assert idx.tolist() == [4, 7, 1, 2, 6, 5, 0, 8, 9, 3] or idx.tolist() == [4, 7, 1, 2, 6, 5, 0, 9, 8, 3]
    vals, idx = O.top_k(s, 3)
    assert idx.tolist() == [4, 7, 1]


def test_matern_taylor_branch_equivalence():
    # ApproxMatern52Kernel Taylor branch (GradientGP.jl:94-101) vs the closed form, at d2 < 1e-10
    for d2 in (0.0, 1e-14, 1e-12, 9.9e-11):
        assert abs((1.0 - 5.0 / 6.0 * d2) - O.kappa(O.MATERN52, d2)) < 2e-15
        assert abs((1.0 - 0.7 * d2) - O.kappa(O.MATERN72, d2)) < 2e-15


def test_standardisation_equivalence():
    # test/test_bayesian_opt.jl:238-356: ZeroMean + mean_only standardisation ≡ ConstMean(mean(y)) prior
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 1, (15, 2)); y = rng.normal(size=15) + 4.0; Z = rng.uniform(0, 1, (7, 2))
    ys, m, s = O.standardize(y, "mean_only")
    a = O.fit(O.SE, 0.7, 1.3, 1e-3, 0.0, X, ys)
    b = O.fit(O.SE, 0.7, 1.3, 1e-3, m, X, y)
    mu_a, var_a = O.predict(a, Z); mu_b, var_b = O.predict(b, Z)
    np.testing.assert_allclose(mu_a + m, mu_b, atol=1e-10)
    np.testing.assert_allclose(var_a, var_b, atol=1e-10)


# ---------------- gradient-enhanced GP oracle (GradientGP) ----------------
from oracle import grad_oracle as G

GRAD = _load("grad_small.json")


@pytest.mark.parametrize("i", range(len(GRAD)))
def test_grad_oracle_against_mpmath_golden(i):
    """Posterior mean and full p×p covariance of all outputs; the golden values differentiate the base kernel
    numerically in 40-digit arithmetic (mp.diff), independent of the analytic φ', φ''.  Case 0 is the reference's
    own closed-form test (test/test_surrogates.jl:291-348, atol 1e-10)."""
    c = GRAD[i]
    st = G.fit(c["family"], c["ell"], c["sigma_f2"], c["noise_var"], c["mean_c"], np.array(c["X"]), c["Ys"])
    for j, z in enumerate(c["Z"]):
        mu, C = G.predict_grad(st, [z], cov=True)
        np.testing.assert_allclose(mu, c["mu"][j], rtol=0, atol=1e-12)
        np.testing.assert_allclose(C, c["cov"][j], rtol=0, atol=1e-12)
        mu2, var2 = G.predict_grad(st, [z])
        np.testing.assert_allclose(var2, np.diag(C), rtol=0, atol=1e-14)
    assert abs(G.nlml(st) - c["nlml"]) < 1e-11
    mu_f, var_f = G.predict(st, np.array(c["Z"]))
    np.testing.assert_allclose(mu_f, [m[0] for m in c["mu"]], atol=1e-12)


def test_grad_kernel_blocks_against_finite_differences():
    # test/test_surrogates.jl:236-287: gradKernel ≡ derivatives of the base kernel
    rng = np.random.default_rng(0)
    for fam in (O.SE, O.MATERN52, O.MATERN72):
        x, z = rng.normal(size=(1, 3)), rng.normal(size=(1, 3))
        K = G.grad_kernel_matrix(fam, 0.7, 1.9, x, z)
        k = lambda a, b: O.kernel_matrix(fam, 0.7, 1.9, a, b)[0, 0]
        h = 1e-5
        for c in range(3):
            e = np.zeros((1, 3)); e[0, c] = h
            assert abs(K[c + 1, 0] - (k(x + e, z) - k(x - e, z)) / (2 * h)) < 1e-8
            assert abs(K[0, c + 1] - (k(x, z + e) - k(x, z - e)) / (2 * h)) < 1e-8
            for c2 in range(3):
                e2 = np.zeros((1, 3)); e2[0, c2] = h
                fd = (k(x + e, z + e2) - k(x + e, z - e2) - k(x - e, z + e2) + k(x - e, z - e2)) / (4 * h * h)
                assert abs(K[c + 1, c2 + 1] - fd) < 5e-6
    Kxx = G.grad_kernel_matrix(O.MATERN52, 0.7, 1.9, rng.normal(size=(5, 2)))
    np.testing.assert_allclose(Kxx, Kxx.T, atol=1e-15)
    assert np.all(np.linalg.eigvalsh(Kxx) > -1e-10)


def test_c_oracle_under_address_and_ub_sanitizers():
    """sanitizers run on the CPU build only (none exist for the GPU on this pool).  oracle/selftest.c drives
    the plain-C restatement through the reference's closed-form cases, the not-PD case, an empty training set and a
    heap-allocated random case under -fsanitize=address,undefined."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "asan-check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
