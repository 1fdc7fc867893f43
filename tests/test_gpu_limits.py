"""GPU tests (-m gpu): the library carries no size limits the reference does not have — the reference is
dimension-agnostic (src/surrogates/StandardGP.jl:79-83) and n_local is a free Int (acq_utils.jl:33-52)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth
from oracle import gp_oracle as O
from oracle import grad_oracle as G

from tests.test_gpu_parity import make_model
from tests.test_gpu_gradient_gp import make_grad


@pytest.mark.parametrize("family,d,N,M,ell", [(O.MATERN52, 40, 300, 700, 4.0), (O.SE, 64, 200, 513, 5.0),
                                              (O.MATERN72, 33, 130, 100, 3.5), (O.MATERN32, 100, 64, 50, 6.0)])
def test_dimensions_beyond_32_against_oracle(family, d, N, M, ell):
    X = synth.points(1, N, d)
    Z = synth.points(2, M, d) * 1.2 - 0.1
    y = synth.objective(X, 0.05)
    sf2, noise = 1.3, 1e-3
    st = O.fit(family, ell, sf2, noise, 0.0, X, y)
    mu_o, var_o = O.predict(st, Z)
    m = abo.update(make_model(family, ell, sf2, noise, n_max=N + 8), X, y)
    L, alpha, _ = abo.get_factor(m)
    assert np.max(np.abs(L - st.L)) < 1e-11
    assert np.max(np.abs(alpha - st.alpha)) < 1e-8 * max(1.0, np.max(np.abs(st.alpha)))
    mu, var = abo.mean_and_var(m, Z)
    assert np.max(np.abs(mu - mu_o)) < 1e-10 * max(1.0, np.max(np.abs(mu_o)))
    assert np.max(np.abs(var - var_o)) < 1e-10 * sf2
    assert abs(abo.nlml_fitted(m) - O.nlml(st)) < 1e-9 * max(1.0, abs(O.nlml(st)))
    # incremental path and the NLML gradient take the same wide-d kernels
    x_new = synth.points(5, 1, d)[0]
    m2 = abo.append(m, x_new, 0.25)
    st2 = O.fit(family, ell, sf2, noise, 0.0, np.vstack([X, x_new]), np.append(y, 0.25))
    mu2, var2 = abo.mean_and_var(m2, Z[:50])
    mo2, vo2 = O.predict(st2, Z[:50])
    assert np.max(np.abs(mu2 - mo2)) < 1e-9 and np.max(np.abs(var2 - vo2)) < 1e-9
    cands = abo.ResidentCandidates(m, Z)
    cands.downdate(m2)
    mu_c, var_c = cands.mean_and_var()
    mo, vo = O.predict(st2, Z)
    assert np.max(np.abs(mu_c - mo)) < 1e-9 and np.max(np.abs(var_c - vo)) < 1e-9
    gp = make_model(family, ell, sf2, noise)
    v, g = abo.nlml_and_grad(gp, [np.log(ell), np.log(sf2)], X, y)
    h = 1e-5
    fd = [(O.nlml(O.fit(family, float(np.exp(np.log(ell) + h)), sf2, noise, 0.0, X, y))
           - O.nlml(O.fit(family, float(np.exp(np.log(ell) - h)), sf2, noise, 0.0, X, y))) / (2 * h),
          (O.nlml(O.fit(family, ell, float(np.exp(np.log(sf2) + h)), noise, 0.0, X, y))
           - O.nlml(O.fit(family, ell, float(np.exp(np.log(sf2) - h)), noise, 0.0, X, y))) / (2 * h)]
    np.testing.assert_allclose(g, fd, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("M,k", [(5000, 1025), (5000, 3000), (5000, 5000), (700, 2500), (4096, 2048)])
def test_selection_of_more_than_1024(M, k):
    """sortperm(scores; rev=true)[1:min(n_local, M)] for any n_local: rounds of 1024, bit-exact against the oracle's
    stable sort; the tail of a request longer than the batch is (NaN, −1).  Scores carry ties and NaNs."""
    d = 2
    X, y = synth.standardized_problem(50, d)
    m = abo.update(make_model(O.SE, 0.4, 1.0, 1e-3), X, y)
    Z = synth.points(2, M, d)
    Z[M // 3: M // 3 + 40] = Z[7]              # 41 exact ties
    Z[11, 0] = np.nan
    s, tv, ti = abo.evaluate(abo.UpperConfidenceBound(1.5), m, Z, k=k, idx_base=10)
    ov, oi = O.top_k(s, k)
    kk = min(k, M)
    np.testing.assert_array_equal(ti[:kk], oi[:kk] + 10)
    np.testing.assert_array_equal(tv[:kk], ov[:kk])
    assert np.all(ti[kk:] == -1) and np.all(np.isnan(tv[kk:]))
    cands = abo.ResidentCandidates(m, Z)
    s2, tv2, ti2 = cands.evaluate(abo.UpperConfidenceBound(1.5), k=k, idx_base=10, return_scores=True)
    np.testing.assert_array_equal(ti2, ti)


def test_gradient_gp_beyond_sixteen_inputs():
    d, N, M = 20, 30, 40
    p = d + 1
    X = synth.points(1, N, d)
    f = np.sin(2 * np.pi * X).sum(axis=1) / np.sqrt(d)
    gF = 2 * np.pi * np.cos(2 * np.pi * X) / np.sqrt(d)
    Ys = np.column_stack([f, gF])
    Z = synth.points(2, M, d)
    mean_c = np.zeros(p)
    st = G.fit(O.MATERN52, 2.5, 1.2, 1e-3, mean_c, X, Ys)
    m = abo.update(make_grad(O.MATERN52, 2.5, 1.2, 1e-3, p, mean_c), X, Ys)
    mu_o, var_o = G.predict_grad(st, Z)
    assert np.max(np.abs(abo.posterior_grad_mean(m, Z) - mu_o)) < 1e-8
    assert np.max(np.abs(abo.posterior_grad_var(m, Z) - var_o)) < 1e-8
    s = abo.GradientNormUCB(2.0)(m, Z[:16])
    np.testing.assert_allclose(s, G.grad_norm_ucb(st, Z[:16], 2.0), rtol=1e-7, atol=1e-8)
    # round 6: 33 … 128 inputs run the slab generator (tests/test_gpu_gradient_gp.py); beyond that the library refuses by name
    with pytest.raises(ValueError) as e:
        abo.update(abo.GradientGP(abo.SqExponentialKernel(), 140, 0.1), synth.points(1, 4, 139), np.zeros((4, 140)))
    assert "129" in str(e.value)                # the message names the limit


def test_candidate_set_refuses_a_model_from_another_factor_of_the_same_size():
    """A candidate set remembers the factor it is synced with by a process-wide generation id, not by address: refits
    of the same size land at recycled addresses, and a down-date against such a stranger must be refused."""
    d, N = 3, 100
    X, y = synth.standardized_problem(N + 1, d)
    Z = synth.points(2, 500, d)
    for _ in range(8):
        a = abo.update(make_model(O.SE, 0.6, 1.0, 1e-3, n_max=N + 16), X[:N], y[:N])
        cands = abo.ResidentCandidates(a, Z)
        del a
        cands.model = None
        b = abo.update(make_model(O.SE, 0.9, 2.0, 1e-3, n_max=N + 16), X[:N], y[:N])    # same sizes, recycled buffers
        b1 = abo.append(b, X[N], y[N])
        with pytest.raises(ValueError):
            cands.downdate(b1)


def _random_cases():
    rng = np.random.default_rng(20260)
    cases = [(1, 1, 7), (2, 3, 1), (127, 2, 130), (128, 5, 129), (129, 16, 257), (255, 33, 64), (256, 1, 2049), (257, 7, 100)]
    for _ in range(10):
        cases.append((int(rng.integers(3, 700)), int(rng.integers(1, 41)), int(rng.integers(1, 1500))))
    return cases


@pytest.mark.parametrize("N,d,M", _random_cases())
def test_edge_and_random_shapes_against_oracle(N, d, M):
    """Shapes around every internal boundary (one point, 127 / 128 / 129 rows: fused small fit vs panel chain; 255 / 256 /
    257: 128- vs 256-row contraction tiles; d = 33: slab kernels) and seeded random ones, all kernel families in turn,
    against the oracle; selection bit-exact."""
    fam = [O.SE, O.MATERN52, O.MATERN72, O.MATERN32][(N + d + M) % 4]
    ell = 0.4 * np.sqrt(d) + 0.2
    sf2, noise, mean_c = 1.7, 1e-3, 0.25 * ((N % 3) - 1)
    X = synth.points(1, N, d)
    Z = synth.points(2, M, d) * 1.3 - 0.15
    y = synth.objective(X, 0.05) + mean_c
    st = O.fit(fam, ell, sf2, noise, mean_c, X, y)
    mu_o, var_o = O.predict(st, Z)
    m = abo.update(make_model(fam, ell, sf2, noise, mean_c), X, y)
    mu, var = abo.mean_and_var(m, Z)
    assert np.max(np.abs(mu - mu_o)) <= 1e-9 * max(1.0, np.max(np.abs(mu_o)))
    assert np.max(np.abs(var - var_o)) <= 1e-9 * sf2
    assert abs(abo.nlml_fitted(m) - O.nlml(st)) <= 1e-9 * max(1.0, abs(O.nlml(st)))
    L, alpha, Linv = abo.get_factor(m)
    assert np.max(np.abs(L - st.L)) <= 1e-10 and np.max(np.abs(Linv @ st.L - np.eye(N))) <= 1e-8
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    k = min(M, 37)
    s, tv, ti = abo.evaluate(acq, m, Z, k=k)
    np.testing.assert_allclose(s, O.expected_improvement(mu, var, float(y.min()), 0.01), rtol=1e-10, atol=1e-14)
    ov, oi = O.top_k(s, k)
    np.testing.assert_array_equal(ti, oi)
    # one bordered append on top (refit fallback when the storage has no spare row, in place otherwise)
    x_new = synth.points(9, 1, d)[0]
    m2 = abo.append(m, x_new, 0.1)
    st2 = O.fit(fam, ell, sf2, noise, mean_c, np.vstack([X, x_new]), np.append(y, 0.1))
    mu2, var2 = abo.mean_and_var(m2, Z[:64])
    mo2, vo2 = O.predict(st2, Z[:64])
    assert np.max(np.abs(mu2 - mo2)) <= 1e-8 * max(1.0, np.max(np.abs(mo2))) and np.max(np.abs(var2 - vo2)) <= 1e-8 * sf2


@pytest.mark.parametrize("N,d,M", [(25, 1, 10000), (100, 2, 10000), (129, 3, 3000), (700, 4, 5000), (1600, 4, 2000)])
def test_fused_update_and_evaluate_equals_the_two_calls(N, d, M):
    """abo_fit_acq = abo_fit + abo_acq with one host synchronisation: the same model and the same scores / selection, bit for
    bit, on the one-launch small fit (N ≤ 128), the blocked fit and the int8 engine's sizes"""
    X, y = synth.standardized_problem(N, d, 0.02)
    Z = synth.points(2, M, d)
    gp = make_model(O.MATERN52, 0.5, 1.0, 1e-4)
    acq = abo.ExpectedImprovement(0.01, 123.0)                       # best_y is replaced by min(ys), as update(acq, ys, m) does
    m1 = abo.update(gp, X, y)
    s1, tv1, ti1 = abo.evaluate(abo.update(acq, y, m1), m1, Z, k=100)
    m2, s2, tv2, ti2 = abo.update_and_evaluate(acq, gp, X, y, Z, k=100)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(tv1, tv2)
    np.testing.assert_array_equal(ti1, ti2)
    for a, b in zip(abo.get_factor(m1), abo.get_factor(m2)):
        np.testing.assert_array_equal(a, b)
    assert abo.nlml_fitted(m1) == abo.nlml_fitted(m2)
    t = m2.timings()
    assert t["fit_total_ms"] > 0 and t["acq_total_ms"] > 0
    # the fused call's model is a normal model: appendable, copyable, refinable
    np.testing.assert_array_equal(abo.posterior_mean(abo.copy(m2), Z[:50]), abo.posterior_mean(m1, Z[:50]))


def test_fused_call_reports_a_failed_factorisation_and_leaves_the_handle_unfitted():
    """test_bayesian_opt.jl:749-786 through the fused entry: PosDefException(3), nothing usable left behind, no crash in the
    acquisition launches that were queued behind the failing fit"""
    X = np.array([[-1.0, -1.0], [5.0, -5.0], [-1.0 + 1e-12, -1.0 + 1e-12]])
    y = np.array([1.0, 2.0, 1.0])
    Z = synth.points(2, 500, 2)
    gp = abo.HipStandardGP(abo.SqExponentialKernel(), 0.0)
    with pytest.raises(abo.PosDefException) as e:
        abo.update_and_evaluate(abo.UpperConfidenceBound(2.0), gp, X, y, Z, k=10)
    assert e.value.info == 3
    # the same on the blocked path (N > 128): a duplicated point far down the matrix
    X2 = synth.points(1, 300, 2)
    X2[250] = X2[17]
    gp2 = abo.HipStandardGP(abo.with_lengthscale(abo.SqExponentialKernel(), 0.01), 0.0)     # K ≈ I except for the duplicated point
    with pytest.raises(abo.PosDefException) as e:
        abo.update_and_evaluate(abo.UpperConfidenceBound(2.0), gp2, X2, np.zeros(300), Z, k=10)
    assert e.value.info == 251
    with pytest.raises(abo.PosDefException) as e2:
        abo.update(gp2, X2, np.zeros(300))
    assert e2.value.info == 251
    with pytest.raises(abo.DimensionMismatch):
        abo.update_and_evaluate(abo.UpperConfidenceBound(2.0), gp, X, y, synth.points(2, 10, 3), k=1)


def test_small_batch_topk_single_launch_radix_select():
    """M ≤ 16384, k ≤ 1024 take the one-launch selection (misc.hip: topk_small_kernel: radix select of the k-th key, ties in index
    order, bitonic sort of the ≤ k selected); beyond that the block-sort passes.  Both must reproduce sortperm(scores; rev=true)[1:k]
    (acq_utils.jl:51-52) bit for bit — here on scores with thousands of exact ties (candidates far from the data all score the prior),
    a NaN, and sizes on both sides of every boundary."""
    rng = np.random.default_rng(0)
    m = abo.update(abo.HipStandardGP(abo.SqExponentialKernel(), 0.1), np.array([[0.0], [1.0]]), np.array([0.0, 1.0]))
    acq = abo.UpperConfidenceBound(2.0)
    # above 16384: slices of 16384 selected by one workgroup each, then one merge — two launches while the slices' picks fit one
    # workgroup (16385: a last slice of ONE entry; 32769; 300000 with k = 1024 is back on the block-sort passes)
    for M in (1, 2, 63, 65, 1000, 1025, 4097, 10000, 16383, 16384, 16385, 20000, 32768, 32769, 65536, 70001, 300000):
        Z = np.concatenate([rng.uniform(0, 1, M // 2), np.full(M - M // 2, 50.0) + np.arange(M - M // 2)])
        rng.shuffle(Z)
        if M > 10:
            Z[3] = np.nan
        for k in (1, 2, 7, 100, 128, 129, 1024):
            s, tv, ti = abo.evaluate(acq, m, Z, k=k)
            ov, oi = O.top_k(s, k)
            kk = min(k, M)
            np.testing.assert_array_equal(ti[:kk], oi[:kk])
            np.testing.assert_array_equal(tv[:kk], ov[:kk])
            assert np.all(ti[kk:] == -1) and np.all(np.isnan(tv[kk:]))


def test_small_read_backs_through_pinned_memory_equal_the_direct_copies():
    """The fit's scalars, the selected (score, index) pairs, an append's scalars and a refinement's results land in a pinned block of
    the handle's context and are copied on behind the call's synchronisation; ABO_NO_PINNED=1 (read once per process: a child
    process) takes the direct copies into the caller's arrays instead.  Same bits either way."""
    import hashlib, os, subprocess, sys
    code = (
        "import hashlib, sys, numpy as np\n"
        "sys.path.insert(0, '.')\n"
        "import importlib; abo = importlib.import_module('abstractbayesopt.jl_amd')\n"
        "from abstractbayesopt.jl_amd import synth\n"
        "X, y = synth.standardized_problem(60, 2, 0.05)\n"
        "Z = synth.points(2, 5000, 2)\n"
        "m = abo.update(abo.HipStandardGP(1.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.6), 1e-3, n_max=64), X[:59], y[:59])\n"
        "_, tv, ti = abo.evaluate(abo.ExpectedImprovement(0.01, float(y.min())), m, Z, k=100, return_scores=False)\n"
        "m2 = abo.append(m, X[59], y[59])\n"
        "_, tv2, ti2 = abo.evaluate(abo.UpperConfidenceBound(2.0), m2, Z, k=37, return_scores=False)\n"
        "h = hashlib.sha256()\n"
        "for a in (tv, ti, tv2, ti2, np.array([abo.nlml_fitted(m), abo.nlml_fitted(m2)])): h.update(np.ascontiguousarray(a).tobytes())\n"
        "print(h.hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    # … and the way a call waits for its stream (polled for short waits, blocking beyond ABO_SPIN_WAIT_US; 0 = always block) changes nothing
    for extra in ({}, {"ABO_NO_PINNED": "1"}, {"ABO_SPIN_WAIT_US": "0"}, {"ABO_SPIN_WAIT_US": "1"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, **extra))
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([ln for ln in r.stdout.split() if len(ln) == 64][-1])
    assert out[0] == out[1] == out[2] == out[3]
