"""GPU parity tests (-m gpu) of the incremental path (BASELINE config 5): bordered
append, O(N·M) posterior down-date on a resident candidate set, greedy q-EI.  The reference has no
counterpart (it always refits), so parity is defined against the from-scratch path on the N+j points —
both this library's own full refit and the CPU oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth
from oracle import gp_oracle as O

from tests.parity_record import check
from tests.test_gpu_parity import make_model


@pytest.mark.parametrize("family,d,N0,n_app,ell,noise,mean_c,n_max", [
    (O.SE, 2, 5, 6, 0.7, 1e-4, 0.0, 64),
    (O.MATERN52, 4, 120, 20, 1.0, 1e-3, 0.5, 256),     # crosses the 128 padding boundary
    (O.MATERN52, 16, 500, 13, 2.0, 1e-2, 0.0, 640),    # C5 dimension
    (O.MATERN72, 3, 250, 10, 0.8, 1e-3, 0.0, 0),       # no spare capacity: first append refits with room to grow
])
def test_append_matches_full_refit(family, d, N0, n_app, ell, noise, mean_c, n_max):
    X = synth.points(1, N0 + n_app, d)
    y = synth.objective(X, 0.05) + mean_c
    Z = synth.points(2, 777, d)
    m = abo.update(make_model(family, ell, 1.3, noise, mean_c, n_max=n_max), X[:N0], y[:N0])
    models = [m]
    for j in range(n_app):
        m = abo.append(m, X[N0 + j], y[N0 + j])
        models.append(m)
    ref = abo.update(make_model(family, ell, 1.3, noise, mean_c), X, y)
    L, al, Li = abo.get_factor(m)
    Lr, alr, Lir = abo.get_factor(ref)
    st = O.fit(family, ell, 1.3, noise, mean_c, X, y)
    tol = max(1e-11, 4e-16 * (1 + (N0 + n_app) * 1.3 / noise))
    case = f"append/fam{family}_d{d}_N{N0}+{n_app}"
    check(case, "L", np.max(np.abs(L - st.L)), tol * 2)
    check(case, "L_vs_own_refit", np.max(np.abs(L - Lr)), tol * 2)
    check(case, "LinvL_minus_I", np.max(np.abs(Li @ st.L - np.eye(N0 + n_app))), tol * 20)
    check(case, "alpha_rel", np.max(np.abs(al - st.alpha)) / max(1.0, np.max(np.abs(st.alpha))), min(1e-6, tol * 1e3))
    mu, var = abo.mean_and_var(m, Z)
    mu_o, var_o = O.predict(st, Z)
    check(case, "mu", np.max(np.abs(mu - mu_o)) / max(1.0, np.max(np.abs(mu_o))), min(1e-6, tol * 1e2))
    check(case, "var", np.max(np.abs(var - var_o)) / 1.3, min(1e-6, tol * 1e2))
    check(case, "nlml_rel", abs(abo.nlml_fitted(m) - O.nlml(st)) / max(1.0, abs(O.nlml(st))), min(1e-6, tol * 1e2))
    # every intermediate model is still valid and unchanged (free rollback, bayesian_opt.jl:116-141)
    k = n_app // 2
    st_k = O.fit(family, ell, 1.3, noise, mean_c, X[:N0 + k], y[:N0 + k])
    mu_k, var_k = abo.mean_and_var(models[k], Z[:100])
    mu_ko, var_ko = O.predict(st_k, Z[:100])
    assert np.max(np.abs(mu_k - mu_ko)) <= tol * 1e2 * max(1.0, np.max(np.abs(mu_ko)))
    assert np.max(np.abs(var_k - var_ko)) <= tol * 1e2 * 1.3


def test_append_posdef_failure_leaves_model_intact():
    # test/test_bayesian_opt.jl:749-786 through the incremental path: duplicate point, zero noise
    X = np.array([[-1.0, -1.0], [5.0, -5.0]])
    m = abo.update(make_model(O.SE, 1.0, 1.0, 0.0, n_max=16), X, [1.0, 2.0])
    with pytest.raises(abo.PosDefException) as e:
        abo.append(m, [-1.0 + 1e-12, -1.0 + 1e-12], 1.0)
    assert e.value.info == 3
    ok = abo.append(m, [2.0, 2.0], 0.5)          # the shared storage is still appendable
    mu = abo.posterior_mean(ok, [[2.0, 2.0]])
    assert abs(mu[0] - 0.5) < 1e-9
    assert np.isfinite(abo.posterior_mean(m, [[0.0, 0.0]])[0])
    with pytest.raises(abo.DimensionMismatch):
        abo.append(m, [1.0, 2.0, 3.0], 0.0)


def test_diverging_appends_copy_on_write():
    X = synth.points(1, 40, 3)
    y = synth.objective(X)
    base = abo.update(make_model(O.MATERN52, 0.9, 1.0, 1e-3, n_max=64), X[:38], y[:38])
    a = abo.append(base, X[38], y[38])
    b = abo.append(base, X[39], y[39])            # second branch from the same parent → private storage
    Z = synth.points(2, 50, 3)
    for mdl, idx in ((a, list(range(38)) + [38]), (b, list(range(38)) + [39])):
        st = O.fit(O.MATERN52, 0.9, 1.0, 1e-3, 0.0, X[idx], y[idx])
        mu_o, var_o = O.predict(st, Z)
        mu, var = abo.mean_and_var(mdl, Z)
        assert np.max(np.abs(mu - mu_o)) < 1e-9 and np.max(np.abs(var - var_o)) < 1e-9


def test_resident_candidates_downdate_matches_full_evaluation():
    d, N0 = 8, 300
    X = synth.points(1, N0 + 5, d)
    y = synth.objective(X, 0.02)
    Z = synth.points(2, 5000, d)
    m = abo.update(make_model(O.MATERN52, 1.0, 1.0, 1e-3, n_max=N0 + 64), X[:N0], y[:N0])
    cands = abo.ResidentCandidates(m, Z)
    mu0, var0 = cands.mean_and_var()
    mu_f, var_f = abo.mean_and_var(m, Z)
    np.testing.assert_array_equal(mu0, mu_f)
    np.testing.assert_array_equal(var0, var_f)
    for j in range(5):
        m = abo.append(m, X[N0 + j], y[N0 + j])
        cands.downdate(m)
    mu1, var1 = cands.mean_and_var()
    st = O.fit(O.MATERN52, 1.0, 1.0, 1e-3, 0.0, X, y)
    mu_o, var_o = O.predict(st, Z)
    assert np.max(np.abs(mu1 - mu_o)) < 1e-9
    assert np.max(np.abs(var1 - var_o)) < 1e-9
    # acquisition epilogue on the stored posterior == fused path on the same model
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    s_r, tv_r, ti_r = cands.evaluate(acq, k=10, return_scores=True)
    s_f, tv_f, ti_f = abo.evaluate(acq, m, Z, k=10)
    np.testing.assert_allclose(s_r, s_f, rtol=0, atol=1e-10)
    # a model that is not the one-point append of the synced one is refused
    with pytest.raises(ValueError):
        cands.downdate(abo.update(make_model(O.MATERN52, 1.0, 1.0, 1e-3), X, y))
    cands.refresh(m)
    np.testing.assert_allclose(cands.mean_and_var()[1], var_o, atol=1e-9)


def test_greedy_qei_matches_from_scratch_loop():
    """Each greedy sub-step has exactly the semantics of update() + EI over the grid: replay it with the
    CPU oracle from scratch and compare picks and EI values."""
    d, N0, M, q = 4, 200, 3000, 6
    X, y = synth.standardized_problem(N0, d, 0.05)
    Z = synth.points(2, M, d)
    best = float(y.min())
    fam, ell, sf2, noise = O.MATERN52, 0.6, 1.0, 1e-6     # near-noiseless: σ² collapses at a picked point
    m = abo.update(make_model(fam, ell, sf2, noise, n_max=N0 + 16), X, y)
    cands = abo.ResidentCandidates(m, Z)
    pts, idxs, vals, m_q = abo.greedy_qei(m, cands, q, 0.01, best)
    Xo, yo = X.copy(), y.copy()
    for j in range(q):
        st = O.fit(fam, ell, sf2, noise, 0.0, Xo, yo)
        mu, var = O.predict(st, Z)
        ei = O.expected_improvement(mu, var, best, 0.01)
        v, i = O.top_k(ei, 1)
        assert i[0] == idxs[j], (j, i[0], idxs[j])
        assert abs(v[0] - vals[j]) <= 1e-9 * max(1.0, abs(v[0]))
        np.testing.assert_array_equal(pts[j], Z[i[0]])
        Xo = np.vstack([Xo, Z[i[0]]])
        yo = np.append(yo, mu[i[0]])              # Kriging believer
    L, al, _ = abo.get_factor(m_q)
    assert L.shape == (N0 + q, N0 + q)
    # the batch does not depend on the q-th conditioning: without it the same picks, bit for bit, and q − 1 fantasies in the model
    del m_q                                       # (while the fantasy models live, an append to `m` is a copy-on-write refit)
    cands.refresh(m)
    pts2, idxs2, vals2, m_q2 = abo.greedy_qei(m, cands, q, 0.01, best, condition_last=False)
    np.testing.assert_array_equal(idxs2, idxs)
    np.testing.assert_array_equal(vals2, vals)
    np.testing.assert_array_equal(pts2, pts)
    assert abo.get_factor(m_q2)[0].shape == (N0 + q - 1, N0 + q - 1)


def test_qei_exploration_then_real_append_on_the_parent():
    """C5 step shape: explore q fantasy appends, drop them, roll the candidate posterior back, then append
    the real observation to the (still valid) parent — in place again once the fantasy models are gone,
    with their stale factor rows masked everywhere."""
    d, N0, M = 5, 130, 2000          # 130: the fantasies spill into a second 128-row block
    X, y = synth.standardized_problem(N0, d, 0.05)
    Z = synth.points(2, M, d)
    fam, ell, sf2, noise = O.MATERN52, 0.8, 1.0, 1e-4
    base = abo.update(make_model(fam, ell, sf2, noise, n_max=N0 + 32), X, y)
    cands = abo.ResidentCandidates(base, Z)
    cands.save()
    for rounds in range(3):
        pts, idxs, vals, m_q = abo.greedy_qei(base, cands, 5, 0.01, float(y.min()))
        del m_q
        cands.restore()
        mu_b, var_b = cands.mean_and_var()
        mu_f, var_f = abo.mean_and_var(base, Z)       # parent unaffected by the dead fantasy rows
        np.testing.assert_allclose(mu_b, mu_f, rtol=0, atol=1e-10)   # (down-dated vs re-evaluated: rounding only)
        np.testing.assert_allclose(var_b, var_f, rtol=0, atol=1e-10)
        y_real = float(np.sin(pts[0]).sum())
        new = abo.append(base, pts[0], y_real)
        cands.downdate(new)
        X = np.vstack([X, pts[0]]); y = np.append(y, y_real)
        st = O.fit(fam, ell, sf2, noise, 0.0, X, y)
        mu_o, var_o = O.predict(st, Z)
        mu_c, var_c = cands.mean_and_var()
        assert np.max(np.abs(mu_c - mu_o)) < 1e-8 and np.max(np.abs(var_c - var_o)) < 1e-8
        mu_n, var_n = abo.mean_and_var(new, Z)
        assert np.max(np.abs(mu_n - mu_o)) < 1e-8 and np.max(np.abs(var_n - var_o)) < 1e-8
        L, al, Li = abo.get_factor(new)
        assert np.max(np.abs(L - st.L)) < 1e-9
        base = new
        cands.save()


def test_full_size_parity_c5():
    """BASELINE config 5 at full size (d = 16, N = 16384 + 3 appended, noisy): the incremental path — three bordered
    appends and three O(N·M) down-dates of a resident grid — against an INDEPENDENT oracle refit on the N + 3 points
    (O.fit: host LAPACK, about a minute) and the oracle posterior on 1024 candidates; the library's own from-scratch
    refit is compared as well (it shares no code path with the append beyond the kernel evaluation)."""
    d, N, M = 16, 16384, 4096
    ell, sf2, noise = 2.0, 1.0, 1e-2
    X = synth.points(1, N + 3, d)
    y = synth.objective(X, 0.1)
    y = (y - y.mean()) / y.std(ddof=1)
    Z = synth.points(2, M, d)
    gp = make_model(O.MATERN52, ell, sf2, noise, n_max=N + 64)
    m = abo.update(gp, X[:N], y[:N])
    cands = abo.ResidentCandidates(m, Z)
    for j in range(3):
        m = abo.append(m, X[N + j], y[N + j])
        cands.downdate(m)
    mu_a, var_a = abo.mean_and_var(m, Z)
    # the engine is PINNED: AUTO would fall back to the fp64 kernels silently if the int8 scratch did not fit the device
    t = m.timings()
    assert t["contraction_engine"] == abo._lib.CONTRACT_INT8 and t["oz_nmod"] == 14, t
    mu_c, var_c = cands.mean_and_var()
    nl_a = abo.nlml_fitted(m)
    L, alpha, _ = abo.get_factor(m)
    # the fp64 engine on the same appended model (abo_set_contraction on the fitted handle: same factor, other contraction)
    abo._lib.check(abo._lib.lib().abo_set_contraction(m._require(), abo._lib.CONTRACT_FP64, 0))
    mu_64, var_64 = abo.mean_and_var(m, Z[:1024])
    assert m.timings()["contraction_engine"] == abo._lib.CONTRACT_FP64
    np.testing.assert_array_equal(mu_64, mu_a[:1024])
    del cands, m
    ref = abo.update(make_model(O.MATERN52, ell, sf2, noise), X, y)
    mu_r, var_r = abo.mean_and_var(ref, Z)
    nl_r = abo.nlml_fitted(ref)
    del ref
    case = "c5/N16384+3_d16"
    check(case, "mu_append_vs_own_refit", np.max(np.abs(mu_a - mu_r)), 1e-8)
    check(case, "var_append_vs_own_refit", np.max(np.abs(var_a - var_r)), 1e-8)
    check(case, "mu_downdated_grid_vs_own_refit", np.max(np.abs(mu_c - mu_r)), 1e-8)
    check(case, "var_downdated_grid_vs_own_refit", np.max(np.abs(var_c - var_r)), 1e-8)
    check(case, "nlml_rel_append_vs_own_refit", abs(nl_a - nl_r) / abs(nl_r), 1e-9)
    assert np.all(var_r > 0) and np.all(var_r < 1.0 + 1e-12)
    st = O.fit(O.MATERN52, ell, sf2, noise, 0.0, X, y)                     # independent: nothing below comes from the device
    check(case, "L", np.max(np.abs(L - st.L)) / np.sqrt(sf2 + noise), 1e-9)
    check(case, "alpha_rel", np.max(np.abs(alpha - st.alpha)) / max(1.0, np.max(np.abs(st.alpha))), 1e-6)
    check(case, "nlml_rel", abs(nl_a - O.nlml(st)) / abs(O.nlml(st)), 1e-9)
    mu_o, var_o = O.predict(st, Z[:1024])
    check(case, "mu", np.max(np.abs(mu_a[:1024] - mu_o)) / max(1.0, np.max(np.abs(mu_o))), 1e-8)
    check(case, "var", np.max(np.abs(var_a[:1024] - var_o)) / sf2, 1e-8)
    check(case, "var_fp64_engine", np.max(np.abs(var_64 - var_o)) / sf2, 1e-8)
    check(case, "var_between_engines", np.max(np.abs(var_64 - var_a[:1024])) / sf2, 1e-8)
    check(case, "mu_downdated_grid", np.max(np.abs(mu_c[:1024] - mu_o)) / max(1.0, np.max(np.abs(mu_o))), 1e-8)
    check(case, "var_downdated_grid", np.max(np.abs(var_c[:1024] - var_o)) / sf2, 1e-8)


def test_full_size_block_qei_c5_against_the_from_scratch_path_and_the_oracle():
    """BASELINE config 5's q-EI hot path AT ITS OWN SIZE (d = 16, N = 16384, a resident grid of 131 072 candidates, q = 8, T = 16:
    the resident K_ZX has 131 072 × 16 448 = 2.16e9 elements, beyond 2³¹) anchored on something other than itself:
      (a) the batch replayed with the library's FROM-SCRATCH path — per pick a full refit on the N + j points (the fantasy value is
          the refitted model's mean: Kriging believer) and one abo_acq(k = 1) over the whole grid, the path test_full_size_parity_c3 /
          _c5 verify against the oracle at this size; it shares neither K_ZX, nor the pass kernel, nor the chain with the block form:
          same picks, EI to 1e-8 relative;
      (b) the INDEPENDENT oracle (O.fit: host LAPACK) on the base model: EI of 3 × 341 grid rows — the first, the middle and the LAST
          rows of K_ZX — and of the 16 best candidates, whose oracle ordering must put pick 1 first;
      (c) after the first pick is appended for real — its down-date column served from the batch's chain — the down-dated (μ, σ²) of
          the same rows against an oracle refit on the N + 1 points.
    Reference semantics per sub-step: src/surrogates/StandardGP.jl:79-83 (update), src/acquisition_functions/ExpectedImprovement.jl:40-66."""
    import torch
    d, N, M, q = 16, 16384, 131072, 8
    ell, sf2, noise, xi = 2.0, 1.0, 1e-2, 0.01
    X = synth.points(1, N, d)
    y = synth.objective(X, 0.1)
    y = (y - y.mean()) / y.std(ddof=1)
    Z = synth.points(2, M, d)
    best = float(y.min())
    acq = abo.ExpectedImprovement(xi, best)
    m = abo.update(make_model(O.MATERN52, ell, sf2, noise, n_max=N + 64), X, y)
    cands = abo.ResidentCandidates(m, Z)
    ei0, tv16, ti16 = cands.evaluate(acq, k=16, return_scores=True)
    mu0, var0 = cands.mean_and_var()
    pts, idx, val, st = cands.qei(q, xi, best, block=16)
    assert st["block"] == 16 and st["block_builds"] >= 1 and st["block_builds"] + st["block_hits"] == q - 1, st
    assert st["pass_bytes"] == 8.0 * N * M                    # the block came from one pass over the resident K_ZX
    np.testing.assert_array_equal(pts, Z[idx])
    assert idx[0] == ti16[0] and val[0] == tv16[0]
    rows = np.concatenate([np.arange(341), np.arange(M // 2 - 170, M // 2 + 171), np.arange(M - 341, M)])
    case = "qei_full/N16384_d16_M131072_q8_T16"
    # (c) first: the real append of pick 1, column from the chain
    y_real = 0.25
    m1 = abo.append(m, pts[0], y_real)
    cands.downdate(m1)
    assert m1.timings()["downdate_from_chain"] == 1
    mu1, var1 = cands.mean_and_var()
    del cands, m1
    abo._lib.lib().abo_pool_trim(0)
    # (a) the from-scratch replay on the device
    Zd = torch.from_numpy(Z).cuda()
    Xj, yj = X.copy(), y.copy()
    idx_s, val_s = [], []
    for j in range(q):
        mj = abo.update(make_model(O.MATERN52, ell, sf2, noise), Xj, yj)
        _, tv, ti = abo.evaluate(acq, mj, Zd, k=1, return_scores=False)
        i = int(ti.cpu().numpy()[0])
        idx_s.append(i); val_s.append(float(tv.cpu().numpy()[0]))
        Xj = np.vstack([Xj, Z[i]])
        yj = np.append(yj, abo.posterior_mean(mj, Z[i][None, :])[0])     # Kriging believer
        del mj
    np.testing.assert_array_equal(idx, np.array(idx_s))
    val_s = np.array(val_s)
    check(case, "ei_vs_from_scratch_refits_rel", np.max(np.abs(val - val_s) / np.maximum(1e-3 * val_s[0], np.abs(val_s))), 1e-8)
    del Zd
    # (b) the independent oracle on the base model
    st0 = O.fit(O.MATERN52, ell, sf2, noise, 0.0, X, y)
    sel = np.concatenate([rows, ti16])
    mu_o, var_o = O.predict(st0, Z[sel])
    ei_o = O.expected_improvement(mu_o, var_o, best, xi)
    # (EI underflows on most of a random grid: the rows are held by their posterior, the EI comparison proper is the top-16's)
    check(case, "mu_rows_first_middle_last", np.max(np.abs(mu0[rows] - mu_o[:len(rows)])) / max(1.0, np.max(np.abs(mu_o))), 1e-8)
    check(case, "var_rows_first_middle_last", np.max(np.abs(var0[rows] - var_o[:len(rows)])) / sf2, 1e-8)
    check(case, "ei_rows_first_middle_last_abs", np.max(np.abs(ei0[rows] - ei_o[:len(rows)])), 1e-9)
    check(case, "ei_top16_rel", np.max(np.abs(tv16 - ei_o[len(rows):]) / np.abs(ei_o[len(rows):])), 1e-8)
    assert int(np.argmax(ei_o[len(rows):])) == 0              # the oracle ranks pick 1 first among the block's candidates
    assert abs(val[0] - ei_o[len(rows)]) <= 1e-8 * abs(ei_o[len(rows)])
    del st0
    # (c) the oracle's refit on the N + 1 points against the chain-served down-date
    st1 = O.fit(O.MATERN52, ell, sf2, noise, 0.0, np.vstack([X, pts[0]]), np.append(y, y_real))
    mu_o1, var_o1 = O.predict(st1, Z[rows])
    check(case, "mu_downdated_from_chain", np.max(np.abs(mu1[rows] - mu_o1)) / max(1.0, np.max(np.abs(mu_o1))), 1e-8)
    check(case, "var_downdated_from_chain", np.max(np.abs(var1[rows] - var_o1)) / sf2, 1e-8)


def test_resident_kzx_and_recomputed_downdates_agree(monkeypatch):
    """The down-date streams a resident K_ZX when it fits the budget (ABO_CAND_KZX_GIB) and re-evaluates the kernel
    otherwise: same posterior either way, including after a rollback that re-uses the appended columns and with
    appends that cross a 128-row block boundary of the factor."""
    d, N0 = 6, 250
    X = synth.points(1, N0 + 12, d)
    y = synth.objective(X, 0.02)
    Z = synth.points(2, 3001, d)
    out = {}
    for mode in ("64", "0"):
        monkeypatch.setenv("ABO_CAND_KZX_GIB", mode)
        m0 = abo.update(make_model(O.MATERN52, 0.8, 1.3, 1e-3, n_max=N0 + 64), X[:N0], y[:N0])
        cands = abo.ResidentCandidates(m0, Z)
        cands.save()
        m = m0
        for j in range(3):                                   # fantasy branch, then rolled back
            m = abo.append(m, Z[10 + j], 0.1 * j)
            cands.downdate(m)
        del m
        cands.restore()
        m = m0
        for j in range(12):                                  # crosses row 256
            m = abo.append(m, X[N0 + j], y[N0 + j])
            cands.downdate(m)
        out[mode] = cands.mean_and_var()
    np.testing.assert_allclose(out["64"][0], out["0"][0], rtol=0, atol=1e-11)
    np.testing.assert_allclose(out["64"][1], out["0"][1], rtol=0, atol=1e-11)
    st = O.fit(O.MATERN52, 0.8, 1.3, 1e-3, 0.0, X, y)
    mu_o, var_o = O.predict(st, Z)
    assert np.max(np.abs(out["64"][0] - mu_o)) < 1e-9 and np.max(np.abs(out["64"][1] - var_o)) < 1e-9


def test_greedy_qei_distinct_batch_under_noise():
    """With observation noise the Kriging-believer rule may return a candidate twice; distinct=True excludes picked
    candidates (their EI becomes 0) and the saved posterior comes back untouched on restore."""
    d, N = 3, 120
    X = synth.points(1, N, d)
    y = synth.objective(X, 0.1)
    Z = synth.points(2, 4000, d)
    m = abo.update(make_model(O.MATERN52, 0.6, 1.0, 5e-2, n_max=N + 32), X, y)
    cands = abo.ResidentCandidates(m, Z)
    mu0, var0 = cands.mean_and_var()
    cands.save()
    pts, idxs, vals, _ = abo.greedy_qei(m, cands, 6, 0.0, float(y.min()), distinct=True)
    assert len(set(idxs.tolist())) == 6
    np.testing.assert_array_equal(pts, Z[idxs])
    assert np.all(np.diff(vals) <= 1e-12) or np.all(vals >= 0)
    s, _, _ = cands.evaluate(abo.ExpectedImprovement(0.0, float(y.min())), k=1, return_scores=True)
    assert np.all(s[idxs] == 0.0)
    cands.restore()
    mu1, var1 = cands.mean_and_var()
    np.testing.assert_array_equal(mu1, mu0)
    np.testing.assert_array_equal(var1, var0)


def _oracle_greedy(fam, ell, sf2, noise, X, y, Z, q, xi, best):
    """the batch from scratch: refit on the N + j points, EI over the grid, arg-max, Kriging-believer value"""
    Xo, yo, idx, val = X.copy(), y.copy(), [], []
    for j in range(q):
        st = O.fit(fam, ell, sf2, noise, 0.0, Xo, yo)
        mu, var = O.predict(st, Z)
        v, i = O.top_k(O.expected_improvement(mu, var, best, xi), 1)
        idx.append(int(i[0])); val.append(float(v[0]))
        Xo = np.vstack([Xo, Z[i[0]]]); yo = np.append(yo, mu[i[0]])
    return np.array(idx), np.array(val)


@pytest.mark.parametrize("fam,d,N0,M,q,ell,noise,block", [
    (O.MATERN52, 4, 200, 3000, 6, 0.6, 1e-6, 0),       # default block (32), near-noiseless
    (O.MATERN52, 16, 500, 5000, 8, 2.0, 1e-2, 16),     # the C5 shape: noisy, long length scale
    (O.SE, 2, 130, 2500, 12, 0.25, 1e-4, 16),          # short length scale: the picks leave the first block (new blocks are built)
    (O.MATERN72, 3, 40, 37, 5, 0.7, 1e-3, 64),         # fewer candidates than a block holds
])
def test_block_qei_equals_the_plain_loop_and_the_from_scratch_batch(fam, d, N0, M, q, ell, noise, block):
    """The block form of greedy q-EI (covariance columns of T points from one pass over K_ZX, rank-1 corrections between picks, no
    fantasy appends) against (a) the plain loop (bordered append + one pass per pick) and (b) the oracle's from-scratch batch: same
    picks, EI values to rounding.  The one-call C driver (abo_cand_qei) and the Python driver over the step calls agree bit for bit."""
    X, y = synth.standardized_problem(N0, d, 0.05)
    Z = synth.points(2, M, d)
    best, xi = float(y.min()), 0.01
    m = abo.update(make_model(fam, ell, 1.0, noise, n_max=N0 + 64), X, y)
    cands = abo.ResidentCandidates(m, Z)
    mu0, var0 = cands.mean_and_var()
    pts_c, idx_c, val_c, st = cands.qei(q, xi, best, block=block)
    assert st["block"] == (32 if block == 0 else block) and st["picks"] == q      # 0: the library's default
    assert st["block_builds"] >= 1 and st["block_builds"] + st["block_hits"] == q - 1
    if fam == O.SE:
        assert st["block_builds"] > 1, st                      # this case exists for the rebuild path
    np.testing.assert_array_equal(cands.mean_and_var()[0], mu0)   # rolled back
    np.testing.assert_array_equal(cands.mean_and_var()[1], var0)
    # the same batch again on the same model: the blocks are still there — no pass over K_ZX — and give the same bits
    pts_r, idx_r, val_r, st_r = cands.qei(q, xi, best, block=block)
    assert st_r["block_builds"] == 0 and st_r["block_hits"] == q - 1, st_r
    np.testing.assert_array_equal(idx_r, idx_c)
    np.testing.assert_array_equal(val_r, val_c)
    cands.refresh(m)                                           # (a refresh starts the set's q-EI state afresh)
    # the Python driver over the step calls (what a host that shards the set itself runs: incremental.py over torch.distributed)
    from abstractbayesopt.jl_amd.incremental import _qei_block_batch
    pts_p, idx_p, val_p, _, stats = _qei_block_batch(m, cands, q, q - 1, xi, best, 0, None, False, block)
    assert stats["block_builds"] == st["block_builds"]
    np.testing.assert_array_equal(idx_p, idx_c)
    np.testing.assert_array_equal(val_p, val_c)
    np.testing.assert_array_equal(pts_p, pts_c)
    pts_l, idx_l, val_l, _ = cands.qei(q, xi, best, block=-1)[:4]   # the plain loop, same call
    np.testing.assert_array_equal(idx_l, idx_c)
    np.testing.assert_array_equal(pts_l, Z[idx_c])
    case = f"qei_block/fam{fam}_d{d}_N{N0}_M{M}_q{q}_T{block}"
    check(case, "ei_vs_plain_loop", np.max(np.abs(val_l - val_c) / np.maximum(1e-3 * val_l[0], np.abs(val_l))), 1e-9)
    idx_o, val_o = _oracle_greedy(fam, ell, 1.0, noise, X, y, Z, q, xi, best)
    np.testing.assert_array_equal(idx_c, idx_o)
    check(case, "ei_vs_from_scratch", np.max(np.abs(val_o - val_c) / np.maximum(1e-3 * val_o[0], np.abs(val_o))), 1e-8)
    np.testing.assert_array_equal(cands.mean_and_var()[1], var0)


def test_real_appends_after_a_block_batch_take_their_columns_from_the_chain(monkeypatch):
    """After a block-form batch the picks are appended for real, in order: every down-date finds its column c_i(z) in the chain the
    batch left with the set (abo_timings.downdate_from_chain) instead of streaming K_ZX — same posterior as the streaming pass to
    rounding, and as the oracle's refit; the appended K_ZX columns are written either way (a later streaming pass is right)."""
    d, N0, M, q = 6, 300, 4000, 5
    X, y = synth.standardized_problem(N0, d, 0.05)
    Z = synth.points(2, M, d)
    fam, ell, noise, best = O.MATERN52, 0.9, 1e-3, float(y.min())
    out = {}
    for mode in ("chain", "stream"):
        if mode == "stream":
            monkeypatch.setenv("ABO_QEI_NO_CHAIN", "1")
        m = abo.update(make_model(fam, ell, 1.0, noise, n_max=N0 + 64), X, y)
        cands = abo.ResidentCandidates(m, Z)
        pts, idx, val, _ = cands.qei(q, 0.01, best)
        Xo, yo, flags = X.copy(), y.copy(), []
        for j in range(q):                                   # q − 1 chain columns; the q-th pick was never conditioned on
            y_real = float(np.sin(3.0 * pts[j]).sum())
            m = abo.append(m, pts[j], y_real)
            cands.downdate(m)
            flags.append(m.timings()["downdate_from_chain"])
            Xo = np.vstack([Xo, pts[j]]); yo = np.append(yo, y_real)
        assert flags == ([1] * (q - 1) + [0] if mode == "chain" else [0] * q), flags
        m = abo.append(m, X[0] + 0.01, 0.3)                  # not a pick: the streaming pass over K_ZX incl. the q new columns
        cands.downdate(m)
        Xo = np.vstack([Xo, X[0] + 0.01]); yo = np.append(yo, 0.3)
        out[mode] = cands.mean_and_var()
    st = O.fit(fam, ell, 1.0, noise, 0.0, Xo, yo)
    mu_o, var_o = O.predict(st, Z)
    case = f"qei_chain/d{d}_N{N0}_M{M}_q{q}"
    check(case, "mu_chain_vs_stream", np.max(np.abs(out["chain"][0] - out["stream"][0])), 1e-10)
    check(case, "var_chain_vs_stream", np.max(np.abs(out["chain"][1] - out["stream"][1])), 1e-10)
    check(case, "mu", np.max(np.abs(out["chain"][0] - mu_o)) / max(1.0, np.max(np.abs(mu_o))), 1e-8)
    check(case, "var", np.max(np.abs(out["chain"][1] - var_o)), 1e-8)


def test_block_batch_refuses_what_it_cannot_do_and_falls_back():
    d, N0 = 3, 60
    X, y = synth.standardized_problem(N0, d, 0.05)
    Z = synth.points(2, 500, d)
    m = abo.update(make_model(O.MATERN52, 0.7, 1.0, 1e-3, n_max=N0 + 128), X, y)
    cands = abo.ResidentCandidates(m, Z)
    L = abo._lib.lib()
    with pytest.raises(ValueError):                          # no batch open
        abo._lib.check(L.abo_cand_qei_pick(m._require(), cands._h.ptr, 0, 1.0, None, 0, -1, None))
    m2 = abo.append(m, Z[0], 0.1)                            # the set is not in sync with m2
    with pytest.raises(ValueError):
        abo._lib.check(L.abo_cand_qei_begin(m2._require(), cands._h.ptr, 4, 0))
    del m2
    pts, idx, val, st = cands.qei(66, 0.01, float(y.min()))  # q > 64: the plain loop
    assert st["block"] == 0 and len(idx) == 66 and np.all(idx >= 0)


@pytest.mark.parametrize("block", [16, 32, 48, 64])
def test_block_pass_kernel_repeats_the_split_k_kernels_bits(monkeypatch, block):
    """The one pass over the resident K_ZX runs on a kernel of its own (gemm.hip: qei_pass_kernel — A through LDS, B fragments of
    the next chunk in flight under the MFMAs); it keeps the lane ↔ k map and the k order of the skinny split-k kernel, so the
    covariance columns — and with them every EI value of the batch and the chain's down-dated posterior — are the same bits."""
    d, N0, M, q = 5, 700, 3001, 6
    X, y = synth.standardized_problem(N0, d, 0.05)
    Z = synth.points(2, M, d)
    out = {}
    for mode in ("pass", "skinny"):
        if mode == "skinny":
            monkeypatch.setenv("ABO_QEI_PASS_SKINNY", "1")
        m = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-3, n_max=N0 + 64), X, y)
        cands = abo.ResidentCandidates(m, Z)
        pts, idx, val, st = cands.qei(q, 0.01, float(y.min()), block=block)
        assert st["block"] == block
        m2 = abo.append(m, pts[0], 0.25)
        cands.downdate(m2)
        assert m2.timings()["downdate_from_chain"] == 1
        out[mode] = (idx, val) + cands.mean_and_var()
    for a, b in zip(out["pass"], out["skinny"]):
        np.testing.assert_array_equal(a, b)


def test_blocks_and_chain_follow_the_model_through_bo_steps(monkeypatch):
    """A BO loop: batch → the first pick appended for real (its column from the chain) → next batch on the appended model → … .  The
    set's blocks and chain follow the model: later batches find most of their picks' columns in the earlier blocks (no pass over
    K_ZX at all in such a step), corrected by the chain entries made since.  A real append that is NOT a pick (its column comes
    from the streaming pass) joins the chain too.  Every step's picks equal the plain loop's (one pass per pick) and the oracle's
    from-scratch batch; the grid's posterior after the loop equals the oracle's refit."""
    d, N0, M, q, steps = 5, 400, 6000, 5, 6
    X, y = synth.standardized_problem(N0, d, 0.05)
    Z = synth.points(2, M, d)
    fam, ell, noise, xi = O.MATERN52, 0.9, 1e-2, 0.01
    runs = {}
    for mode in ("reuse", "fresh", "plain"):
        if mode == "fresh":
            monkeypatch.setenv("ABO_QEI_NO_REUSE", "1")
        m = abo.update(make_model(fam, ell, 1.0, noise, n_max=N0 + 64), X, y)
        cands = abo.ResidentCandidates(m, Z)
        Xo, yo, best, log, builds, chain = X.copy(), y.copy(), float(y.min()), [], [], []
        for step in range(steps):
            pts, idx, val, st = cands.qei(q, xi, best, block=(-1 if mode == "plain" else 16))
            if mode == "reuse" and step < 3:
                idx_o, val_o = _oracle_greedy(fam, ell, 1.0, noise, Xo, yo, Z, q, xi, best)
                np.testing.assert_array_equal(idx, idx_o)
                assert np.max(np.abs(val - val_o) / np.maximum(1e-3 * val_o[0], np.abs(val_o))) <= 1e-8
            log.append((idx.copy(), val.copy()))
            builds.append(st["block_builds"])
            x_new = pts[0] if step != 3 else Z[17] + 1e-3          # step 3: a real observation that is not a pick
            y_new = float(np.sin(3.0 * x_new).sum() - 0.5)
            m = abo.append(m, x_new, y_new)
            cands.downdate(m)
            chain.append(m.timings()["downdate_from_chain"])
            Xo = np.vstack([Xo, x_new]); yo = np.append(yo, y_new)
            best = min(best, y_new)
        runs[mode] = (log, builds, chain, cands.mean_and_var())
    for mode in ("fresh", "plain"):
        for (ia, va), (ib, vb) in zip(runs["reuse"][0], runs[mode][0]):
            np.testing.assert_array_equal(ia, ib)
            assert np.max(np.abs(va - vb) / np.maximum(1e-3 * vb[0], np.abs(vb))) <= 1e-8, mode
    assert runs["reuse"][2] == [1, 1, 1, 0, 1, 1] and runs["plain"][2] == [0] * steps
    assert runs["fresh"][1][0] >= 1 and all(b >= 1 for b in runs["fresh"][1])      # without reuse every batch builds its block
    assert sum(runs["reuse"][1]) < sum(runs["fresh"][1]), (runs["reuse"][1], runs["fresh"][1])   # … with it, later batches find theirs
    st = O.fit(fam, ell, 1.0, noise, 0.0, Xo, yo)
    mu_o, var_o = O.predict(st, Z)
    case = f"qei_steps/d{d}_N{N0}_M{M}_q{q}x{steps}"
    check(case, "mu_after_loop", np.max(np.abs(runs["reuse"][3][0] - mu_o)) / max(1.0, np.max(np.abs(mu_o))), 1e-8)
    check(case, "var_after_loop", np.max(np.abs(runs["reuse"][3][1] - var_o)), 1e-8)
    check(case, "var_reuse_vs_plain", np.max(np.abs(runs["reuse"][3][1] - runs["plain"][3][1])), 1e-10)
