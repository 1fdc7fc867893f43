"""GPU parity of the int8-residue contraction engine (csrc/ozaki.hip, ABO_CONTRACT_INT8) — the N²·M product behind
posterior_var (reference: src/surrogates/StandardGP.jl:377-379) computed exactly on fixed-point images of its operands.
Checked against the CPU oracle on the same seeded inputs AND against the fp64 MFMA engine of the same library; errors are
scaled as everywhere else (|Δσ²| / σ_f²), recorded by tests/parity_record.py and held to the same kind of bound: a hard bar
(≤ the north star's 1e-6) and 100 × the error recorded on an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth
from oracle import gp_oracle as O

from tests.parity_record import check
from tests.test_gpu_parity import make_model


def _fit_pair(family, ell, sf2, noise, X, y, **kw):
    a = abo.update(make_model(family, ell, sf2, noise, contraction="fp64", **kw), X, y)
    b = abo.update(make_model(family, ell, sf2, noise, contraction="int8", **kw), X, y)
    return a, b


CASES = [
    # family, d, N, M, ell, sf2, noise: sizes around the 256-tile borders of the residue GEMM, forced onto the int8 engine
    (O.SE, 2, 40, 100, 0.7, 1.0, 1e-6),          # one partial tile each way
    (O.MATERN52, 3, 256, 256, 0.8, 1.0, 1e-4),   # exactly one tile
    (O.MATERN52, 4, 300, 700, 1.0, 2.5, 1e-3),   # Np = 384 pads to 512; M = 700 pads to 768
    (O.SE, 4, 1024, 3000, 0.5, 1.0, 1e-4),       # C2 training shape (cond ≈ 1e7)
    (O.MATERN72, 8, 1500, 1025, 1.2, 0.5, 1e-3),
    (O.MATERN52, 8, 2304, 2049, 1.0, 1.0, 1e-3), # nine row blocks: the size the AUTO engine switches at and beyond
    (O.MATERN32, 6, 2100, 515, 0.9, 1.0, 1e-2),
]


@pytest.mark.parametrize("family,d,N,M,ell,sf2,noise", CASES)
def test_int8_engine_against_oracle_and_fp64_engine(family, d, N, M, ell, sf2, noise):
    X = synth.points(1, N, d)
    Z = synth.points(2, M, d) * 1.2 - 0.1
    y = synth.objective(X, 0.05)
    st = O.fit(family, ell, sf2, noise, 0.0, X, y)
    mu_o, var_o = O.predict(st, Z)
    m64, m8 = _fit_pair(family, ell, sf2, noise, X, y)
    mu64, var64 = abo.mean_and_var(m64, Z)
    mu8, var8 = abo.mean_and_var(m8, Z)
    assert m64.timings()["contraction_engine"] == abo._lib.CONTRACT_FP64
    t8 = m8.timings()
    assert t8["contraction_engine"] == abo._lib.CONTRACT_INT8 and t8["oz_nmod"] == 14
    np.testing.assert_array_equal(mu8, mu64)          # the mean never goes through the contraction
    cond = 1.0 + N * sf2 / noise
    tol = min(1e-6, max(1e-11, 4e-16 * cond) * 1e2)
    case = f"int8/fam{family}_d{d}_N{N}"
    e8 = np.max(np.abs(var8 - var_o)) / sf2
    e64 = np.max(np.abs(var64 - var_o)) / sf2
    check(case, "var", e8, tol)
    check(case, "var_fp64_engine", e64, tol)
    check(case, "var_between_engines", np.max(np.abs(var8 - var64)) / sf2, tol)
    # scores and the selection agree with the oracle's on the int8 engine's posterior
    best = float(np.min(y))
    acq = abo.ExpectedImprovement(0.01, best)
    s, tv, ti = abo.evaluate(acq, m8, Z, k=min(50, M))
    np.testing.assert_allclose(s, O.acquisition(O.ACQ_EI, mu8, var8, 0.01, best), rtol=1e-10, atol=1e-14)
    ov, oi = O.top_k(s, min(50, M))
    np.testing.assert_array_equal(ti, oi)


def test_moduli_count_trades_accuracy():
    """each modulus less costs about 3.85 bits per operand: the error against the oracle grows monotonically from 14 moduli
    down, and 14 sits at the fp64 engine's own level"""
    N, d, M = 1024, 4, 2048
    X = synth.points(1, N, d)
    Z = synth.points(2, M, d)
    y = synth.objective(X, 0.05)
    st = O.fit(O.MATERN52, 0.8, 1.0, 1e-4, 0.0, X, y)
    _, var_o = O.predict(st, Z)
    errs = {}
    for n in (10, 11, 12, 13, 14, 16):
        m = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-4, contraction=f"int8:{n}"), X, y)
        var = abo.posterior_var(m, Z)
        assert m.timings()["oz_nmod"] == n
        errs[n] = float(np.max(np.abs(var - var_o)))
        check("int8/moduli_sweep", f"var_n{n}", errs[n], 1e-6 if n >= 11 else 1e-4)
    assert errs[10] > errs[12] > errs[14] * 0.5
    m64 = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-4, contraction="fp64"), X, y)
    e64 = float(np.max(np.abs(abo.posterior_var(m64, Z) - var_o)))
    assert errs[14] <= max(10 * e64, 1e-13) and errs[16] <= max(10 * e64, 1e-13)


def test_int8_engine_is_chunk_independent_and_deterministic():
    """exact integer products: the result cannot depend on how the candidates are cut into chunks, nor on the run"""
    N, d, M = 700, 5, 3000
    X = synth.points(1, N, d)
    Z = synth.points(2, M, d)
    y = synth.objective(X, 0.05)
    ref = None
    for chunk in (0, 256, 1024, 1152):
        m = abo.update(make_model(O.MATERN52, 0.9, 1.0, 1e-3, contraction="int8", chunk=chunk), X, y)
        var = abo.posterior_var(m, Z)
        var2 = abo.posterior_var(m, Z)
        np.testing.assert_array_equal(var, var2)
        if ref is None:
            ref = var
        np.testing.assert_array_equal(var, ref)


def test_non_finite_candidates_stay_nan_on_the_int8_engine():
    X = synth.points(1, 300, 3)
    y = synth.objective(X, 0.05)
    Z = synth.points(2, 600, 3)
    Z[17, 1] = np.nan
    Z[400, 0] = np.inf
    m64, m8 = _fit_pair(O.SE, 0.7, 1.0, 1e-4, X, y)
    mu8, var8 = abo.mean_and_var(m8, Z)
    mu64, var64 = abo.mean_and_var(m64, Z)
    assert np.isnan(var8[17]) and np.isnan(var64[17])
    assert np.isnan(var8[400]) == np.isnan(var64[400])
    ok = np.ones(600, bool)
    ok[[17, 400]] = False
    np.testing.assert_allclose(var8[ok], var64[ok], rtol=0, atol=1e-12)
    s, tv, ti = abo.evaluate(abo.UpperConfidenceBound(2.0), m8, Z, k=3)
    assert ti[0] == 17 and np.isnan(tv[0])            # NaN sorts first, as on the fp64 engine


def test_appended_model_and_resident_grid_on_the_int8_engine():
    """a bordered append adds a row to W: the residue planes are rebuilt for the larger view, and the refreshed grid of a
    candidate set goes through the same engine"""
    N, d, M = 600, 4, 2000
    X = synth.points(1, N + 2, d)
    y = synth.objective(X, 0.05)
    Z = synth.points(2, M, d)
    base = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-3, contraction="int8", n_max=N + 8), X[:N], y[:N])
    v0 = abo.posterior_var(base, Z)
    m1 = abo.append(base, X[N], y[N])
    m2 = abo.append(m1, X[N + 1], y[N + 1])
    v2 = abo.posterior_var(m2, Z)
    assert m2.timings()["contraction_engine"] == abo._lib.CONTRACT_INT8
    st = O.fit(O.MATERN52, 0.8, 1.0, 1e-3, 0.0, X, y)
    _, var_o = O.predict(st, Z)
    check("int8/append_N600", "var", np.max(np.abs(v2 - var_o)), 1e-9)
    np.testing.assert_array_equal(abo.posterior_var(base, Z), v0)        # the older view still answers for its own N
    cands = abo.ResidentCandidates(m2, Z)
    mu_c, var_c = cands.mean_and_var()
    np.testing.assert_array_equal(var_c, v2)


def test_auto_engine_switches_on_size():
    X = synth.points(1, 2100, 4)
    y = synth.objective(X, 0.05)
    Z = synth.points(2, 512, 4)
    big = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-3), X, y)
    abo.posterior_var(big, Z)
    assert big.timings()["contraction_engine"] == abo._lib.CONTRACT_INT8
    small = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-3), X[:1000], y[:1000])
    abo.posterior_var(small, Z)
    assert small.timings()["contraction_engine"] == abo._lib.CONTRACT_FP64
    with pytest.raises(ValueError):
        make_model(O.SE, 1.0, 1.0, 1e-3, contraction="fp32")


def test_sharded_handle_on_the_int8_engine_equals_the_single_handle():
    """the multi-device handle creates its shards inside the library: they take the process default engine, and — exact integer
    products — scores and selection equal the single handle's bit for bit however the candidates are cut"""
    from tests.test_gpu_multigpu import sharded
    d, N, M = 4, 500, 3001
    X, y = synth.standardized_problem(N, d)
    Z = synth.points(2, M, d)
    abo.set_default_contraction("int8")
    try:
        one = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-3), X, y)
        grp = abo.update(sharded(O.MATERN52, 0.8, 1.0, 1e-3, (0, 0, 0)), X, y)
        acq = abo.ExpectedImprovement(0.01, float(y.min()))
        s1, tv1, ti1 = abo.evaluate(acq, one, Z, k=64)
        s2, tv2, ti2 = abo.evaluate(acq, grp, Z, k=64)
        assert one.timings()["contraction_engine"] == abo._lib.CONTRACT_INT8
    finally:
        abo.set_default_contraction("auto")
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(tv1, tv2)
    np.testing.assert_array_equal(ti1, ti2)
    ref = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-3, contraction="fp64"), X, y)
    s0, _, ti0 = abo.evaluate(acq, ref, Z, k=64)
    np.testing.assert_allclose(s1, s0, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(ti1, ti0)


@pytest.mark.parametrize("sf2,noise,ell", [(37.5, 1e-2, 0.8), (3e-4, 1e-9, 0.8), (1.0, 1e-10, 1.5), (1e6, 1.0, 0.6)])
def test_int8_engine_scales_and_conditioning(sf2, noise, ell):
    """kernel scales far from 1 (the fixed-point image of K_XZ is scaled by σ_f², that of W by its row norms) and a factor whose
    inverse has entries of 1e4 and more (SE kernel, noise 1e-10·σ_f²): the int8 engine stays at the fp64 engine's error level
    against the oracle — both lose the digits the conditioning takes, neither more than the other"""
    N, d, M = 700, 3, 1500
    X = synth.points(1, N, d)
    Z = synth.points(2, M, d) * 1.1 - 0.05
    y = synth.objective(X, 0.05) * np.sqrt(sf2)
    st = O.fit(O.SE, ell, sf2, noise, 0.0, X, y)
    _, var_o = O.predict(st, Z)
    m64, m8 = _fit_pair(O.SE, ell, sf2, noise, X, y)
    var64 = abo.posterior_var(m64, Z)
    var8 = abo.posterior_var(m8, Z)
    e64 = np.max(np.abs(var64 - var_o)) / sf2
    e8 = np.max(np.abs(var8 - var_o)) / sf2
    case = f"int8/scale_sf2_{sf2:g}_noise_{noise:g}"
    check(case, "var", e8, 1e-6)
    check(case, "var_fp64_engine", e64, 1e-6)
    assert e8 <= 4 * e64 + 1e-13, (e8, e64)
    _, linv = abo.get_factor(m8)[0], abo.get_factor(m8)[2]
    check(case, "log10_max_abs_Linv", float(np.log10(np.max(np.abs(linv)) * np.sqrt(sf2))), 12.0)


@pytest.mark.parametrize("Np,Mc,nvalid,n", [(128, 128, 100, 14), (640, 384, 600, 14), (1280, 896, 1280, 12), (384, 256, 384, 16)])
def test_engine_on_arbitrary_operands_against_its_cpu_restatement(Np, Mc, nvalid, n):
    """the quantisers, the residue GEMM and the reconstruction on operands that are NOT a GP's: a lower-triangular W with rows of
    very different norms and entries over twelve orders of magnitude, K_XZ of both signs — against oracle/ozaki_oracle.py (same
    fixed-point images, products in exact integer arithmetic) and against a long-double product of the fp64 operands"""
    import torch
    from oracle import ozaki_oracle as Zo
    rng = np.random.default_rng(Np + Mc)
    W = np.tril(rng.standard_normal((Np, Np)) * 10.0 ** rng.uniform(-6, 6, (Np, 1)) * 10.0 ** rng.uniform(-3, 0, (Np, Np)))
    W[nvalid:] = 0.0
    W[:, nvalid:] = 0.0
    W[np.arange(nvalid, Np), np.arange(nvalid, Np)] = 1.0          # identity padding, as the library keeps it
    kmax = 3.7
    K = rng.uniform(-kmax, kmax, (Mc, Np)) * (rng.random((Mc, Np)) < 0.7)
    K[:, nvalid:] = 0.0
    Wd, Kd = torch.from_numpy(W).cuda(), torch.from_numpy(K).cuda()
    part = torch.full((Np // 128, Mc), -1.0, dtype=torch.float64).cuda()
    torch.cuda.synchronize()
    abo._lib.check(abo._lib.lib().abo_test_oz_contract(0, Wd.data_ptr(), Np, Np, nvalid, Kd.data_ptr(), Np, Mc, kmax, n,
                                                       part.data_ptr(), Mc))
    got = part.cpu().numpy()
    V, _ = Zo.contract(W[:nvalid, :nvalid], K[:, :nvalid].T.copy(), kmax, n)
    Vp = np.zeros((Np, Mc))
    Vp[:nvalid] = V
    want = (Vp.reshape(Np // 128, 128, Mc) ** 2).sum(1)
    scale = np.maximum(want, 1e-300)
    assert np.max(np.abs(got - want) / scale) < 1e-13          # same images, exact products: only the order of the 128 squares differs
    Vld = W[:nvalid, :nvalid].astype(np.longdouble) @ K[:, :nvalid].T.astype(np.longdouble)
    wld = np.zeros((Np, Mc), dtype=np.longdouble)
    wld[:nvalid] = Vld
    wld = (wld.reshape(Np // 128, 128, Mc) ** 2).sum(1)
    err = float(np.max(np.abs(got - wld) / np.maximum(wld, 1e-300)))
    check(f"int8/arbitrary_Np{Np}_n{n}", "sumsq_rel_vs_long_double", err, {12: 1e-7, 14: 1e-11, 16: 1e-11}[n])


@pytest.mark.parametrize("family,d,N,ell", [(O.MATERN52, 3, 150, 0.25), (O.SE, 2, 200, 3.0), (O.MATERN72, 5, 120, 1.0)])
def test_gradient_enhanced_model_on_the_int8_engine(family, d, N, ell):
    """posterior of a gradient-enhanced model (the rows of its K_XZ mix function and derivative covariances, apart by √c/ℓ): the
    engine takes the product as (W·D⁻¹)(D·K_XZ·E) with exact power-of-two D (training rows) and E (candidate outputs) — short and long
    lengthscales alike stay at the fp64 engine's error against oracle/grad_oracle.py.  Function-value posterior (mean_and_var), all
    outputs of the candidates (posterior_grad_var: GradientGP.jl:951-953, rows by outputs, the chunk straddles output boundaries) and
    the per-point covariance blocks + GradientNormUCB scores (posterior_grad_cov: GradientGP.jl:966-971, gradNormUCB.jl:43-51, where
    the reconstruction writes V itself)."""
    from oracle import grad_oracle as G
    from tests.test_gpu_gradient_gp import make_grad
    p = d + 1
    X = synth.points(1, N, d)
    f = np.sin(2 * np.pi * X).sum(axis=1) / np.sqrt(d)
    gF = 2 * np.pi * np.cos(2 * np.pi * X) / np.sqrt(d)
    Ys = np.column_stack([f, gF])
    Zc = synth.points(2, 700, d)
    st = G.fit(family, ell, 1.3, 1e-3, np.zeros(p), X, Ys)
    _, vf = G.predict(st, Zc)
    _, vall = G.predict_grad(st, Zc)
    Zs = Zc[:37]
    covs = np.stack([G.predict_grad(st, Zs[j:j + 1], cov=True)[1] for j in range(len(Zs))])
    ucb = G.grad_norm_ucb(st, Zs, 2.0)
    pv = G.prior_var(family, ell, 1.3, d)                   # per-output prior variances: the scale of each output's error
    out = {}
    for eng in ("fp64", "int8"):
        want = abo._lib.CONTRACT_INT8 if eng == "int8" else abo._lib.CONTRACT_FP64
        m = abo.update(make_grad(family, ell, 1.3, 1e-3, p, contraction=eng), X, Ys)
        mu, var = abo.mean_and_var(m, Zc)
        assert m.timings()["contraction_engine"] == want
        va = abo.posterior_grad_var(m, Zc)
        assert m.timings()["contraction_engine"] == want
        mu_pm, cv, sc = abo.posterior_grad_cov(m, Zs, beta=2.0, return_all=True)
        assert m.timings()["contraction_engine"] == want
        out[eng] = (mu, var, va, cv, sc, abo.posterior_grad_mean(m, Zc))
    np.testing.assert_array_equal(out["int8"][0], out["fp64"][0])
    np.testing.assert_array_equal(out["int8"][5], out["fp64"][5])
    e8 = np.max(np.abs(out["int8"][1] - vf)) / 1.3
    e64 = np.max(np.abs(out["fp64"][1] - vf)) / 1.3
    case = f"int8/grad_fam{family}_d{d}_ell{ell:g}"
    check(case, "var", e8, 1e-8)
    check(case, "var_fp64_engine", e64, 1e-8)
    assert e8 <= 4 * e64 + 1e-13, (e8, e64)
    scale_rows = np.repeat(pv, len(Zc))
    a8 = np.max(np.abs(out["int8"][2] - vall) / scale_rows)
    a64 = np.max(np.abs(out["fp64"][2] - vall) / scale_rows)
    check(case, "var_all_outputs", a8, 1e-8)
    check(case, "var_all_outputs_fp64_engine", a64, 1e-8)
    assert a8 <= 4 * a64 + 1e-13, (a8, a64)
    cs = np.sqrt(np.outer(pv, pv))[None]
    c8 = np.max(np.abs(out["int8"][3] - covs) / cs)
    c64 = np.max(np.abs(out["fp64"][3] - covs) / cs)
    check(case, "cov_blocks", c8, 1e-8)
    check(case, "cov_blocks_fp64_engine", c64, 1e-8)
    assert c8 <= 4 * c64 + 1e-13, (c8, c64)
    us = np.maximum(np.abs(ucb), 1e-6)
    u8 = np.max(np.abs(out["int8"][4] - ucb) / us)
    u64 = np.max(np.abs(out["fp64"][4] - ucb) / us)
    check(case, "grad_norm_ucb_rel", u8, 1e-6)
    check(case, "grad_norm_ucb_rel_fp64_engine", u64, 1e-6)


def test_engine_and_moduli_can_change_on_a_fitted_handle():
    """abo_set_contraction on a handle that already answered posterior calls: the cached residue planes of W belong to the plan they
    were made with and are rebuilt when the moduli count changes"""
    X = synth.points(1, 500, 3)
    y = synth.objective(X, 0.05)
    Zc = synth.points(2, 800, 3)
    m = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-3, contraction="int8"), X, y)
    v14 = abo.posterior_var(m, Zc)
    lib = abo._lib.lib()
    abo._lib.check(lib.abo_set_contraction(m._require(), abo._lib.CONTRACT_INT8, 12))
    v12 = abo.posterior_var(m, Zc)
    assert m.timings()["oz_nmod"] == 12 and 1e-13 < np.max(np.abs(v12 - v14)) < 1e-7
    abo._lib.check(lib.abo_set_contraction(m._require(), abo._lib.CONTRACT_FP64, 0))
    v64 = abo.posterior_var(m, Zc)
    assert m.timings()["contraction_engine"] == abo._lib.CONTRACT_FP64
    abo._lib.check(lib.abo_set_contraction(m._require(), abo._lib.CONTRACT_INT8, 14))
    np.testing.assert_array_equal(abo.posterior_var(m, Zc), v14)
    np.testing.assert_allclose(v64, v14, rtol=0, atol=1e-12)


def test_scratch_that_does_not_fit_halves_the_chunk_then_falls_back_to_fp64(monkeypatch):
    """the engine's chunk buffers (2 x 14 bytes per candidate and factor row) not fitting the device is not an error: the chunk is
    halved down to 4096 candidates, below that the call runs on the fp64 kernels — same results either way"""
    N, d, M = 1800, 4, 20000
    X = synth.points(1, N, d)
    y = synth.objective(X, 0.05)
    Zc = synth.points(2, M, d)
    ref = abo.posterior_var(abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-3, contraction="int8"), X, y), Zc)
    # 1792 factor rows pad to 2048: a 4096-candidate chunk takes 2·14·4096·2048 B = 224 MiB, the whole batch 1.1 GiB
    monkeypatch.setenv("ABO_OZ_SCRATCH_LIMIT_MB", "300")
    m = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-3, contraction="int8"), X, y)
    v = abo.posterior_var(m, Zc)
    assert m.timings()["contraction_engine"] == abo._lib.CONTRACT_INT8 and m.timings()["var_gemm_launches"] >= 4
    np.testing.assert_array_equal(v, ref)                   # exact products: the chunk size cannot show
    monkeypatch.setenv("ABO_OZ_SCRATCH_LIMIT_MB", "100")
    m = abo.update(make_model(O.MATERN52, 0.8, 1.0, 1e-3, contraction="int8"), X, y)
    v = abo.posterior_var(m, Zc)
    assert m.timings()["contraction_engine"] == abo._lib.CONTRACT_FP64
    np.testing.assert_allclose(v, ref, rtol=0, atol=1e-12)


@pytest.mark.parametrize("Np", [4096])
def test_adversarial_dense_equal_magnitude_rows_keep_their_guaranteed_bits(Np):
    """The engine's guarantee is stated per ROW of W: its fixed-point image keeps what lies within ≈ 50 bits below the row's L1
    norm (oz_rowscale_kernel: s_i = min(eP − 53 − e(L1), 52 − e(max))).  The worst case for an ENTRY is a dense row of equal
    magnitudes, L1/max ≈ N: the largest entry then keeps only 53 − (53 + e(L1) − eP + …) ≈ 55 − log2(N) bits — 43 at N = 4096.  This
    builds such rows (random signs, magnitudes within a factor 2, full rows k ≤ i at the bottom of the matrix), runs them through the
    engine's own quantisers, GEMM and reconstruction, and records (a) the bits the largest entry of the worst row keeps (43) and (b)
    the error of Σ V² against a long-double product: 1.7e-13 relative on an MI355X — the plain fp64 product of the same operands
    is at 2.7e-15, so this IS the engine's worst case showing, and it is seven orders of magnitude inside the north star's 1e-6
    (the lost digits lie 43 + ½·log2(N) bits below the sum they enter)."""
    import torch
    from oracle import ozaki_oracle as Zo
    rng = np.random.default_rng(7)
    Mc, n, kmax = 256, 14, 1.0
    W = np.tril(rng.choice([-1.0, 1.0], (Np, Np)) * rng.uniform(1.0, 2.0, (Np, Np)))
    W *= 10.0 ** rng.uniform(-3, 3, (Np, 1))                   # row scales are per row: the scale of a row cannot matter
    K = rng.uniform(-kmax, kmax, (Mc, Np))
    pl = Zo.plan(n)
    si = Zo.row_scales(W, pl["eP"])
    mx = np.abs(W).max(1)
    kept = si + np.frexp(mx)[1]                                  # bits of the row's largest entry above the rounding grid 2^-s_i
    l1_over_max = np.abs(W).sum(1) / mx
    worst = int(np.argmin(kept))
    assert l1_over_max[worst] > 0.5 * Np / 2 and worst > Np // 2
    assert 40 <= kept[worst] <= 46, kept[worst]                   # eP = 108: 108 − 53 − log2(L1/max) − (0…1) + …
    Wd, Kd = torch.from_numpy(W).cuda(), torch.from_numpy(K).cuda()
    part = torch.full((Np // 128, Mc), -1.0, dtype=torch.float64).cuda()
    torch.cuda.synchronize()
    abo._lib.check(abo._lib.lib().abo_test_oz_contract(0, Wd.data_ptr(), Np, Np, Np, Kd.data_ptr(), Np, Mc, kmax, n,
                                                       part.data_ptr(), Mc))
    got = part.cpu().numpy()
    Vld = W.astype(np.longdouble) @ K.T.astype(np.longdouble)
    want = (Vld.reshape(Np // 128, 128, Mc) ** 2).sum(1)
    err = float(np.max(np.abs(got - want) / want))
    # the same product in plain fp64 (what the fp64 engine's arithmetic does to these operands), for scale
    V64 = W @ K.T
    err64 = float(np.max(np.abs((V64.reshape(Np // 128, 128, Mc) ** 2).sum(1) - want) / want))
    case = f"int8/adversarial_dense_rows_N{Np}"
    check(case, "retained_bits_of_largest_entry_worst_row", float(kept[worst]), 53.0)
    check(case, "l1_over_max_worst_row", float(l1_over_max[worst]), float(Np))
    check(case, "sumsq_rel_vs_long_double", err, 1e-11)
    check(case, "sumsq_rel_vs_long_double_plain_fp64_product", err64, 1e-11)
    assert err <= 1e-12


def test_requested_engine_survives_append_rescale_and_hyperparameter_rebuilds():
    """A handle's explicit contraction engine must follow the model through every path that builds a new handle from it: a
    bordered append (also of a gradient-enhanced model: abo_append_grad), rescale_model, _update_model_parameters (what
    optimize_hyperparameters returns) — at a size where AUTO would pick the other engine, so a dropped choice shows."""
    from tests.test_gpu_gradient_gp import make_grad
    N, d = 1600, 3                                             # 1664 padded factor rows: AUTO → int8, requested: fp64
    X = synth.points(1, N + 1, d)
    y = synth.objective(X, 0.05)
    Zc = synth.points(2, 300, d)

    def engine(model):
        abo.posterior_var(model, Zc)
        return model.timings()["contraction_engine"]

    gp = make_model(O.MATERN52, 0.8, 1.0, 1e-3, contraction="fp64", n_max=N + 8)
    m = abo.update(gp, X[:N], y[:N])
    assert engine(m) == abo._lib.CONTRACT_FP64
    assert engine(abo.append(m, X[N], y[N])) == abo._lib.CONTRACT_FP64
    r = abo.rescale_model(gp, 2.0)
    assert r.contraction == "fp64" and engine(abo.update(r, X[:N], y[:N] / 2.0)) == abo._lib.CONTRACT_FP64
    u = abo._update_model_parameters(gp, 1.5 * abo.with_lengthscale(abo.Matern52Kernel(), 0.6))
    assert u.contraction == "fp64" and engine(abo.update(u, X[:N], y[:N])) == abo._lib.CONTRACT_FP64
    # and the other way round below the AUTO threshold: int8 requested at a size where AUTO takes fp64
    gs = make_model(O.MATERN52, 0.8, 1.0, 1e-3, contraction="int8", n_max=408)
    ms = abo.update(gs, X[:400], y[:400])
    assert engine(ms) == abo._lib.CONTRACT_INT8 and engine(abo.append(ms, X[400], y[400])) == abo._lib.CONTRACT_INT8
    # gradient-enhanced model: p·N = 4·420 = 1680 factor rows
    Ng = 420
    f = np.sin(2 * np.pi * X[:Ng + 1]).sum(axis=1)
    Y = np.column_stack([f, 2 * np.pi * np.cos(2 * np.pi * X[:Ng + 1])])
    gg = make_grad(O.MATERN52, 0.5, 1.0, 1e-2, d + 1, contraction="fp64", n_max=Ng + 4)
    mg = abo.update(gg, X[:Ng], Y[:Ng])
    assert engine(mg) == abo._lib.CONTRACT_FP64
    assert engine(abo.append(mg, X[Ng], Y[Ng])) == abo._lib.CONTRACT_FP64          # abo_append_grad copies the choice
    assert abo.rescale_model(gg, [2.0] * (d + 1)).contraction == "fp64"
    assert abo._update_model_parameters(gg, 1.5 * abo.with_lengthscale(abo.Matern52Kernel(), 0.6)).contraction == "fp64"


@pytest.mark.parametrize("N,d,M", [(100, 2, 300), (700, 3, 5000), (1400, 4, 9000), (2304, 8, 20000)])      # 1, 3, 6 (not a multiple of 4), 9 row blocks
def test_persistent_residue_gemm_on_ragged_launch_shapes(N, d, M):
    """The persistent residue GEMM (one workgroup per CU drawing tiles from per-XCD lists, the next tile's operands prefetched during
    the epilogue) where the launch is not a whole number of 4-row-block groups, the last column group is ragged, a tile is a single
    block, or there are fewer tiles than CUs: against the oracle and the fp64 engine, and the same bits on a second run (exact integer
    arithmetic, fixed-order sums; round 2's one-tile-per-workgroup kernel it used to be compared with left the library in round 6)."""
    import hashlib
    X, y = synth.standardized_problem(N, d, 0.03)
    Z = synth.points(2, M, d)
    kern = lambda: 1.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.9)
    m = abo.update(abo.HipStandardGP(kern(), 1e-3, contraction="int8"), X, y)
    v = abo.posterior_var(m, Z)
    assert m.timings()["contraction_engine"] == abo._lib.CONTRACT_INT8
    v2 = abo.posterior_var(abo.update(abo.HipStandardGP(kern(), 1e-3, contraction="int8"), X, y), Z)
    assert hashlib.sha256(v.tobytes()).hexdigest() == hashlib.sha256(v2.tobytes()).hexdigest()
    v64 = abo.posterior_var(abo.update(abo.HipStandardGP(kern(), 1e-3, contraction="fp64"), X, y), Z)
    st = O.fit(O.MATERN52, 0.9, 1.0, 1e-3, 0.0, X, y)
    vo = O.predict(st, Z)[1]
    case = f"oz_ragged/N{N}_d{d}_M{M}"
    check(case, "var_vs_oracle", np.max(np.abs(v - vo)), 1e-9)
    check(case, "var_vs_fp64_engine", np.max(np.abs(v - v64)), 1e-9)
