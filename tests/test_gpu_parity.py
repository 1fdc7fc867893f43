"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs, against the committed golden fixtures, and — at full benchmark sizes — through
size-independent properties.  Tolerances (fp64, north star: ≤1e-6 relative):
    |Δμ|  ≤ tol · max(1, max|μ|)          |Δσ²| ≤ tol · σ_f²
with tol = 1e-9 on well-conditioned cases (the reference's own closed-form tests use atol 1e-10
at N = 3) and the north-star 1e-6 as the hard bar everywhere."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth
from oracle import gp_oracle as O

from tests.parity_record import check

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FAMS = {O.SE: abo.SqExponentialKernel, O.MATERN52: abo.Matern52Kernel, O.MATERN72: abo.ApproxMatern72Kernel,
        O.MATERN32: abo.Matern32Kernel}


def make_model(family, ell, sf2, noise, mean_c=0.0, **kw):
    mean = abo.ConstMean(mean_c) if mean_c != 0.0 else None
    return abo.HipStandardGP(sf2 * abo.with_lengthscale(FAMS[family](), ell), noise, mean=mean, **kw)


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


# ------------------------------------------------------------------------------------------------
def test_library_loaded_and_version():
    assert abo._lib.lib().abo_abi_version() == abo._lib.ABI_VERSION == 7


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (256, 384, 128), (128, 256, 272)])
def test_mfma_gemm_core_against_fp64_reference(M, N, K):
    """A = asymmetric random, B = asymmetric random: catches swapped C/D lane maps and k-permutation
    mismatches of the fp64 MFMA tile core."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K + 6, generator=g, dtype=torch.float64)
    B = torch.randn(N, K + 2, generator=g, dtype=torch.float64)
    C0 = torch.randn(M, N, generator=g, dtype=torch.float64)
    Ad, Bd, Cd = A.cuda(), B.cuda(), C0.cuda().clone()
    torch.cuda.synchronize()
    st = abo._lib.lib().abo_test_gemm_nt(0, Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), M, N, K, K + 6, K + 2, N, -1.5, 0.5)
    abo._lib.check(st)
    ref = -1.5 * (A[:, :K] @ B[:, :K].T) + 0.5 * C0
    err = (Cd.cpu() - ref).abs().max().item()
    assert err < 1e-12 * K, err


def test_small_tile_gemm_is_bit_identical_to_the_lds_tiled_one(monkeypatch):
    """launches with few 128×128 tiles run as 32×32 workgroups without LDS; same k order → same bits"""
    import torch
    g = torch.Generator(device="cpu").manual_seed(7)
    out = {}
    for M, N, K in [(128, 128, 128), (384, 256, 512)]:
        A = torch.randn(M, K + 6, generator=g, dtype=torch.float64).cuda()
        B = torch.randn(N, K + 2, generator=g, dtype=torch.float64).cuda()
        C0 = torch.randn(M, N, generator=g, dtype=torch.float64).cuda()
        for mode in ("0", "1"):
            monkeypatch.setenv("ABO_GEMM_SMALL", mode)
            Cd = C0.clone()
            torch.cuda.synchronize()
            abo._lib.check(abo._lib.lib().abo_test_gemm_nt(0, A.data_ptr(), B.data_ptr(), Cd.data_ptr(), M, N, K, K + 6, K + 2, N, -1.5, 0.5))
            out[mode] = Cd.cpu()
        assert torch.equal(out["0"], out["1"])
        ref = -1.5 * (A.cpu()[:, :K] @ B.cpu()[:, :K].T) + 0.5 * C0.cpu()
        assert (out["1"] - ref).abs().max().item() < 1e-12 * K


# ------------------------------------------------------------------------------------------------
def test_lower_tiles_only_syrk_launch_is_bit_identical_to_the_2d_one(monkeypatch):
    """The square trailing updates of the factorisation run as a 1-D launch over the lower tiles (XCD-chunked tile order);
    a tile's arithmetic does not depend on the workgroup that runs it: the factor equals the 2-D launch's bit for bit
    (18 and 20 tile rows: a last super-row of 2 and a whole number of them)."""
    for N in (2700, 3000):
        X = synth.points(1, N, 5)
        y = synth.objective(X, 0.05)
        monkeypatch.delenv("ABO_GEMM_NO_SWIZZLE", raising=False)
        L1, a1, W1 = abo.get_factor(abo.update(make_model(O.MATERN52, 1.0, 1.0, 1e-3), X, y))
        monkeypatch.setenv("ABO_GEMM_NO_SWIZZLE", "1")
        L0, a0, W0 = abo.get_factor(abo.update(make_model(O.MATERN52, 1.0, 1.0, 1e-3), X, y))
        monkeypatch.delenv("ABO_GEMM_NO_SWIZZLE")
        np.testing.assert_array_equal(L1, L0)
        np.testing.assert_array_equal(a1, a0)
        np.testing.assert_array_equal(W1, W0)


@pytest.mark.parametrize("N", [256, 640, 1100, 2304, 2700, 6300])
def test_panel_chain_against_the_oracle_and_its_failing_pivot(N):
    """The factorisation's panel chain (chol.hip: potf2_pipe_kernel — side work beside the register steps — → trsm_stream_kernel fed by
    the operand stream it leaves → in-strip update; api.hip: factorise) at sizes with one to many strips, ragged last strips and a
    super-strip: L, L⁻¹ and α against the oracle's LAPACK factor (round 6: parity is the oracle's tolerance — the round-5 variants
    that were held to round 4's BITS are gone from the library, and with them the rule that froze the arithmetic); a refit on the
    same handle gives the same bits (fixed-order reductions, no atomics); a failed pivot reports LAPACK's `info` — the first
    non-positive leading minor — in the first, a middle and the last 16-column sub-step of a block (bayesian_opt.jl:126-141 relies
    on the exception, test_bayesian_opt.jl:759-784 on the failure)."""
    X = synth.points(1, N, 5)
    y = synth.objective(X, 0.05)
    gp = make_model(O.MATERN52, 1.0, 1.3, 1e-4)
    m = abo.update(gp, X, y)
    L, al, Li = abo.get_factor(m)
    L2, al2, Li2 = abo.get_factor(abo.update(gp, X, y))
    np.testing.assert_array_equal(L2, L)
    np.testing.assert_array_equal(Li2, Li)
    np.testing.assert_array_equal(al2, al)
    st = O.fit(O.MATERN52, 1.0, 1.3, 1e-4, 0.0, X, y)
    case = f"chain/N{N}_d5"
    check(case, "L", np.max(np.abs(L - st.L)) / np.sqrt(1.3 + 1e-4), 1e-9)
    check(case, "LinvL_minus_I", np.max(np.abs(Li @ st.L - np.eye(N))), 1e-7)
    check(case, "alpha_rel", np.max(np.abs(al - st.alpha)) / max(1.0, np.max(np.abs(st.alpha))), 1e-6)
    for bad in (5, 128 + 70, N - 7, N - 3):
        Xb = X.copy()
        Xb[bad] = Xb[1]                                    # a duplicate point, no noise: the factorisation fails at row bad + 1
        with pytest.raises(abo.PosDefException) as e:
            abo.update(make_model(O.MATERN52, 1.0, 1.3, 0.0), Xb, y)
        assert e.value.info == bad + 1, (bad, e.value.info)


@pytest.mark.parametrize("name", ["kat1", "kat3", "kat4", "kat5"])
def test_kat_closed_forms(name):
    """test/test_surrogates.jl:59-105,:145-170; test/test_acquisition.jl; test/test_bayesian_opt.jl:
    461-487,:512-559 — the reference's own atol is 1e-10."""
    c = _load("kat.json")[name]
    m = abo.update(make_model(c["family"], c["ell"], c["sigma_f2"], c["noise_var"], c["mean_c"]), c["X"], c["y"])
    mu = abo.posterior_mean(m, c["Z"])
    var = abo.posterior_var(m, c["Z"])
    np.testing.assert_allclose(mu, c["mu"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(var, c["var"], rtol=0, atol=1e-12)
    assert abs(abo.nlml_fitted(m) - c["nlml"]) < 1e-11
    assert np.all(var >= 0.0)
    if "ei" in c:
        np.testing.assert_allclose(abo.ExpectedImprovement(c["xi"], c["best_y"])(m, c["Z"]), c["ei"], rtol=1e-8, atol=1e-15)
        np.testing.assert_allclose(abo.UpperConfidenceBound(c["beta"])(m, c["Z"]), c["ucb"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(abo.ProbabilityImprovement(c["xi"], c["best_y"])(m, c["Z"]), c["pi"], rtol=1e-8, atol=1e-15)


def test_kat1_scalar_inputs_and_nlml_signature():
    # 1-D Vector{Float64} inputs and the scalar wrappers (StandardGP.jl:329-347); nlml(model, params, x, y)
    c = _load("kat.json")["kat1"]
    gp = abo.HipStandardGP(abo.SqExponentialKernel(), 0.1)
    m = abo.update(gp, [0.0, 0.5, 1.0], [0.0, 0.25, 1.0])
    assert abs(abo.posterior_mean(m, 0.25)[0] - c["mu"][0]) < 1e-12
    assert abs(abo.posterior_var(m, 0.25)[0] - c["var"][0]) < 1e-12
    assert abs(abo.nlml(gp, [np.log(1.0), np.log(1.0)], [0.0, 0.5, 1.0], [0.0, 0.25, 1.0]) - c["nlml"]) < 1e-11
    assert m.noise_var == 0.1 and m.gpx is not None and gp.gpx is None


def test_kat6_posdef_failure_and_rollback_protocol():
    """test/test_bayesian_opt.jl:749-786: appending a 1e-12 duplicate with zero noise must raise
    PosDefException so the driver restores the previous model."""
    c = _load("kat.json")["kat6"]
    gp = make_model(c["family"], c["ell"], c["sigma_f2"], c["noise_var"])
    prev = abo.update(gp, c["X"][:2], c["y"][:2])
    with pytest.raises(abo.PosDefException) as e:
        abo.update(prev, c["X"], c["y"])
    assert e.value.info == 3
    # the previous model is untouched and still usable
    assert np.isfinite(abo.posterior_mean(prev, [[0.0, 0.0]])[0])
    # opt-in jitter rescues the same system
    ok = abo.update(make_model(c["family"], c["ell"], c["sigma_f2"], 0.0, jitter=1e-8), c["X"], c["y"])
    assert np.isfinite(abo.posterior_var(ok, [[0.0, 0.0]])[0])


def test_dimension_mismatch():
    # test/test_bayesian_opt.jl:788-817
    m = abo.update(make_model(O.SE, 1.0, 1.0, 1e-2), [[0.0, 0.0], [1.0, 1.0]], [0.0, 1.0])
    with pytest.raises(abo.DimensionMismatch):
        abo.posterior_mean(m, [[0.5]])
    with pytest.raises(abo.DimensionMismatch):
        abo.update(m, [[0.0, 0.0], [1.0]], [0.0, 1.0])
    with pytest.raises(abo.DimensionMismatch):
        abo.update(m, [[0.0, 0.0], [1.0, 1.0]], [0.0])
    with pytest.raises(ValueError):
        abo.posterior_mean(make_model(O.SE, 1.0, 1.0, 1e-2), [[0.5, 0.5]])     # gpx === nothing


def test_copy_semantics():
    # test/test_surrogates.jl:130-143
    m = abo.update(make_model(O.SE, 1.0, 1.0, 0.1), [0.0, 0.5, 1.0], [0.0, 0.25, 1.0])
    c = abo.copy(m)
    assert c is not m and c.gp == m.gp and c.gpx is not m.gpx and c.noise_var == m.noise_var
    a = abo.posterior_mean(m, [0.25])
    del m
    np.testing.assert_array_equal(abo.posterior_mean(c, [0.25]), a)


@pytest.mark.parametrize("i", range(16))
def test_random_small_golden(i):
    c = _load("random_small.json")[i]
    m = abo.update(make_model(c["family"], c["ell"], c["sigma_f2"], c["noise_var"], c["mean_c"]), c["X"], c["y"])
    mu, var = abo.mean_and_var(m, c["Z"])
    tol = max(1e-12, 1e-13 / min(1.0, c["noise_var"] / c["sigma_f2"]))
    np.testing.assert_allclose(mu, c["mu"], rtol=0, atol=tol * max(1.0, np.max(np.abs(c["mu"]))))
    np.testing.assert_allclose(var, c["var"], rtol=0, atol=tol * c["sigma_f2"])
    assert abs(abo.nlml_fitted(m) - c["nlml"]) < 10 * tol * max(1.0, abs(c["nlml"]))
    np.testing.assert_allclose(abo.ExpectedImprovement(c["xi"], c["best_y"])(m, c["Z"]), c["ei"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(abo.UpperConfidenceBound(c["beta"])(m, c["Z"]), c["ucb"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(abo.ProbabilityImprovement(c["xi"], c["best_y"])(m, c["Z"]), c["pi"], rtol=1e-6, atol=1e-12)


# ------------------------------------------------------------------------------------------------
CASES = [
    # family, d, N, M, ell, sf2, noise, mean_c   (N chosen to hit: <128, exact tile, ragged block counts 3 and 5)
    (O.SE, 1, 5, 300, 1.0, 1.0, 1e-6, 0.0),
    (O.MATERN52, 1, 25, 1000, 0.3, 1.0, 1e-6, 0.0),      # C1 shape
    (O.SE, 4, 128, 257, 0.5, 1.0, 1e-4, 0.0),
    (O.MATERN52, 8, 300, 1025, 1.0, 1.0, 1e-3, 0.7),
    (O.MATERN72, 3, 640, 513, 0.8, 2.5, 1e-3, 0.0),
    (O.MATERN32, 2, 1000, 4096, 0.6, 0.5, 1e-2, -1.2),
    (O.SE, 4, 1024, 8192, 0.5, 1.0, 1e-4, 0.0),          # C2 training shape
    (O.SE, 4, 1024, 65536, 0.5, 1.0, 1e-4, 0.0),         # C2 at its own size (BASELINE config 2: M = 65 536, UCB in the epilogue checks)
    (O.MATERN52, 16, 1152, 2048, 2.0, 1.0, 1e-2, 0.0),   # C5 dimension, 9 blocks
    (O.MATERN52, 5, 1664, 700, 0.9, 1.0, 1e-3, 0.3),     # 13 blocks: three 512-wide Cholesky strips + a 128 remainder
    (O.SE, 7, 2304, 600, 1.2, 1.5, 1e-3, 0.0),           # 18 blocks: both GEMM variants (small-launch and LDS-tiled) in one fit
    (O.MATERN52, 6, 2700, 400, 1.0, 1.0, 1e-3, 0.0),     # 22 blocks: the first trailing update (18 tile rows, not a multiple of
                                                         # the super-row height) goes through the 1-D lower-tiles-only launch
    (O.MATERN52, 5, 6300, 300, 1.2, 1.0, 1e-3, 0.0),     # 50 blocks: one super-strip of 1024 (≥ 6144 rows remain), plain strips of
                                                         # 512 behind it, a ragged last strip — the factorisation's third blocking level
    (O.MATERN72, 32, 200, 300, 3.0, 1.0, 1e-3, 0.0),     # largest supported dimension
    (O.MATERN32, 13, 520, 129, 1.5, 0.7, 1e-2, 0.0),     # odd dimension (padded to 16), ragged everything
]


@pytest.mark.parametrize("family,d,N,M,ell,sf2,noise,mean_c", CASES)
def test_against_oracle(family, d, N, M, ell, sf2, noise, mean_c):
    """Errors are scaled as DESIGN §4 states (μ by max(1, max|μ|), σ² by σ_f², L by √(σ_f²+σ²_n), α and NLML relative),
    recorded, and asserted against the hard bar `tol` (a conditioning-aware bound, ≤ the north star's 1e-6) and against
    100 × the error recorded on an MI355X for this very case (tests/parity_record.py)."""
    X = synth.points(1, N, d)
    Z = synth.points(2, M, d) * 1.2 - 0.1
    y = synth.objective(X, 0.05) + mean_c
    st = O.fit(family, ell, sf2, noise, mean_c, X, y)
    mu_o, var_o = O.predict(st, Z)
    m = abo.update(make_model(family, ell, sf2, noise, mean_c), X, y)
    L, alpha, Linv = abo.get_factor(m)
    cond = 1.0 + N * sf2 / noise                 # crude bound on cond(K)
    tol = max(1e-11, 4e-16 * cond)
    assert tol <= 1e-6
    case = f"oracle/fam{family}_d{d}_N{N}" + (f"_M{M}" if M >= 65536 else "")
    check(case, "L", np.max(np.abs(L - st.L)) / np.sqrt(sf2 + noise), tol)
    check(case, "LinvL_minus_I", np.max(np.abs(Linv @ st.L - np.eye(N))), tol * 10)
    check(case, "alpha_rel", np.max(np.abs(alpha - st.alpha)) / max(1.0, np.max(np.abs(st.alpha))), min(1e-6, tol * 1e3))
    mu, var = abo.mean_and_var(m, Z)
    check(case, "mu", np.max(np.abs(mu - mu_o)) / max(1.0, np.max(np.abs(mu_o))), min(1e-6, tol * 1e2))
    check(case, "var", np.max(np.abs(var - var_o)) / sf2, min(1e-6, tol * 1e2))
    check(case, "nlml_rel", abs(abo.nlml_fitted(m) - O.nlml(st)) / max(1.0, abs(O.nlml(st))), min(1e-6, tol * 1e2))
    # separate entry points agree with the fused one bit for bit
    np.testing.assert_array_equal(abo.posterior_mean(m, Z), mu)
    np.testing.assert_array_equal(abo.posterior_var(m, Z), var)
    best = float(np.min(y))
    for acq, kind, p0 in ((abo.ExpectedImprovement(0.01, best), O.ACQ_EI, 0.01), (abo.UpperConfidenceBound(2.0), O.ACQ_UCB, 2.0),
                          (abo.ProbabilityImprovement(0.01, best), O.ACQ_PI, 0.01)):
        s = acq(m, Z)
        np.testing.assert_allclose(s, O.acquisition(kind, mu, var, p0, best), rtol=1e-10, atol=1e-14)
        k = min(100, M)
        s2, tv, ti = abo.evaluate(acq, m, Z, k=k)
        np.testing.assert_array_equal(s2, s)
        ov, oi = O.top_k(s, k)
        np.testing.assert_array_equal(ti, oi)
        np.testing.assert_array_equal(tv, ov)


def test_candidate_on_training_point_and_variance_floor():
    # EI == max(Δ, 0) branch (σ² ≤ 1e-12) and the +1e-18 FiniteGP jitter
    X = synth.points(1, 40, 2)
    y = synth.objective(X)
    m = abo.update(make_model(O.SE, 0.7, 1.0, 1e-14), X, y)
    var = abo.posterior_var(m, X[:5])
    assert np.all(var < 1e-9)
    ei = abo.ExpectedImprovement(0.0, float(y.min()) + 1.0)(m, X[:5])
    mu = abo.posterior_mean(m, X[:5])
    np.testing.assert_allclose(ei, O.expected_improvement(mu, var, float(y.min()) + 1.0, 0.0), rtol=1e-9, atol=1e-12)


def test_topk_ties_nan_and_short_batches():
    # constant prior far from data → all scores tie → lowest indices win (stable sortperm)
    m = abo.update(make_model(O.SE, 0.01, 1.0, 1e-2), [[0.0, 0.0]], [0.0])
    Z = np.full((5000, 2), 50.0) + np.arange(5000)[:, None]
    s, tv, ti = abo.evaluate(abo.UpperConfidenceBound(2.0), m, Z, k=64)
    assert np.all(s == s[0])
    np.testing.assert_array_equal(ti, np.arange(64))
    # M < k: tail is (NaN, −1)
    s, tv, ti = abo.evaluate(abo.UpperConfidenceBound(2.0), m, Z[:3], k=8)
    np.testing.assert_array_equal(ti, [0, 1, 2, -1, -1, -1, -1, -1])
    assert np.all(np.isnan(tv[3:]))
    # NaN candidates sort first (isless puts NaN last; rev=true flips it)
    Zn = Z[:300].copy()
    Zn[17, 0] = np.nan
    s, tv, ti = abo.evaluate(abo.UpperConfidenceBound(2.0), m, Zn, k=4, idx_base=1000)
    assert ti[0] == 1017 and np.isnan(tv[0]) and ti[1] == 1000
    # empty batch
    s, tv, ti = abo.evaluate(abo.UpperConfidenceBound(2.0), m, np.zeros((0, 2)), k=2)
    assert s.shape == (0,) and ti.tolist() == [-1, -1]


def test_argmax_path_of_a_large_batch_is_the_first_entry_of_the_sorted_selection():
    """k = 1 over more than 16384 scores runs two plain reductions instead of the block sorts (a greedy q-EI pick): same total
    order — larger score first, ties → lowest index, NaN first — i.e. exactly entry 0 of a k = 2 selection."""
    X, y = synth.standardized_problem(60, 3)
    m = abo.update(make_model(O.MATERN52, 0.7, 1.0, 1e-3), X, y)
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    for M in (16385, 50000, 131072 + 5):
        Z = synth.points(7, M, 3)
        s, tv1, ti1 = abo.evaluate(acq, m, Z, k=1, idx_base=7)
        _, tv2, ti2 = abo.evaluate(acq, m, Z, k=2, idx_base=7)
        assert ti1[0] == ti2[0] == 7 + int(np.argmax(s)) and tv1[0] == tv2[0] == s.max()
    # all scores tie → index 0; a NaN candidate comes first wherever it sits
    far = abo.update(make_model(O.SE, 0.01, 1.0, 1e-2), [[0.0, 0.0]], [0.0])
    Z = np.full((40000, 2), 50.0) + np.arange(40000)[:, None]
    _, tv, ti = abo.evaluate(abo.UpperConfidenceBound(2.0), far, Z, k=1)
    assert ti[0] == 0
    Z[33333, 1] = np.nan
    _, tv, ti = abo.evaluate(abo.UpperConfidenceBound(2.0), far, Z, k=1, idx_base=5)
    assert ti[0] == 5 + 33333 and np.isnan(tv[0])


def test_device_resident_candidates_match_host_path():
    import torch
    X, y = synth.standardized_problem(500, 4)
    Z = synth.points(2, 3000, 4)
    m = abo.update(make_model(O.SE, 0.5, 1.0, 1e-4), X, y)
    acq = abo.UpperConfidenceBound(2.0)
    s_h, tv_h, ti_h = abo.evaluate(acq, m, Z, k=10)
    s_d, tv_d, ti_d = abo.evaluate(acq, m, torch.from_numpy(Z).cuda(), k=10)
    np.testing.assert_array_equal(s_d.cpu().numpy(), s_h)
    np.testing.assert_array_equal(ti_d.cpu().numpy(), ti_h)
    md = abo.update(make_model(O.SE, 0.5, 1.0, 1e-4), torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda())
    np.testing.assert_array_equal(abo.posterior_var(md, Z), abo.posterior_var(m, Z))


def test_chunking_is_invisible():
    X, y = synth.standardized_problem(300, 3)
    Z = synth.points(2, 5000, 3)
    a = abo.update(make_model(O.MATERN52, 0.9, 1.0, 1e-3), X, y)
    b = abo.update(make_model(O.MATERN52, 0.9, 1.0, 1e-3, chunk=256), X, y)
    for f in (abo.posterior_mean, abo.posterior_var):
        np.testing.assert_array_equal(f(a, Z), f(b, Z))


def test_standardisation_equivalence_on_device():
    # test/test_bayesian_opt.jl:238-356 (|Δμ|, |Δσ²| < 1e-10)
    X = synth.points(1, 60, 2)
    y = synth.objective(X) * 3.0 + 5.0
    Z = synth.points(2, 200, 2)
    base = make_model(O.SE, 0.7, 1.3, 1e-3)
    mu0, sd0 = abo.get_mean_std(base, y, "mean_only")
    a = abo.update(base, X, abo.std_y(base, y, mu0, sd0))
    b = abo.update(make_model(O.SE, 0.7, 1.3, 1e-3, mean_c=mu0), X, y)
    np.testing.assert_allclose(abo.posterior_mean(a, Z) + mu0, abo.posterior_mean(b, Z), atol=1e-10)
    np.testing.assert_allclose(abo.posterior_var(a, Z), abo.posterior_var(b, Z), atol=1e-10)
    # mean_scale ≡ rescaled model (scale / σ², noise / σ²) on standardised targets
    mu1, sd1 = abo.get_mean_std(base, y, "mean_scale")
    rm = abo.update(abo.rescale_model(base, sd1), X, abo.std_y(base, y, mu1, sd1))
    m_un, v_un = abo.unstandardized_mean_and_var(rm, Z, [mu1, sd1])
    ref = abo.update(make_model(O.SE, 0.7, 1.3, 1e-3, mean_c=mu1), X, y)
    np.testing.assert_allclose(m_un, abo.posterior_mean(ref, Z), atol=1e-9)
    np.testing.assert_allclose(v_un, abo.posterior_var(ref, Z), atol=1e-9)


def test_bo_loop_plumbing_c1():
    """BASELINE config 1 shape: 1-D f(x) = sin(x) on [0, 10], 5 initial points, EI, mean_only
    standardisation, 20 iterations (loop semantics of test/test_bayesian_opt.jl:186-223: the
    iteration count advances and the incumbent never worsens); grid stage only."""
    rng = np.random.default_rng(42)
    dom = abo.ContinuousDomain([0.0], [10.0])
    xs = list(rng.uniform(0, 10, 5))
    ys = [float(np.sin(x)) for x in xs]
    gp = abo.HipStandardGP(abo.Matern52Kernel(), 1e-9)
    best_hist = []
    for it in range(20):
        mu_y, _ = abo.get_mean_std(gp, ys, "mean_only")
        yst = abo.std_y(gp, ys, mu_y, 1.0)
        try:
            model = abo.update(gp, xs, yst)
        except abo.PosDefException:      # driver protocol: roll back the last point and stop
            xs.pop(); ys.pop()           # (src/bayesian_opt.jl:126-141)
            break
        acq = abo.update(abo.ExpectedImprovement(0.0, 0.0), yst, model)
        x_new = abo.optimize_acquisition(acq, model, dom, n_grid=10_000, n_local=100, rng=rng)
        xs.append(float(x_new[0]))
        ys.append(float(np.sin(x_new[0])))
        best_hist.append(min(ys))
    assert len(xs) >= 10
    assert all(b1 <= b0 + 1e-15 for b0, b1 in zip(best_hist, best_hist[1:]))
    assert best_hist[-1] < -0.99          # min of sin on [0, 10] is −1 at 3π/2


@pytest.mark.parametrize("sharded", [False, True])
def test_bo_loop_through_the_driver_shaped_calls_2d(sharded):
    """The stock EGO loop of the reference (src/bayesian_opt.jl:364-449) written out with this backend's drop-in calls, on a 2-D
    function with several local minima: standardise → update → hyper-parameter MLE every 5 iterations (optimize_hyperparameters:
    the Dual-aware nlml's arithmetic, abo_nlml_grad) → acquisition update → optimize_acquisition in ONE call (device grid, top-100,
    on-device L-BFGS of every start) → evaluate → append, on a single handle and on a two-shard group: the incumbent never
    worsens and the global minimum is found (the next test pins the sharded proposal to the single-device one bit for bit)."""
    from abstractbayesopt.jl_amd import multigpu

    def f(x):                                                  # minimum −1.9133 at (0.75, 0.25)-ish basin
        return float(np.sin(2 * np.pi * x[0]) + np.cos(2 * np.pi * x[1]) + 2.0 * (x[0] - 0.7) ** 2 + 2.0 * (x[1] - 0.45) ** 2)

    dom = abo.ContinuousDomain(np.zeros(2), np.ones(2))
    X0 = synth.points(3, 8, 2)
    xs = [X0[i] for i in range(8)]
    ys = [f(x) for x in xs]
    k0 = 1.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.3)
    gp = abo.HipShardedGP(k0, 1e-6, devices=(0, 0)) if sharded else abo.HipStandardGP(k0, 1e-6)
    best_hist, traj = [], []
    for it in range(25):
        mu_y, sd_y = abo.get_mean_std(gp, ys, "mean_scale")
        yst = abo.std_y(gp, ys, mu_y, sd_y)
        Xa = np.asarray(xs)
        if it % 5 == 0 and not sharded:
            # hyper-parameter MLE on the single handle (bayesian_opt.jl:388 cadence); the group run reuses its trajectory below
            old = [np.log(abo.get_lengthscale(gp)[0]), np.log(abo.get_scale(gp)[0])]
            gp = abo.optimize_hyperparameters(gp, Xa, yst, old, num_restarts=1, domain=dom, rng=np.random.default_rng(it))
        model = abo.update(gp, Xa, yst)
        acq = abo.update(abo.ExpectedImprovement(0.01, 0.0), yst, model)
        x_new = abo.optimize_acquisition_device(acq, model, dom, n_grid=10_000, n_local=100, seed=1000 + it)
        assert np.all(x_new >= 0.0) and np.all(x_new <= 1.0)
        traj.append(x_new.copy())
        xs.append(x_new)
        ys.append(f(x_new))
        best_hist.append(min(ys))
    assert all(b1 <= b0 + 1e-15 for b0, b1 in zip(best_hist, best_hist[1:]))
    grid = synth.points(9, 200_000, 2)
    true_min = float(np.min(np.sin(2 * np.pi * grid[:, 0]) + np.cos(2 * np.pi * grid[:, 1]) + 2.0 * (grid[:, 0] - 0.7) ** 2
                            + 2.0 * (grid[:, 1] - 0.45) ** 2))
    assert best_hist[-1] <= true_min + 2e-3, (best_hist[-1], true_min)


def test_sharded_optimize_acquisition_walks_the_single_device_trajectory():
    """one BO step from the same state on a single handle and on a two-/three-shard group: the proposed point is the same bits"""
    d = 2
    X = synth.points(3, 40, d)
    y = np.sin(2 * np.pi * X[:, 0]) + np.cos(2 * np.pi * X[:, 1])
    y = (y - y.mean()) / y.std(ddof=1)
    dom = abo.ContinuousDomain(np.zeros(d), np.ones(d))
    k0 = 1.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.3)
    one = abo.update(abo.HipStandardGP(k0, 1e-6), X, y)
    acq = abo.update(abo.ExpectedImprovement(0.01, 0.0), y, one)
    x1 = abo.optimize_acquisition_device(acq, one, dom, n_grid=10_000, n_local=100, seed=77)
    for devs in ((0, 0), (0, 0, 0)):
        grp = abo.update(abo.HipShardedGP(k0, 1e-6, devices=devs), X, y)
        np.testing.assert_array_equal(abo.optimize_acquisition_device(acq, grp, dom, n_grid=10_000, n_local=100, seed=77), x1)


# ------------------------------------------------------------------------------------------------
_C3_ORACLE = []


def c3_oracle():
    """(X, y, oracle state) of BASELINE config 3 / 4's training set: the independent host-LAPACK refit (≈ 7 s), computed once per
    test process (test_full_size_parity_c3 here and the 8-shard config-4 test in tests/test_gpu_multigpu.py share it)."""
    if not _C3_ORACLE:
        X, y = synth.standardized_problem(8192, 8, 0.03)
        _C3_ORACLE.append((X, y, O.fit(O.MATERN52, 1.0, 1.0, 1e-3, 0.0, X, y)))
    return _C3_ORACLE[0]


def test_full_size_parity_c3():
    """BASELINE config 3 at the size the metric is quoted on (N = 8192, d = 8, Matérn-5/2, M = 2²⁰, EI): an INDEPENDENT
    oracle refit (O.fit: LAPACK dpotrf on the host, ≈7 s) and the oracle posterior on 2304 candidates taken from the first,
    a middle and the last chunk of the 2²⁰-candidate batch the device scores in one abo_acq call — nothing on the oracle
    side comes from the device.  Plus the size-independent properties: prior recovery far from the data, determinism."""
    N, d, M = 8192, 8, 1 << 20
    ell, sf2, noise = 1.0, 1.0, 1e-3
    X, y, st = c3_oracle()
    Z = synth.points(2, M, d)
    m = abo.update(make_model(O.MATERN52, ell, sf2, noise), X, y)
    L, alpha, Linv = abo.get_factor(m)
    case = "c3/N8192_d8_M1048576"
    check(case, "L", np.max(np.abs(L - st.L)) / np.sqrt(sf2 + noise), 1e-9)
    check(case, "alpha_rel", np.max(np.abs(alpha - st.alpha)) / max(1.0, np.max(np.abs(st.alpha))), 1e-6)
    rows = np.array([0, 1, 127, 128, 129, 4095, 4096, 8000, 8191])
    check(case, "LinvL_minus_I_rows", np.max(np.abs(Linv[rows] @ st.L - np.eye(N)[rows])), 1e-8)
    check(case, "nlml_rel", abs(abo.nlml_fitted(m) - O.nlml(st)) / abs(O.nlml(st)), 1e-9)
    best = float(y.min())
    acq = abo.ExpectedImprovement(0.01, best)
    import torch
    Zd = torch.from_numpy(Z).cuda()
    s_d, tv, ti = abo.evaluate(acq, m, Zd, k=100)                     # the benchmarked call: all 2²⁰ candidates, top-100
    # this test is the headline kernel's only full-size oracle check: the engine that ran the scored call is PINNED (AUTO falls
    # back to the fp64 kernels without an error when the int8 scratch does not fit the device — that must fail here, not pass)
    t = m.timings()
    assert t["contraction_engine"] == abo._lib.CONTRACT_INT8 and t["oz_nmod"] == 14, t
    mu_d, var_d = abo.mean_and_var(m, Zd)
    t = m.timings()
    assert t["contraction_engine"] == abo._lib.CONTRACT_INT8 and t["oz_nmod"] == 14, t
    s, mu, var = s_d.cpu().numpy(), mu_d.cpu().numpy(), var_d.cpu().numpy()
    sl = np.concatenate([np.arange(0, 768), np.arange(M // 2 - 384, M // 2 + 384), np.arange(M - 768, M)])
    mu_o, var_o = O.predict(st, Z[sl])
    check(case, "mu", np.max(np.abs(mu[sl] - mu_o)) / max(1.0, np.max(np.abs(mu_o))), 1e-8)
    check(case, "var", np.max(np.abs(var[sl] - var_o)) / sf2, 1e-8)
    # the other engine on the same slice of the same model (same factor: only the contraction differs)
    m64 = abo.update(make_model(O.MATERN52, ell, sf2, noise, contraction="fp64"), X, y)
    mu64, var64 = abo.mean_and_var(m64, Z[sl])
    assert m64.timings()["contraction_engine"] == abo._lib.CONTRACT_FP64
    np.testing.assert_array_equal(mu64, mu[sl])                       # the mean never goes through the contraction
    check(case, "var_fp64_engine", np.max(np.abs(var64 - var_o)) / sf2, 1e-8)
    check(case, "var_between_engines", np.max(np.abs(var64 - var[sl])) / sf2, 1e-8)
    del m64
    ei_o = O.expected_improvement(mu_o, var_o, best, 0.01)
    check(case, "ei_abs", np.max(np.abs(s[sl] - ei_o)), 1e-9)
    # the selection is the stable reverse sort of the device's own scores, bit for bit, over the full 2²⁰
    ov, oi = O.top_k(s, 100)
    np.testing.assert_array_equal(ti.cpu().numpy(), oi)
    np.testing.assert_array_equal(tv.cpu().numpy(), ov)
    assert np.all(var > 0) and np.all(var <= sf2 + 1e-12)
    far = abo.mean_and_var(m, np.full((3, d), 100.0))
    assert np.max(np.abs(far[0])) < 1e-12 and np.max(np.abs(far[1] - sf2)) < 1e-12
    mu2, var2 = abo.mean_and_var(m, Zd)
    assert torch.equal(mu2, mu_d) and torch.equal(var2, var_d)


@pytest.mark.parametrize("family", [O.SE, O.MATERN52, O.MATERN72, O.MATERN32])
def test_kappa_device_math(family):
    """The kernel-matrix generator uses its own exp / sqrt sequences (no libm calls): check them against
    60-digit mpmath over the whole useful range, including the κ(0) = 1 neighbourhood the reference
    special-cases (GradientGP.jl:94-101) and the underflow tail."""
    import mpmath as mp
    import torch
    mp.mp.dps = 60
    rng = np.random.default_rng(family)
    d2 = np.concatenate([[0.0, 1e-300, 1e-200, 1e-30, 1e-12, 1e-10, 9.9e-11, 1e-6, 0.5, 1.0, 2.0, 100.0, 1e3, 1e5, 2e5, 1e6],
                         10.0 ** rng.uniform(-8, 3, 3000), rng.uniform(0, 50, 3000)])
    x = torch.from_numpy(d2).cuda()
    out = torch.empty_like(x)
    torch.cuda.synchronize()
    abo._lib.check(abo._lib.lib().abo_test_kappa(0, family, x.data_ptr(), out.data_ptr(), x.numel()))
    got = out.cpu().numpy()

    def exact(v):
        v = mp.mpf(float(v))
        if family == O.SE:
            return mp.exp(-v / 2)
        d = mp.sqrt(v)
        if family == O.MATERN52:
            return (1 + mp.sqrt(5) * d + 5 * v / 3) * mp.exp(-mp.sqrt(5) * d)
        if family == O.MATERN72:
            return (1 + mp.sqrt(7) * d + mp.mpf(14) / 5 * v + 7 * mp.sqrt(7) / 15 * v * d) * mp.exp(-mp.sqrt(7) * d)
        return (1 + mp.sqrt(3) * d) * mp.exp(-mp.sqrt(3) * d)

    ref = np.array([float(exact(v)) for v in d2])
    big = ref > 1e-300
    rel = np.abs(got[big] - ref[big]) / ref[big]
    # the exponent's argument −c·d is itself rounded to fp64, so the attainable accuracy is ~ε·|argument|
    arg = {O.SE: 0.5 * d2, O.MATERN52: np.sqrt(5 * d2), O.MATERN72: np.sqrt(7 * d2), O.MATERN32: np.sqrt(3 * d2)}[family]
    allowed = 1.5e-15 + 2.5e-16 * arg[big]
    assert np.all(rel <= allowed), (np.max(rel / allowed), d2[big][np.argmax(rel / allowed)])
    assert np.all(np.abs(got[~big] - ref[~big]) < 1e-300)
    assert got[0] == 1.0 and np.all(np.isfinite(got))


def test_optimize_acquisition_refinement_stage():
    """acq_utils.jl:55-71: every start is refined inside the box and the best refined point is returned.
    The batched projected L-BFGS must (i) never lose against its start, (ii) stay in the box, (iii) reach
    the optimum SciPy's L-BFGS-B finds on the CPU oracle's acquisition from the same start."""
    from scipy.optimize import minimize
    from abstractbayesopt.jl_amd.acquisition import refine_starts
    d = 3
    X, y = synth.standardized_problem(60, d, 0.02)
    fam, ell, sf2, noise = O.MATERN52, 0.5, 1.0, 1e-3
    m = abo.update(make_model(fam, ell, sf2, noise), X, y)
    st = O.fit(fam, ell, sf2, noise, 0.0, X, y)
    lower, upper = np.zeros(d), np.ones(d)
    for acq, oracle in ((abo.UpperConfidenceBound(2.0), lambda z: O.upper_confidence_bound(*O.predict(st, z), 2.0)),
                        (abo.ExpectedImprovement(0.01, float(y.min())),
                         lambda z: O.expected_improvement(*O.predict(st, z), float(y.min()), 0.01))):
        starts = synth.points(7, 12, d)
        f0 = acq(m, starts)
        xr, fr = refine_starts(acq, m, starts, lower, upper)
        assert np.all(fr >= f0 - 1e-15)
        assert np.all(xr >= lower) and np.all(xr <= upper)
        np.testing.assert_allclose(acq(m, xr), fr, rtol=0, atol=1e-12)
        for i in range(4):
            res = minimize(lambda z: -float(oracle(z[None, :])[0]), starts[i], method="L-BFGS-B",
                           bounds=list(zip(lower, upper)), options={"ftol": 1e-14, "gtol": 1e-8})
            # same basin is not guaranteed for every start; require ≥ SciPy's value up to the stopping tolerances
            assert fr[i] >= -res.fun - 1e-5 * max(1.0, abs(res.fun)) or fr[i] >= f0[i], (i, fr[i], -res.fun)
    dom = abo.ContinuousDomain(lower, upper)
    rng = np.random.default_rng(1)
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    best, starts, vals = abo.optimize_acquisition(acq, m, dom, n_grid=2000, n_local=20, rng=rng, return_starts=True)
    assert acq(m, [best])[0] >= vals[0] - 1e-15 and np.all(best >= lower) and np.all(best <= upper)
    rng = np.random.default_rng(1)
    grid_only = abo.optimize_acquisition(acq, m, dom, n_grid=2000, n_local=20, rng=rng, refine=False)
    np.testing.assert_array_equal(grid_only, starts[0])


@pytest.mark.parametrize("family,d,N", [(O.SE, 2, 40), (O.MATERN52, 4, 300), (O.MATERN72, 3, 130), (O.MATERN32, 1, 77),
                                        (O.MATERN52, 8, 1024)])
def test_nlml_gradient_against_oracle_finite_differences(family, d, N):
    """abo_nlml_grad vs central differences of the CPU oracle's NLML in (log ℓ, log σ_f²) — what the reference
    obtains with ForwardDiff in optimize_hyperparameters (bayesian_opt.jl:253-285)."""
    X = synth.points(1, N, d)
    y = synth.objective(X, 0.1) + 0.3
    ell, sf2, noise, mean_c = 0.6, 1.7, 1e-2, 0.3
    gp = make_model(family, 1.0, 1.0, noise, mean_c)
    p = np.array([np.log(ell), np.log(sf2)])
    v, g = abo.nlml_and_grad(gp, p, X, y)

    def f(q):
        return O.nlml(O.fit(family, float(np.exp(q[0])), float(np.exp(q[1])), noise, mean_c, X, y))

    assert abs(v - f(p)) <= 1e-9 * max(1.0, abs(v))
    h = 1e-5
    fd = np.array([(f(p + h * e) - f(p - h * e)) / (2 * h) for e in np.eye(2)])
    np.testing.assert_allclose(g, fd, rtol=2e-6, atol=1e-6 * max(1.0, abs(v)))


@pytest.mark.parametrize("which", ["standard", "gradient"])
def test_nlml_on_dual_parameters_is_the_chain_rule_of_the_analytic_gradient(which):
    """The stock driver's `autodiff=:forward` (bayesian_opt.jl:276-285) evaluates nlml / nlml_ls on ForwardDiff.Dual
    parameters; the Julia shim (integration/julia/HipStandardGP.jl) answers with Dual(v, g₁·∂p₁ + g₂·∂p₂) from
    abo_nlml_grad.  Julia is not in this image, so the same arithmetic runs here on a dual-number stand-in
    (hyperparams.Dual): seeded with the unit partials it must return the gradient — checked against central differences of
    the value-only nlml —, through a linear reparametrisation p = A·t it must return Aᵀg, and with a plain-float scale
    (nlml_ls: only log ℓ is a Dual) only the first component."""
    from abstractbayesopt.jl_amd.hyperparams import Dual
    if which == "standard":
        X = synth.points(1, 300, 3)
        y = synth.objective(X, 0.05)
        gp = make_model(O.MATERN52, 1.0, 1.0, 1e-3)
        nl, nl_ls = abo.nlml, abo.nlml_ls
    else:
        from tests.test_gpu_gradient_gp import make_grad
        X = synth.points(1, 60, 2)
        f = np.sin(2 * np.pi * X).sum(axis=1)
        y = np.column_stack([f, 2 * np.pi * np.cos(2 * np.pi * X)])
        gp = make_grad(O.MATERN52, 1.0, 1.0, 1e-3, 3)
        nl, nl_ls = abo.nlml, abo.nlml_ls
    p = np.array([np.log(0.7), np.log(1.4)])
    out = nl(gp, [Dual(p[0], [1.0, 0.0]), Dual(p[1], [0.0, 1.0])], X, y)
    v = nl(gp, p, X, y)
    assert isinstance(out, Dual) and out.value == v
    h = 1e-5
    fd = np.array([(nl(gp, p + h * e, X, y) - nl(gp, p - h * e, X, y)) / (2 * h) for e in np.eye(2)])
    np.testing.assert_allclose(out.partials, fd, rtol=5e-6, atol=1e-6 * max(1.0, abs(v)))
    g = out.partials
    # chain rule through p = A t, three seeds
    A = np.array([[0.5, -1.0, 2.0], [3.0, 0.25, -0.5]])
    t = [Dual(0.0, e) for e in np.eye(3)]
    pd = [p[i] + A[i, 0] * t[0] + A[i, 1] * t[1] + A[i, 2] * t[2] for i in range(2)]
    out2 = nl(gp, pd, X, y)
    assert out2.value == v
    np.testing.assert_allclose(out2.partials, A.T @ g, rtol=1e-14, atol=1e-14 * np.abs(g).max())
    # nlml_ls: the scale is the clamped start value, a plain float (bayesian_opt.jl:247, :255)
    out3 = nl_ls(gp, Dual(p[0], [1.0]), float(p[1]), X, y)
    assert out3.value == v and out3.partials.shape == (1,) and out3.partials[0] == g[0]


def test_optimize_hyperparameters_improves_nlml():
    """test/test_bayesian_opt.jl:597-653: the optimised kernel has a lower NLML than the starting one, stays in
    the bounds, and length_scale_only keeps the scale."""
    d, N = 2, 80
    X = synth.points(1, N, d)
    y = np.sin(3 * X[:, 0]) + np.cos(2 * X[:, 1])
    dom = abo.ContinuousDomain(np.zeros(d), np.ones(d))
    gp = abo.HipStandardGP(1.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.05), 1e-4)
    old = [np.log(0.05), np.log(1.0)]
    v0, _ = abo.nlml_and_grad(gp, old, X, y)
    new = abo.optimize_hyperparameters(gp, X, y, old, num_restarts=3, domain=dom, rng=np.random.default_rng(0))
    assert new.gpx is None                                  # un-conditioned, like _update_model_parameters
    ell, sc = abo.get_lengthscale(new)[0], abo.get_scale(new)[0]
    v1, g1 = abo.nlml_and_grad(gp, [np.log(ell), np.log(sc)], X, y)
    assert v1 < v0 - 1.0
    lo, hi = abo.lengthscale_bounds(X, dom, rng=np.random.default_rng(0))
    assert max(lo.min(), 1e-6) * 0.999 <= ell <= hi.max() * 1.001 and 1e-3 <= sc <= 1e6
    only = abo.optimize_hyperparameters(gp, X, y, old, length_scale_only=True, domain=dom, rng=np.random.default_rng(0))
    assert abo.get_scale(only) == [1.0] and abo.get_lengthscale(only)[0] != 0.05


@pytest.mark.parametrize("N,d,ns", [(7, 2, 1000), (1024, 8, 10000), (300, 40, 2001)])
def test_fill_distance_on_the_device_equals_the_host_scan(N, d, ns):
    """monte_carlo_fill_distance (src/BO_utils.jl:140-159) — the lower length-scale bound of optimize_hyperparameters — with the
    N × n_samples scan on the device (abo_fill_distance): the same sample points (drawn on the host from the same RNG state) give the
    host NumPy scan's value to rounding, whichever coordinate count."""
    X = synth.points(1, N, d) * 3.0 - 1.0
    dom = abo.ContinuousDomain(np.full(d, -1.5), np.full(d, 2.5))
    h_dev = abo.monte_carlo_fill_distance(X, dom, n_samples=ns, rng=np.random.default_rng(5), device=0)
    h_host = abo.monte_carlo_fill_distance(X, dom, n_samples=ns, rng=np.random.default_rng(5))
    assert h_host > 0 and abs(h_dev - h_host) <= 1e-13 * h_host, (h_dev, h_host)
    lo_d, hi_d = abo.lengthscale_bounds(X, dom, rng=np.random.default_rng(2), device=0)
    lo_h, hi_h = abo.lengthscale_bounds(X, dom, rng=np.random.default_rng(2))
    np.testing.assert_allclose(lo_d, lo_h, rtol=1e-13)
    np.testing.assert_array_equal(hi_d, hi_h)


def test_ensemble_acquisition_is_weighted_sum_on_one_posterior():
    # test/test_acquisition.jl:223-253: ensemble value == Σ wᵢ·acqᵢ
    import torch
    X, y = synth.standardized_problem(200, 3, 0.05)
    Z = synth.points(2, 1500, 3)
    m = abo.update(make_model(O.MATERN52, 0.6, 1.0, 1e-3), X, y)
    ei, ucb, pi = abo.ExpectedImprovement(0.01, float(y.min())), abo.UpperConfidenceBound(2.0), abo.ProbabilityImprovement(0.01, float(y.min()))
    ens = abo.EnsembleAcquisition([1.0, 2.0, 1.0], [ei, ucb, pi])
    want = 0.25 * ei(m, Z) + 0.5 * ucb(m, Z) + 0.25 * pi(m, Z)
    np.testing.assert_allclose(ens(m, Z), want, rtol=0, atol=1e-14)
    got_dev = ens(m, torch.from_numpy(Z).cuda())
    np.testing.assert_allclose(got_dev.cpu().numpy(), want, rtol=0, atol=1e-14)
    best = abo.optimize_acquisition(ens, m, abo.ContinuousDomain(np.zeros(3), np.ones(3)), n_grid=3000, n_local=10,
                                    rng=np.random.default_rng(0))
    assert np.all(best >= 0) and np.all(best <= 1)


def test_device_latin_hypercube():
    """abo_lhs: one point per stratum in every coordinate (the LHS property of acq_utils.jl:44-47), shards of
    the same design are consistent, different seeds/coordinates give different permutations."""
    for n in (1, 7, 1000, 65537):
        lower, upper = np.array([0.0, -2.0, 5.0]), np.array([1.0, 2.0, 6.0])
        Z = abo.device_latin_hypercube(n, lower, upper, seed=12345).cpu().numpy()
        assert Z.shape == (n, 3)
        for c in range(3):
            strata = np.floor((Z[:, c] - lower[c]) / (upper[c] - lower[c]) * n).astype(np.int64)
            assert sorted(strata.tolist()) == list(range(n)), (n, c)
    n = 5000
    full = abo.device_latin_hypercube(n, [0.0, 0.0], [1.0, 1.0], seed=7).cpu().numpy()
    part = abo.device_latin_hypercube(n, [0.0, 0.0], [1.0, 1.0], seed=7, first=1234, count=777).cpu().numpy()
    np.testing.assert_array_equal(full[1234:1234 + 777], part)
    other = abo.device_latin_hypercube(n, [0.0, 0.0], [1.0, 1.0], seed=8).cpu().numpy()
    assert not np.array_equal(full, other)
    assert abs(np.corrcoef(full[:, 0], full[:, 1])[0, 1]) < 0.05       # coordinates permuted independently
    # grid generated, scored and reduced on the device
    X, y = synth.standardized_problem(100, 2, 0.05)
    m = abo.update(make_model(O.SE, 0.4, 1.0, 1e-3), X, y)
    dom = abo.ContinuousDomain([0.0, 0.0], [1.0, 1.0])
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    b, starts, vals = abo.optimize_acquisition(acq, m, dom, n_grid=20000, n_local=50, rng=np.random.default_rng(3),
                                               device_grid=True, refine=False, return_starts=True)
    np.testing.assert_allclose(acq(m, starts), vals, rtol=0, atol=1e-15)
    assert np.all(np.diff(vals) <= 0)


@pytest.mark.parametrize("N", [700, 600])
def test_contraction_tilings_against_the_oracle(N):
    """The two tilings of the fp64 N²·M contraction, each on the sizes it serves: the 256×128 tile with the structural zeros of the
    diagonal blocks skipped at 16-row granularity (Np a multiple of 256: N = 700 → Np = 768) and the 128×128 tile (an odd number of
    128-row blocks: N = 600 → Np = 640) against the oracle — also on an appended view whose last row block carries masked stale rows
    (a discarded branch of the shared factor storage)."""
    X, y = synth.standardized_problem(N, 5, 0.05)
    Z = synth.points(2, 3000, 5)
    m = abo.update(make_model(O.MATERN52, 0.9, 1.0, 1e-3, n_max=-(-N // 128) * 128, contraction="fp64"), X, y)
    m2 = abo.append(abo.append(m, Z[0], 0.3), Z[1], -0.2)
    dead = abo.append(m2, Z[2], 0.0)                          # a discarded branch leaves a stale factor row
    del dead
    st0 = O.fit(O.MATERN52, 0.9, 1.0, 1e-3, 0.0, X, y)
    st2 = O.fit(O.MATERN52, 0.9, 1.0, 1e-3, 0.0, np.vstack([X, Z[:2]]), np.append(y, [0.3, -0.2]))
    case = f"tilings/N{N}_d5"
    check(case, "var", np.max(np.abs(abo.posterior_var(m, Z) - O.predict(st0, Z)[1])), 1e-9)
    check(case, "var_appended_view", np.max(np.abs(abo.posterior_var(m2, Z) - O.predict(st2, Z)[1])), 1e-9)


def test_near_singular_and_extreme_hyperparameters():
    """test/test_bayesian_opt.jl:889-937: very close points with tiny noise must give a finite prediction or a
    PosDefException (never garbage or a crash); extreme lengthscales stay finite and differ."""
    gp = abo.HipStandardGP(abo.SqExponentialKernel(), 1e-12)
    try:
        m = abo.update(gp, [0.0, 1e-10, 2e-10], [1.0, 1.001, 1.002])
        assert np.isfinite(abo.posterior_mean(m, [0.5])[0])
    except abo.PosDefException as e:
        assert 1 <= e.info <= 3
    xs, ys = [-1.0, 0.0, 1.0], [1.0, 0.0, 1.0]
    big = abo.update(abo.HipStandardGP(abo.SqExponentialKernel(), 0.01), xs, ys)
    small = abo.update(abo.HipStandardGP(1.0 * abo.with_lengthscale(abo.SqExponentialKernel(), 1e-6), 0.01), xs, ys)
    pb, ps = abo.posterior_mean(big, [0.5])[0], abo.posterior_mean(small, [0.5])[0]
    assert np.isfinite(pb) and np.isfinite(ps) and abs(pb - ps) > 0.01
    huge = abo.update(abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 1e3), 0.01), xs, ys)
    assert np.isfinite(abo.posterior_var(huge, [0.5])[0])
    # non-finite inputs do not hang or crash: NaN propagates to a PosDefException (first pivot is NaN)
    with pytest.raises(abo.PosDefException):
        abo.update(gp, [0.0, np.nan, 1.0], [1.0, 2.0, 3.0])


def test_kernel_matrix_symmetry_and_psd_through_the_factor():
    # test/test_bayesian_opt.jl:489-510: K symmetric PSD — here: L·Lᵀ reproduces the oracle's K + σ²I
    X = synth.points(1, 200, 3)
    for fam in (O.SE, O.MATERN52, O.MATERN72, O.MATERN32):
        m = abo.update(make_model(fam, 0.7, 1.4, 1e-4), X, synth.objective(X))
        L, _, _ = abo.get_factor(m)
        K = L @ L.T
        Ko = O.kernel_matrix(fam, 0.7, 1.4, X) + 1e-4 * np.eye(200)
        assert np.max(np.abs(K - Ko)) < 1e-12 and np.max(np.abs(K - K.T)) == 0.0
        assert np.all(np.diag(L) > 0) and np.all(np.triu(L, 1) == 0)


def test_checkpoint_resume_by_pickle():
    """Checkpoint = hyper-parameters + the training data read back from the device (abo_get_data); resume = refit.
    (The reference has no serialisation code: a BOStruct is rebuilt from xs, ys and hyper-parameters.)"""
    import pickle
    X, y = synth.standardized_problem(200, 3, 0.05)
    Z = synth.points(2, 500, 3)
    m = abo.update(make_model(O.MATERN52, 0.8, 1.4, 1e-3, mean_c=0.2, n_max=256), X, y)
    Xb, yb = abo.training_data(m)
    np.testing.assert_array_equal(Xb, X)
    np.testing.assert_array_equal(yb, y)
    m2 = pickle.loads(pickle.dumps(m))
    assert m2.gp == m.gp and m2.noise_var == m.noise_var and m2.gpx is not None and m2.gpx is not m.gpx
    np.testing.assert_array_equal(abo.posterior_mean(m2, Z), abo.posterior_mean(m, Z))     # same data, same kernels
    np.testing.assert_array_equal(abo.posterior_var(m2, Z), abo.posterior_var(m, Z))
    ma = abo.append(m, Z[0], 0.3)                                                          # appended view → refit
    mb = pickle.loads(pickle.dumps(ma))
    assert abo.training_data(mb)[0].shape == (201, 3)
    np.testing.assert_allclose(abo.posterior_var(mb, Z), abo.posterior_var(ma, Z), rtol=0, atol=1e-11)
    empty = pickle.loads(pickle.dumps(make_model(O.SE, 1.0, 1.0, 0.1)))
    assert empty.gpx is None
    g = abo.update(abo.GradientGP(abo.SqExponentialKernel(), 3, 0.1), [[0.0, 0.0], [0.5, 0.5], [1.0, 1.0]],
                   [[1.0, 0.1, 0.1], [0.5, 0.0, 0.0], [0.0, -0.1, -0.1]])
    g2 = pickle.loads(pickle.dumps(g))
    assert isinstance(g2, abo.GradientGP) and g2.p == 3
    np.testing.assert_array_equal(abo.posterior_grad_mean(g2, [[0.25, 0.25]]), abo.posterior_grad_mean(g, [[0.25, 0.25]]))


def test_bad_arguments_are_refused_not_crashed():
    """every C-ABI entry returns a status for unusable arguments (the process never aborts: SURVEY §8(b) error convention)"""
    L = abo._lib.lib()
    X, y = synth.standardized_problem(50, 3, 0.05)
    m = abo.update(make_model(O.SE, 0.8, 1.0, 1e-3, n_max=64), X, y)
    Z = synth.points(2, 20, 3)
    assert abo.posterior_mean(m, np.zeros((0, 3))).shape == (0,)                      # empty batch is fine
    with pytest.raises(abo.DimensionMismatch):
        abo.posterior_var(m, np.zeros((4, 2)))
    _, tv, ti = abo.evaluate(abo.UpperConfidenceBound(2.0), m, Z, k=5000)            # k > M: every candidate, then (NaN, −1)
    assert sorted(ti[:20].tolist()) == list(range(20)) and np.all(ti[20:] == -1) and np.all(np.isnan(tv[20:]))
    with pytest.raises((ValueError, abo.AboError)):
        abo.evaluate(abo.UpperConfidenceBound(2.0), m, Z, k=-1)
    with pytest.raises((ValueError, abo.AboError)):
        abo.update(make_model(O.SE, 1.0, 1.0, 1e-3), np.zeros((2, 70000)), np.zeros(2))   # d beyond 65536
    with pytest.raises((ValueError, abo.AboError)):
        abo.update(make_model(O.SE, 1.0, 1.0, 1e-3), np.zeros((0, 3)), np.zeros(0))   # no data
    cands = abo.ResidentCandidates(m, Z)
    with pytest.raises((ValueError, abo.AboError)):
        cands.exclude(20)
    with pytest.raises((ValueError, abo.AboError)):
        cands.point(-1)
    with pytest.raises(abo.DimensionMismatch):
        abo.append(m, [0.1, 0.2], 0.0)                                               # wrong dimension of the new point
    for bad in (float("nan"), float("inf")):
        with pytest.raises((abo.PosDefException, ValueError, abo.AboError)):
            abo.update(make_model(O.SE, 1.0, 1.0, 1e-3), np.array([[0.0, 0.0], [bad, 1.0]]), np.array([0.0, 1.0]))
    # the model and the candidate set are still usable afterwards
    assert np.isfinite(abo.posterior_mean(m, Z)).all() and np.isfinite(cands.mean_and_var()[1]).all()
