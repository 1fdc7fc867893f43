"""GPU tests (-m gpu) of the on-device refinement stage of optimize_acquisition (csrc/refine.hip; C-ABI abo_refine,
abo_optimize_acquisition, abo_mgpu_optimize_acquisition) — reference: src/acquisition_functions/acq_utils.jl:33-73.

The reference differentiates the acquisition function by finite differences of M = 1 posterior calls; the library evaluates
the analytic gradient (∇μ, ∇σ² through ∂k/∂x, L⁻¹, L⁻ᵀ and the closed-form partials of EI / UCB / PI).  Checked here:
the gradient against central differences of the library's own scores AND of the CPU oracle's, the refinement against SciPy's
L-BFGS-B on the oracle's acquisition and against the host-driven finite-difference loop, the one-call entry against its parts,
the sharded group against the single handle (bit for bit)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth
from abstractbayesopt.jl_amd.acquisition import _refine_starts_fd, acquisition_value_and_grad, refine_starts
from oracle import gp_oracle as O

from tests.parity_record import check
from tests.test_gpu_parity import make_model


def _oracle_acq(acq, st):
    if isinstance(acq, abo.UpperConfidenceBound):
        return lambda z: O.upper_confidence_bound(*O.predict(st, z), acq.beta)
    kind = O.ACQ_EI if isinstance(acq, abo.ExpectedImprovement) else O.ACQ_PI
    return lambda z: O.acquisition(kind, *O.predict(st, z), acq.xi, acq.best_y)


BASIN_RADIUS = 0.25      # max-norm distance beyond which two end points are different local maximisers (boxes of width 1.4 … 2,
                         # length scales 0.5 …: the other-maximiser cases measured on an MI355X lie 0.57 … 1.5 apart, tools/refine_diag.py)


def _against_scipy(case, name, oracle, refine, starts, xr, fr, lower, upper, all_starts_bar=1e-5):
    """The refinement's end points against an INDEPENDENT optimiser on INDEPENDENT arithmetic: SciPy L-BFGS-B on the oracle's
    acquisition from the same starts.  A start passes when the device's value reaches SciPy's (to 1e-5 relative), or when the two
    end points are different local maximisers (more than BASIN_RADIUS apart — SciPy's first line-search step is long and often leaves
    the start's basin); for those the device's end point must be a stationary point of the ORACLE's acquisition: a device run with
    the stopping rules tightened ends at the same value, and the oracle's projected central-difference gradient there is zero to
    1e-4 of the box-scale slope.  A start that ends below SciPy INSIDE SciPy's basin fails — the count is recorded and must be 0
    (round 4 found 5 of 12 such starts in one case: failed line searches along quasi-Newton directions, since then retried along
    the projected gradient; profiles/r05_refine_diag.txt has the per-start table before and after)."""
    from scipy.optimize import minimize
    xt, ft = refine(max_iter=1000, g_tol=1e-9, f_abstol=1e-300, x_abstol=1e-12)
    below_in_basin, other, fs_best, fs_best_same, fr_best_same = 0, 0, -np.inf, -np.inf, -np.inf
    for i in range(len(starts)):
        res = minimize(lambda z: -float(oracle(z[None, :])[0]), starts[i], method="L-BFGS-B", bounds=list(zip(lower, upper)),
                       options={"ftol": 1e-14, "gtol": 1e-8})
        fs = -res.fun
        tol = 1e-5 * max(1.0, abs(fs))
        fs_best = max(fs_best, fs)
        if np.max(np.abs(xr[i] - res.x)) <= BASIN_RADIUS:      # both optimisers ended at the same maximiser from this start
            fs_best_same, fr_best_same = max(fs_best_same, fs), max(fr_best_same, float(fr[i]))
        if fr[i] >= fs - tol:
            continue
        if np.max(np.abs(xr[i] - res.x)) <= BASIN_RADIUS and np.max(np.abs(xt[i] - res.x)) <= BASIN_RADIUS:
            below_in_basin += 1
            continue
        other += 1
        if ft[i] >= fs - tol:
            continue                               # the reference's own rules (f_abstol on a plateau of EI / PI) ended the run: the
                                                   # tightened run climbs to SciPy's maximiser
        # another local maximiser: the tightened run stays there, and it is a stationary point of the oracle's acquisition
        assert abs(ft[i] - fr[i]) <= 1e-3 * max(1.0, abs(ft[i])), (name, i, fr[i], ft[i])
        g = _fd4(oracle, xt[i][None, :], 1e-5)[0]
        # (a bound counts as active within 1e-8: the tightened run creeps to within an ulp of a face without landing on it)
        pg = np.where(((xt[i] <= lower + 1e-8) & (g < 0)) | ((xt[i] >= upper - 1e-8) & (g > 0)), 0.0, g)
        if np.max(np.abs(pg)) > 1e-4 * max(1.0, abs(ft[i])):
            # not stationary after 1000 iterations: only a start on a plateau may do that (|∇| ≤ g_tol at the start itself)
            g0 = _fd4(oracle, np.clip(starts[i], lower, upper)[None, :], 1e-5)[0]
            assert np.max(np.abs(g0)) <= 1e-4, (name, i, np.max(np.abs(pg)), np.max(np.abs(g0)))
    check(case, f"{name}_starts_below_scipy_inside_its_basin", float(below_in_basin), 0.0, tighten=False)
    check(case, f"{name}_fraction_of_starts_at_another_maximiser", other / len(starts), 0.75, tighten=False)    # (measured ≤ 0.57: GradientNormUCB, d = 3)
    # what optimize_acquisition returns is the BEST over the starts (acq_utils.jl:66-72): the device's against SciPy's
    # held to the per-start tolerance (every recorded value of the small cases is ≤ 5.4e-8): a result 1e-4 below SciPy's best fails.
    # all_starts_bar is looser only where the caller says why (d = 8, 8 starts: which basin a start's FIRST line-search step lands
    # in differs between the two optimisers in both directions — profiles/r06_refine_diag_c3.txt); the same-basin figure below is
    # held to 1e-5 everywhere.
    check(case, f"{name}_best_of_starts_shortfall_rel", max(0.0, fs_best - float(np.max(fr))) / max(1.0, abs(fs_best)), all_starts_bar, tighten=False)
    if np.isfinite(fs_best_same):
        check(case, f"{name}_best_of_same_basin_starts_shortfall_rel", max(0.0, fs_best_same - fr_best_same) / max(1.0, abs(fs_best_same)),
              1e-5, tighten=False)


@pytest.mark.parametrize("family,d,N", [(O.SE, 1, 30), (O.MATERN52, 3, 200), (O.MATERN72, 8, 500), (O.MATERN32, 2, 64),
                                        (O.MATERN52, 40, 150), (O.SE, 5, 1100)])
def test_analytic_acquisition_gradient_against_central_differences(family, d, N):
    X, y = synth.standardized_problem(N, d, 0.03)
    # a noisy model on purpose: σ stays at a few tenths everywhere, so z = Δ/σ is moderate and EI / PI vary smoothly over the
    # whole box (with noise 1e-3 on a dense design σ collapses, and EI / PI are flat 0 or flat Δ / 1 at most test points)
    ell, sf2, noise = 0.7 * np.sqrt(d), 1.3, 0.1
    m = abo.update(make_model(family, ell, sf2, noise), X, y)
    st = O.fit(family, ell, sf2, noise, 0.0, X, y)
    Z = synth.points(5, 24, d) * 2.0 - 0.5                  # half of them outside the data's hull: σ from a few hundredths to ≈ √σ_f²
    # best_y = the median target, so that the improvement Δ = best − ξ − μ has both signs over the test points and EI / PI are of
    # order 0.1 … 1 there (with best = min(y) on a dense design both underflow and their gradient check would be vacuous)
    best = float(np.median(y))
    h = 2e-5
    for acq in (abo.ExpectedImprovement(0.01, best), abo.UpperConfidenceBound(2.0), abo.ProbabilityImprovement(0.01, best)):
        f, g = acquisition_value_and_grad(acq, m, Z)
        np.testing.assert_allclose(f, acq(m, Z), rtol=1e-9, atol=1e-10)   # the scored path's value (another summation order)
        oracle = _oracle_acq(acq, st)
        np.testing.assert_allclose(f, oracle(Z), rtol=1e-8, atol=1e-11)
        # fourth-order central differences of the ORACLE's acquisition (independent arithmetic), all coordinates in one batch
        # (near the data σ is small and EI / PI are steep functions of x: a second-order stencil's own error would show)
        pts = np.repeat(Z[:, None, :], 4 * d, axis=1)
        for c in range(d):
            for q, mult in enumerate((2.0, 1.0, -1.0, -2.0)):
                pts[:, 4 * c + q, c] += mult * h
        vals = oracle(pts.reshape(-1, d)).reshape(len(Z), 4 * d)
        fd = (-vals[:, 0::4] + 8.0 * vals[:, 1::4] - 8.0 * vals[:, 2::4] + vals[:, 3::4]) / (12.0 * h)
        # relative to the point's largest component, absolute (2e-8) below gradients of 1e-3: where PI has saturated at 1 its
        # gradient is ~1e-8 and the stencil's own rounding (ε·f/h ≈ 5e-12) is 1e-3 of that
        scale = np.maximum(np.max(np.abs(fd), axis=1, keepdims=True), 1e-3)
        # the check must not be vacuous: at least a quarter of the points carry gradients of order one (where μ is far above
        # best_y and σ is small, EI and PI underflow together with their gradients — those points test only that)
        assert np.percentile(np.max(np.abs(fd), axis=1), 50) > 1e-3
        err = float(np.max(np.abs(g - fd) / scale))
        check(f"refine/grad_fam{family}_d{d}_N{N}", f"{type(acq).__name__}_rel_vs_oracle_central_differences", err, 2e-5)


def test_gradient_on_the_small_variance_branch_and_at_training_points():
    """σ² ≤ 1e-12 → EI = PI = max(Δ, 0) (ExpectedImprovement.jl:44-46, ProbabilityImprovement.jl:42-44): the gradient there is
    −∇μ where Δ > 0 and 0 elsewhere; UCB's σ term drops at σ² ≤ 0"""
    X = np.array([[0.2], [0.5], [0.8]])
    y = np.array([0.3, -1.0, 0.4])
    m = abo.update(make_model(O.SE, 0.3, 1.0, 0.0), X, y)             # noise-free: σ² = 1e-18 at the data
    acq = abo.ExpectedImprovement(0.0, 0.0)
    f, g = acquisition_value_and_grad(acq, m, X)
    var = abo.posterior_var(m, X)
    assert np.all(var <= 1e-12)
    np.testing.assert_allclose(f, np.maximum(0.0 - y, 0.0), atol=1e-10)
    assert g[0, 0] == 0.0 and g[2, 0] == 0.0 and np.isfinite(g[1, 0])          # Δ ≤ 0 at the outer points, Δ = 1 in the middle
    fp, gp_ = acquisition_value_and_grad(abo.ProbabilityImprovement(0.0, 0.0), m, X)
    np.testing.assert_array_equal(fp, f)                                       # the same branch value for PI
    np.testing.assert_array_equal(gp_, g)


@pytest.mark.parametrize("family,d,N", [(O.MATERN52, 3, 60), (O.SE, 2, 100), (O.MATERN72, 6, 400)])
def test_refinement_against_scipy_on_the_oracle_and_against_the_finite_difference_loop(family, d, N):
    X, y = synth.standardized_problem(N, d, 0.02)
    ell, sf2, noise = 0.5, 1.0, 0.05                        # (noisy on purpose, as in the gradient test)
    m = abo.update(make_model(family, ell, sf2, noise), X, y)
    st = O.fit(family, ell, sf2, noise, 0.0, X, y)
    lower, upper = np.full(d, -0.5), np.full(d, 1.5)          # a box that reaches beyond the data: σ grows towards its faces
    best = float(np.median(y))               # EI / PI of order 0.1 … 1 over the box: something to climb (see the gradient test)
    for acq in (abo.UpperConfidenceBound(2.0), abo.ExpectedImprovement(0.01, best), abo.ProbabilityImprovement(0.01, best)):
        starts = synth.points(7, 16, d) * 2.0 - 0.5
        f0 = acq(m, starts)
        xr, fr, it = refine_starts(acq, m, starts, lower, upper, return_iters=True)
        assert np.all(fr >= f0 - 1e-15), "a refined start lost against its start"
        assert np.all(xr >= lower) and np.all(xr <= upper)
        np.testing.assert_allclose(acq(m, xr), fr, rtol=1e-9, atol=1e-10)       # the reported value is the score of the reported point
        assert np.all(it[:, 0] <= 100) and np.all(it[:, 1] <= 1 + (2 * 100 + 2) * 20)
        assert it[:, 1].sum() > 3 * len(starts) and np.median(fr - f0) > 1e-4, (type(acq).__name__, it[:, 1].sum(), np.median(fr - f0))
        oracle = _oracle_acq(acq, st)
        _against_scipy(f"refine/quality_fam{family}_d{d}_N{N}", type(acq).__name__, oracle,
                       lambda **kw: refine_starts(acq, m, starts, lower, upper, **kw), starts, xr, fr, lower, upper)
        xf, ff = _refine_starts_fd(acq, m, starts, lower, upper)
        # the analytic-gradient run is at least as good as the finite-difference one on (nearly) every start
        assert np.sum(fr >= ff - 1e-6 * np.maximum(1.0, np.abs(ff))) >= len(starts) - 2


def test_refinement_at_config_3_size_against_the_oracle_and_scipy():
    """The lockstep rounds at the size the headline is quoted on (N = 8192, d = 8, Matérn-5/2: split-k products over L⁻¹, compacted
    batches — the path DESIGN §3c times at 34 ms and no quality test reached before round 6): 8 starts inside the data's box; never
    below the start, inside the box, the reported value IS the oracle's acquisition at the reported point (1e-8), and the end points
    against SciPy L-BFGS-B on the oracle's acquisition from the same starts (acq_utils.jl:55-71) by the criteria of the small cases:
    no start ends below SciPy inside SciPy's basin; the best over the starts where both end at the same maximiser agrees to 1e-5.
    The best over ALL starts is recorded but barred loosely (0.05): the first run of this test (profiles/r06_refine_diag_c3.txt) has
    UCB's start 1 at another maximiser 0.76 away where SciPy is 0.67 higher, and starts 3 and 5 at other maximisers where the device
    is 1.46 and 1.16 higher — with 8 starts in 8 dimensions which optimiser's best-of-starts wins is decided by those, not by quality."""
    from tests.test_gpu_parity import c3_oracle
    X, y, st = c3_oracle()
    d = 8
    m = abo.update(make_model(O.MATERN52, 1.0, 1.0, 1e-3), X, y)
    lower, upper = np.zeros(d), np.ones(d)
    starts = synth.points(9, 8, d)
    case = "refine/quality_c3_N8192_d8"
    for acq in (abo.UpperConfidenceBound(2.0), abo.ExpectedImprovement(0.01, float(np.median(y)))):
        name = type(acq).__name__
        oracle = _oracle_acq(acq, st)
        f0 = oracle(starts)
        xr, fr, it = refine_starts(acq, m, starts, lower, upper, return_iters=True)
        assert np.all(xr >= lower) and np.all(xr <= upper)
        assert np.all(fr >= f0 - 1e-9 * np.maximum(1.0, np.abs(f0))), (name, "a refined start lost against its start")
        assert it[:, 1].sum() > 3 * len(starts) and np.max(fr - f0) > 1e-4, (name, it[:, 1].sum())
        fo = oracle(xr)
        check(case, f"{name}_reported_value_vs_oracle_at_reported_point", np.max(np.abs(fr - fo) / np.maximum(1.0, np.abs(fo))), 1e-8)
        _against_scipy(case, name, oracle, lambda **kw: refine_starts(acq, m, starts, lower, upper, **kw), starts, xr, fr, lower, upper,
                       all_starts_bar=0.05)


def test_refine_edge_cases():
    d = 2
    X, y = synth.standardized_problem(40, d, 0.02)
    m = abo.update(make_model(O.MATERN52, 0.4, 1.0, 1e-3), X, y)
    acq = abo.UpperConfidenceBound(2.0)
    # a degenerate box side (lower == upper), starts outside the box (clipped), a start on a corner
    lower, upper = np.array([0.3, 0.0]), np.array([0.3, 1.0])
    starts = np.array([[0.9, 0.5], [0.3, 0.0], [-4.0, 7.0]])
    xr, fr = refine_starts(acq, m, starts, lower, upper)
    assert np.all(xr[:, 0] == 0.3) and np.all(xr[:, 1] >= 0.0) and np.all(xr[:, 1] <= 1.0)
    np.testing.assert_allclose(acq(m, xr), fr, rtol=1e-9, atol=1e-10)
    clipped = np.clip(starts, lower, upper)
    assert np.all(fr >= acq(m, clipped) - 1e-10) and np.any(fr > acq(m, clipped) + 1e-3)    # the free coordinate does move
    # a non-finite start value is handed back as it came
    xn, fn = refine_starts(acq, m, np.array([[np.nan, 0.5]]), np.zeros(2), np.ones(2))
    assert np.isnan(fn[0])
    # wrong dimension / reversed bounds are refused, never a crash
    with pytest.raises(abo.DimensionMismatch):
        refine_starts(acq, m, np.zeros((2, 3)), np.zeros(3), np.ones(3))
    with pytest.raises(ValueError):
        refine_starts(acq, m, np.zeros((2, 2)), np.ones(2), np.zeros(2))
    # determinism: one workgroup per start, fixed-order reductions
    a = refine_starts(acq, m, synth.points(3, 50, d), np.zeros(2), np.ones(2))
    b = refine_starts(acq, m, synth.points(3, 50, d), np.zeros(2), np.ones(2))
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_one_call_optimize_acquisition_equals_its_parts_and_the_sharded_group():
    """abo_optimize_acquisition = device LHS → abo_acq top-k → abo_refine → arg-max; the group version shards grid and starts
    and must return the same bits"""
    from tests.test_gpu_multigpu import sharded
    d, N = 3, 120
    X, y = synth.standardized_problem(N, d, 0.02)
    dom = abo.ContinuousDomain(np.zeros(d), np.ones(d))
    m = abo.update(make_model(O.MATERN52, 0.5, 1.0, 1e-3), X, y)
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    best, val, sx, sv, rx, rv = abo.optimize_acquisition_device(acq, m, dom, n_grid=5000, n_local=40, seed=11, return_all=True)
    t = m.timings()
    assert t["refine_starts"] == 40 and t["refine_evals"] >= 40 and t["refine_ms"] > 0
    # parts: the same grid (same counter-based generator and seed), the same selection, the same refinement
    grid = abo.device_latin_hypercube(5000, dom.lower, dom.upper, 11)
    _, tv, ti = abo.evaluate(acq, m, grid, k=40, return_scores=False)
    np.testing.assert_array_equal(sv, tv.cpu().numpy())
    np.testing.assert_array_equal(sx, grid[ti].cpu().numpy())
    xr, fr = refine_starts(acq, m, sx, dom.lower, dom.upper)
    np.testing.assert_array_equal(rx, xr)
    np.testing.assert_array_equal(rv, fr)
    j = int(np.argmax(np.where(np.isfinite(rv), rv, -np.inf)))
    assert val == max(rv[j], sv[0]) and np.array_equal(best, rx[j] if rv[j] >= sv[0] else sx[0])
    assert val >= sv[0] and np.all(best >= 0) and np.all(best <= 1)
    np.testing.assert_allclose(acq(m, [best])[0], val, rtol=1e-9, atol=1e-10)
    # host API: optimize_acquisition(device_grid=True) is that one call
    rng = np.random.default_rng(5)
    seed = int(np.random.default_rng(5).integers(0, 2 ** 63))
    b2 = abo.optimize_acquisition(acq, m, dom, n_grid=5000, n_local=40, rng=rng, device_grid=True)
    np.testing.assert_array_equal(b2, abo.optimize_acquisition_device(acq, m, dom, 5000, 40, seed=seed))
    # two and three shards on this device
    for devs in ((0, 0), (0, 0, 0)):
        g = abo.update(sharded(O.MATERN52, 0.5, 1.0, 1e-3, devs), X, y)
        gb, gval, gsx, gsv, grx, grv = abo.optimize_acquisition_device(acq, g, dom, n_grid=5000, n_local=40, seed=11, return_all=True)
        np.testing.assert_array_equal(gsx, sx)
        np.testing.assert_array_equal(gsv, sv)
        np.testing.assert_array_equal(grx, rx)
        np.testing.assert_array_equal(grv, rv)
        np.testing.assert_array_equal(gb, best)
        assert gval == val


def test_refinement_on_an_appended_model_ignores_stale_rows():
    """rows / columns ≥ N of the shared factor storage may hold a discarded fantasy branch: the evaluation masks by the view's N"""
    d, N = 2, 126
    X, y = synth.standardized_problem(N + 4, d, 0.02)
    gp = make_model(O.MATERN52, 0.4, 1.0, 1e-3, n_max=N + 8)
    base = abo.update(gp, X[:N], y[:N])
    fant = abo.append(abo.append(abo.append(base, X[N], 3.0), X[N + 1], -2.0), X[N + 2], 1.0)   # crosses the 128-row block border
    del fant
    acq = abo.UpperConfidenceBound(2.0)
    starts = synth.points(9, 10, d)
    # (the same capacity: an N ≤ 128 model without spare capacity takes the one-launch fit, whose factor differs in the last bits)
    ref = abo.update(make_model(O.MATERN52, 0.4, 1.0, 1e-3, n_max=N + 8), X[:N], y[:N])
    xa, fa = refine_starts(acq, base, starts, np.zeros(d), np.ones(d))
    xb, fb = refine_starts(acq, ref, starts, np.zeros(d), np.ones(d))
    np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(fa, fb)


@pytest.mark.parametrize("family,d,N,S", [(O.MATERN52, 3, 300, 40), (O.SE, 2, 200, 130), (O.MATERN72, 8, 900, 17)])
def test_lockstep_variant_equals_the_one_launch_kernel_to_rounding(family, d, N, S, monkeypatch):
    """From 1024 factor rows on `abo_refine` advances the starts in lockstep rounds and batches a round's evaluations on the fp64
    MFMA tile core (L⁻¹ read once per round instead of once per start).  Same algorithm, another summation order in v and u:
    forced here at small N (ABO_REFINE_LOCKSTEP_NP), both variants must land on the same optimum value for every start (their
    iterates may differ in the last bits, and with them an occasional line-search decision), stay in the box, never lose."""
    X, y = synth.standardized_problem(N, d, 0.02)
    m = abo.update(make_model(family, 0.5, 1.0, 0.05, n_max=N + 8), X, y)
    lower, upper = np.full(d, -0.5), np.full(d, 1.5)
    best = float(np.median(y))
    for acq in (abo.UpperConfidenceBound(2.0), abo.ExpectedImprovement(0.01, best)):
        starts = synth.points(11, S, d) * 2.0 - 0.5
        starts[0, 0] = np.nan                                   # a non-finite start is handed back unchanged by both
        f0 = acq(m, starts)
        monkeypatch.setenv("ABO_REFINE_LOCKSTEP_NP", "0")
        xa, fa, ita = refine_starts(acq, m, starts, lower, upper, return_iters=True)
        monkeypatch.setenv("ABO_REFINE_LOCKSTEP_NP", "128")
        xb, fb, itb = refine_starts(acq, m, starts, lower, upper, return_iters=True)
        xb2, fb2 = refine_starts(acq, m, starts, lower, upper)
        np.testing.assert_array_equal(fb, fb2)                   # deterministic
        np.testing.assert_array_equal(xb, xb2)
        assert np.isnan(fa[0]) and np.isnan(fb[0])
        ok = np.isfinite(fa)
        assert np.all(fb[ok] >= f0[ok] - 1e-10) and np.all(xb[ok] >= lower) and np.all(xb[ok] <= upper)
        np.testing.assert_allclose(acq(m, xb[ok]), fb[ok], rtol=1e-9, atol=1e-10)
        close = np.abs(fa[ok] - fb[ok]) <= 1e-6 * np.maximum(1.0, np.abs(fa[ok]))
        assert close.mean() >= 0.9, (type(acq).__name__, close.mean())
        assert itb[ok, 1].sum() > 2 * ok.sum() and np.all(itb[:, 0] <= 100)
    # an appended view with a discarded fantasy branch behind it (stale rows ≥ N of the shared factor)
    Xn = synth.points(5, 3, d)
    fant = abo.append(abo.append(m, Xn[0], 2.0), Xn[1], -3.0)
    del fant
    acq = abo.UpperConfidenceBound(2.0)
    starts = synth.points(12, 9, d)
    xc, fc = refine_starts(acq, m, starts, lower, upper)
    monkeypatch.setenv("ABO_REFINE_LOCKSTEP_NP", "0")
    xd, fd = refine_starts(acq, m, starts, lower, upper)
    np.testing.assert_allclose(fc, fd, rtol=1e-6, atol=1e-8)


# ---- round 4: weighted-sum objectives (EnsembleAcquisition) and gradient-enhanced models on the device -----------------------------
def _fd4(fun, Z, h):
    """fourth-order central differences of fun over the points Z (all coordinates in one batch)"""
    n, d = Z.shape
    pts = np.repeat(Z[:, None, :], 4 * d, axis=1)
    for c in range(d):
        for q, mult in enumerate((2.0, 1.0, -1.0, -2.0)):
            pts[:, 4 * c + q, c] += mult * h
    vals = fun(pts.reshape(-1, d)).reshape(n, 4 * d)
    return (-vals[:, 0::4] + 8.0 * vals[:, 1::4] - 8.0 * vals[:, 2::4] + vals[:, 3::4]) / (12.0 * h)


def _no_host_loop(monkeypatch):
    """the finite-difference host loop must not be reached any more from the device paths"""
    import abstractbayesopt.jl_amd.acquisition as A

    def boom(*a, **k):
        raise AssertionError("_refine_starts_fd reached: the objective should have been served by abo_refine_terms")
    monkeypatch.setattr(A, "_refine_starts_fd", boom)


@pytest.mark.parametrize("family,d,N", [(O.MATERN52, 3, 200), (O.SE, 2, 1100)])
def test_ensemble_objective_gradient_and_refinement_on_the_device(family, d, N, monkeypatch):
    """EnsembleAcquisition (EnsembleAcq.jl:53-55) as ONE objective of the refinement stage: value = Σ wᵢ·acqᵢ on one posterior
    evaluation, ∇ = Σ wᵢ ∇acqᵢ — against the oracle's weighted sum and its central differences; refinement vs SciPy on the
    oracle; the one-launch kernel (N = 200) and the lockstep variant (N = 1100 ≥ 1024 rows)."""
    _no_host_loop(monkeypatch)
    X, y = synth.standardized_problem(N, d, 0.03)
    ell, sf2, noise = 0.7 * np.sqrt(d), 1.3, 0.1
    m = abo.update(make_model(family, ell, sf2, noise), X, y)
    st = O.fit(family, ell, sf2, noise, 0.0, X, y)
    best = float(np.median(y))
    members = [abo.ExpectedImprovement(0.01, best), abo.UpperConfidenceBound(2.0), abo.ProbabilityImprovement(0.05, best)]
    ens = abo.EnsembleAcquisition([0.5, 0.2, 0.3], members)
    nested = abo.EnsembleAcquisition([1.0, 1.0], [ens, abo.UpperConfidenceBound(0.5)])
    for acq in (ens, nested):
        def oracle(z, acq=acq):
            def val(a, w):
                if isinstance(a, abo.EnsembleAcquisition):
                    return sum(val(mm, w * wi) for wi, mm in zip(a.weights, a.acquisitions))
                return w * _oracle_acq(a, st)(z)
            return val(acq, 1.0)
        Z = synth.points(5, 20, d) * 2.0 - 0.5
        f, g = acquisition_value_and_grad(acq, m, Z)
        np.testing.assert_allclose(f, oracle(Z), rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(acq(m, Z), f, rtol=1e-9, atol=1e-10)             # abo_acq_terms: the scored value
        fd = _fd4(oracle, Z, 2e-5)
        scale = np.maximum(np.max(np.abs(fd), axis=1, keepdims=True), 1e-3)
        err = float(np.max(np.abs(g - fd) / scale))
        check(f"refine/ensemble_fam{family}_d{d}_N{N}", f"{'nested' if acq is nested else 'flat'}_grad_rel_vs_oracle_cd", err, 2e-5)
    lower, upper = np.full(d, -0.5), np.full(d, 1.5)
    starts = synth.points(7, 12, d) * 2.0 - 0.5
    f0 = ens(m, starts)
    xr, fr, it = refine_starts(ens, m, starts, lower, upper, return_iters=True)
    assert np.all(fr >= f0 - 1e-12) and np.all(xr >= lower) and np.all(xr <= upper)
    np.testing.assert_allclose(ens(m, xr), fr, rtol=1e-9, atol=1e-10)
    assert np.median(fr - f0) > 1e-4
    # against SciPy on the oracle's weighted sum (independent optimiser, independent arithmetic); the finite-difference host loop —
    # the same algorithm under the same rules on the library's own values — is compared separately
    _against_scipy(f"refine/quality_ensemble_fam{family}_d{d}_N{N}", "ensemble", lambda z: oracle(z, ens),
                   lambda **kw: refine_starts(ens, m, starts, lower, upper, **kw), starts, xr, fr, lower, upper)
    xf, ff = _refine_starts_fd(ens, m, starts, lower, upper)      # (the imported function itself: the patch guards the device paths)
    assert np.sum(fr >= ff - 1e-6 * np.maximum(1.0, np.abs(ff))) >= len(starts) - 2
    # the whole optimize_acquisition in one call, and the sharded group returns the same bits
    dom = abo.ContinuousDomain(lower, upper)
    b1, v1, sx, sv, rx, rv = abo.optimize_acquisition_device(ens, m, dom, n_grid=4000, n_local=20, seed=5, return_all=True)
    assert v1 >= sv[0] and np.all(b1 >= lower) and np.all(b1 <= upper)
    grid = abo.device_latin_hypercube(4000, lower, upper, 5)
    s_all = ens(m, grid).cpu().numpy()
    ov, oi = O.top_k(s_all, 20)
    np.testing.assert_array_equal(sv, ov)
    if N < 1024:
        from tests.test_gpu_multigpu import sharded
        g2 = abo.update(sharded(family, ell, sf2, noise, (0, 0)), X, y)
        b2, v2, *_ = abo.optimize_acquisition_device(ens, g2, dom, n_grid=4000, n_local=20, seed=5, return_all=True)
        np.testing.assert_array_equal(b2, b1)
        assert v2 == v1


def _grad_problem(N, d, seed=1):
    X = synth.points(seed, N, d)
    f = np.sin(2 * np.pi * X).sum(axis=1) / np.sqrt(d)
    g = 2 * np.pi * np.cos(2 * np.pi * X) / np.sqrt(d)
    return X, np.column_stack([f, g])


@pytest.mark.parametrize("family,d,N", [(O.MATERN52, 2, 40), (O.SE, 3, 60), (O.MATERN72, 5, 70)])
def test_gradient_enhanced_objective_gradients_against_the_oracle(family, d, N):
    """Value and gradient of EI / UCB / PI, GradientNormUCB (gradNormUCB.jl:43-51) and an ensemble of them on a gradient-enhanced
    handle — ∇μ = E[∇f] − m_∇ and ∇σ² = 2·Cov(f, ∇f) from ONE all-output posterior evaluation, GradientNormUCB by central
    differences on the device — against fourth-order central differences of oracle/grad_oracle.py's acquisition (≤ 2e-5)."""
    from oracle import grad_oracle as G
    from tests.test_gpu_gradient_gp import make_grad
    p = d + 1
    X, Y = _grad_problem(N, d)
    ell, sf2, noise = 0.6, 1.2, 0.05
    mean_c = np.concatenate([[0.1], np.linspace(-0.2, 0.3, d)])           # non-zero prior means of the gradient outputs on purpose
    m = abo.update(make_grad(family, ell, sf2, noise, p, mean_c), X, Y)
    st = G.fit(family, ell, sf2, noise, mean_c, X, Y)
    best = float(np.median(Y[:, 0]))
    Z = synth.points(5, 16, d) * 1.4 - 0.2

    def o_fv(acq):
        def f(z):
            mu, var = G.predict(st, z)
            if isinstance(acq, abo.UpperConfidenceBound):
                return O.upper_confidence_bound(mu, var, acq.beta)
            return O.acquisition(O.ACQ_EI if isinstance(acq, abo.ExpectedImprovement) else O.ACQ_PI, mu, var, acq.xi, acq.best_y)
        return f

    cases = [(a, o_fv(a)) for a in (abo.ExpectedImprovement(0.01, best), abo.UpperConfidenceBound(2.0),
                                    abo.ProbabilityImprovement(0.01, best))]
    gn = abo.GradientNormUCB(1.5)
    cases.append((gn, lambda z: G.grad_norm_ucb(st, z, 1.5)))
    ens = abo.EnsembleAcquisition([0.6, 0.4], [abo.UpperConfidenceBound(2.0), gn])
    cases.append((ens, lambda z: 0.6 * o_fv(abo.UpperConfidenceBound(2.0))(z) + 0.4 * G.grad_norm_ucb(st, z, 1.5)))
    for acq, oracle in cases:
        f, g = acquisition_value_and_grad(acq, m, Z)
        np.testing.assert_allclose(f, oracle(Z), rtol=1e-8, atol=1e-10)
        fd = _fd4(oracle, Z, 2e-5)
        scale = np.maximum(np.max(np.abs(fd), axis=1, keepdims=True), 1e-3)
        assert np.percentile(np.max(np.abs(fd), axis=1), 75) > 1e-3      # (EI / PI underflow where μ is far above best_y)
        err = float(np.max(np.abs(g - fd) / scale))
        check(f"refine/gradgp_fam{family}_d{d}_N{N}", f"{type(acq).__name__}_rel_vs_oracle_central_differences", err, 2e-5)


def test_gradient_enhanced_optimize_acquisition_beyond_32_inputs(monkeypatch):
    """optimize_acquisition on a GradientGP with d = 36 inputs (round 6: the slab generator serves d = 33 … 128; GradientGP.jl:617-639
    has no limit): one C-ABI call — grid, scores, starts, refinement in lockstep rounds on the all-output posterior (GradientNormUCB
    through its 2d-point stencils) — never below the best start, inside the box, the reported value is the ORACLE's acquisition at
    the reported point; the host loop is not reached."""
    from oracle import grad_oracle as G
    from tests.test_gpu_gradient_gp import make_grad
    _no_host_loop(monkeypatch)
    d, N = 36, 8
    X, Y = _grad_problem(N, d)
    ell, sf2, noise = 2.5, 1.0, 0.02
    m = abo.update(make_grad(O.MATERN52, ell, sf2, noise, d + 1), X, Y)
    st = G.fit(O.MATERN52, ell, sf2, noise, np.zeros(d + 1), X, Y)
    best = float(np.median(Y[:, 0]))
    lower, upper = np.zeros(d), np.ones(d)
    dom = abo.ContinuousDomain(lower, upper)

    def o_ei(z):
        mu, var = G.predict(st, z)
        return O.expected_improvement(mu, var, best, 0.01)

    for acq, oracle in ((abo.ExpectedImprovement(0.01, best), o_ei), (abo.GradientNormUCB(2.0), lambda z: G.grad_norm_ucb(st, z, 2.0))):
        b, v, sx, sv, rx, rv = abo.optimize_acquisition_device(acq, m, dom, n_grid=400, n_local=4, seed=5, return_all=True)
        assert v >= sv[0] - 1e-12 and np.all(b >= lower) and np.all(b <= upper) and np.all(rv >= sv - 1e-9)
        np.testing.assert_allclose(oracle(sx), sv, rtol=1e-8, atol=1e-10)          # the grid stage's scores are the oracle's
        np.testing.assert_allclose(oracle(b[None, :])[0], v, rtol=1e-7, atol=1e-9)
        assert m.timings()["refine_evals"] >= 4


@pytest.mark.parametrize("family,d,N", [(O.MATERN52, 2, 40), (O.SE, 3, 50)])
def test_gradient_enhanced_refinement_and_optimize_acquisition_on_the_device(family, d, N, monkeypatch):
    """optimize_acquisition(acqf, ::GradientGP, domain) — what the reference's tutorials run (gradNormUCB.jl:39-51 on a
    GradientGP) — as one C-ABI call: refinement never below the start, inside the box, ≥ SciPy L-BFGS-B on the oracle's
    acquisition; the finite-difference host loop is not reached; the sharded group returns the single handle's bits."""
    from oracle import grad_oracle as G
    from tests.test_gpu_gradient_gp import make_grad
    _no_host_loop(monkeypatch)
    p = d + 1
    X, Y = _grad_problem(N, d)
    ell, sf2, noise = 0.5, 1.0, 0.02
    m = abo.update(make_grad(family, ell, sf2, noise, p), X, Y)
    st = G.fit(family, ell, sf2, noise, np.zeros(p), X, Y)
    best = float(np.median(Y[:, 0]))
    lower, upper = np.full(d, -0.2), np.full(d, 1.2)
    dom = abo.ContinuousDomain(lower, upper)

    def o_ei(z):
        mu, var = G.predict(st, z)
        return O.expected_improvement(mu, var, best, 0.01)

    for acq, oracle in ((abo.ExpectedImprovement(0.01, best), o_ei), (abo.GradientNormUCB(2.0), lambda z: G.grad_norm_ucb(st, z, 2.0))):
        starts = synth.points(7, 14, d) * 1.4 - 0.2
        f0 = oracle(starts)
        xr, fr, it = refine_starts(acq, m, starts, lower, upper, return_iters=True)
        assert np.all(fr >= f0 - 1e-9) and np.all(xr >= lower) and np.all(xr <= upper)
        np.testing.assert_allclose(oracle(xr), fr, rtol=1e-8, atol=1e-9)          # the reported value is the oracle's value there
        assert it[:, 1].sum() > 2 * len(starts) and np.all(it[:, 0] <= 100)
        _against_scipy(f"refine/quality_gradgp_fam{family}_d{d}_N{N}", type(acq).__name__, oracle,
                       lambda **kw: refine_starts(acq, m, starts, lower, upper, **kw), starts, xr, fr, lower, upper)
        xf, ff = _refine_starts_fd(acq, m, starts, lower, upper)    # (the host loop itself, called directly: the same algorithm on stencils)
        assert np.sum(fr >= ff - 1e-6 * np.maximum(1.0, np.abs(ff))) >= len(starts) - 2
        # deterministic
        xr2, fr2 = refine_starts(acq, m, starts, lower, upper)
        np.testing.assert_array_equal(xr, xr2)
        np.testing.assert_array_equal(fr, fr2)
        b1, v1, sx, sv, rx, rv = abo.optimize_acquisition_device(acq, m, dom, n_grid=3000, n_local=16, seed=9, return_all=True)
        assert v1 >= sv[0] - 1e-12 and np.all(b1 >= lower) and np.all(b1 <= upper)
        np.testing.assert_allclose(oracle(b1[None, :])[0], v1, rtol=1e-8, atol=1e-9)
        grid = abo.device_latin_hypercube(3000, lower, upper, 9).cpu().numpy()
        ov, oi = O.top_k(oracle(grid), 16)
        np.testing.assert_allclose(sv, ov, rtol=1e-8, atol=1e-10)                  # the grid stage picked the oracle's starts
        # the host API reaches the same one call
        rng = np.random.default_rng(3)
        seed = int(np.random.default_rng(3).integers(0, 2 ** 63))
        b3 = abo.optimize_acquisition(acq, m, dom, n_grid=3000, n_local=16, rng=rng, device_grid=True)
        np.testing.assert_array_equal(b3, abo.optimize_acquisition_device(acq, m, dom, 3000, 16, seed=seed))
        grp = abo.update(abo.HipShardedGradientGP(sf2 * abo.with_lengthscale(FAMS_[family](), ell), p, noise, devices=(0, 0)), X, Y)
        b2, v2, *_ = abo.optimize_acquisition_device(acq, grp, dom, n_grid=3000, n_local=16, seed=9, return_all=True)
        np.testing.assert_array_equal(b2, b1)
        assert v2 == v1


def test_refinement_limits_are_clamped_not_overflowed():
    """abo_refine_opts with max_iter = INT32_MAX ("no limit") on a model that takes the lockstep rounds (≥ 1024 factor rows): the
    round budget max_iter × linesearch_max is computed in 64 bits and the fields are clamped — every start comes back refined"""
    d, N = 2, 1100
    X, y = synth.standardized_problem(N, d, 0.03)
    m = abo.update(make_model(O.SE, 0.5, 1.0, 0.05), X, y)
    acq = abo.UpperConfidenceBound(2.0)
    starts = synth.points(7, 9, d)
    ref_x, ref_f = refine_starts(acq, m, starts, np.zeros(d), np.ones(d))
    x, f, it = refine_starts(acq, m, starts, np.zeros(d), np.ones(d), max_iter=2 ** 31 - 1, return_iters=True)
    assert np.all(np.isfinite(f)) and np.all(f >= acq(m, starts) - 1e-12)
    np.testing.assert_allclose(acq(m, x), f, rtol=1e-9, atol=1e-10)
    assert np.all(f >= ref_f - 1e-9)                     # more iterations allowed: never worse than the default budget
    assert np.all(it[:, 0] <= 10000)


from tests.test_gpu_parity import FAMS as FAMS_  # noqa: E402


@pytest.mark.parametrize("family,d,N,S", [(O.MATERN52, 3, 700, 40), (O.SE, 2, 2100, 130)])
def test_split_k_form_of_the_lockstep_round_equals_the_plain_launch_to_rounding(family, d, N, S, monkeypatch):
    """From 2048 factor rows on a round's two products (128 starts × Np against L⁻¹: one row of output tiles, k up to Np) are cut
    into k-chunks — one workgroup per (tile, chunk) on the LDS-tiled MFMA core, partial products summed in chunk order
    (gemm.hip: GemmArgs::ksplit, splitk_reduce_kernel).  Same algorithm, another summation order: forced on (ABO_REFINE_KSPLIT)
    and off (= 0), the two forms must agree on every start's optimum to rounding; a chunk that is not a divisor of Np and a
    view with fewer valid rows than Np are part of the cases (N = 700 → Np = 768 with chunks of 256 and 128)."""
    X, y = synth.standardized_problem(N, d, 0.02)
    m = abo.update(make_model(family, 0.5, 1.0, 0.05), X, y)
    lower, upper = np.full(d, -0.5), np.full(d, 1.5)
    acq = abo.ExpectedImprovement(0.01, float(np.median(y)))
    starts = synth.points(11, S, d) * 2.0 - 0.5
    monkeypatch.setenv("ABO_REFINE_LOCKSTEP_NP", "128")
    monkeypatch.setenv("ABO_REFINE_KSPLIT", "0")
    xa, fa, ita = refine_starts(acq, m, starts, lower, upper, return_iters=True)
    for ks in ("256", "128"):
        monkeypatch.setenv("ABO_REFINE_KSPLIT", ks)
        xb, fb, itb = refine_starts(acq, m, starts, lower, upper, return_iters=True)
        xb2, fb2 = refine_starts(acq, m, starts, lower, upper)
        np.testing.assert_array_equal(fb, fb2)                   # deterministic: fixed-order sum of the partials
        np.testing.assert_array_equal(xb, xb2)
        assert np.all(fb >= acq(m, starts) - 1e-10) and np.all(xb >= lower) and np.all(xb <= upper)
        np.testing.assert_allclose(acq(m, xb), fb, rtol=1e-9, atol=1e-10)
        close = np.abs(fa - fb) <= 1e-6 * np.maximum(1.0, np.abs(fa))
        assert close.mean() >= 0.9, (ks, close.mean())
    monkeypatch.delenv("ABO_REFINE_KSPLIT")
    if N >= 2048:                                                # the default at this size IS the split-k form
        xc, fc = refine_starts(acq, m, starts, lower, upper)
        close = np.abs(fa - fc) <= 1e-6 * np.maximum(1.0, np.abs(fa))
        assert close.mean() >= 0.9


def test_boundary_maximisers_are_returned_on_the_bound_and_match_the_finite_difference_loop():
    """The reference's Fminbox keeps its iterates strictly inside the box (log barrier); this stage projects and may stop ON a
    bound — stated in include/abo_hip.h ("parity unpinned").  What must hold: a start whose ascent leaves the box ends on the
    face, its value is the score of that point, and the best refined value is at least the finite-difference host loop's best
    (the same projected L-BFGS under the same rules) within f_abstol — on a box small enough that most maxima sit on its faces."""
    d, N = 2, 150
    X, y = synth.standardized_problem(N, d, 0.02)
    m = abo.update(make_model(O.MATERN52, 0.5, 1.0, 0.05), X, y)
    lower, upper = np.full(d, 0.42), np.full(d, 0.58)        # a small box inside the data: UCB keeps rising towards some face
    starts = lower + (upper - lower) * synth.points(13, 24, d)
    for acq in (abo.UpperConfidenceBound(2.0), abo.ExpectedImprovement(0.01, float(np.median(y)))):
        xr, fr = refine_starts(acq, m, starts, lower, upper)
        assert np.all(xr >= lower) and np.all(xr <= upper)
        on_face = np.any((xr == lower) | (xr == upper), axis=1)
        assert on_face.sum() >= 3, on_face.sum()                # maximisers ON the bound are returned as such
        np.testing.assert_allclose(acq(m, xr), fr, rtol=1e-9, atol=1e-10)
        xf, ff = _refine_starts_fd(acq, m, starts, lower, upper)
        assert fr.max() >= ff.max() - 2.2e-9
        assert np.sum(fr >= ff - 1e-6 * np.maximum(1.0, np.abs(ff))) >= len(starts) - 2
