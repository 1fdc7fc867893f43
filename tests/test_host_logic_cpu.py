"""CPU tests of the host-side mirror of the reference interface (no device work)."""
import numpy as np
import pytest

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import distributed as D
from abstractbayesopt.jl_amd import synth
from oracle import gp_oracle as O


def test_kernel_normal_form():
    # test/test_surrogates.jl:10-57: any kernel is reduced to ScaledKernel(inner ∘ ScaleTransform(1/ℓ), σ²)
    gp = abo.HipStandardGP(abo.SqExponentialKernel(), 0.1)
    assert abo.get_lengthscale(gp) == [1.0] and abo.get_scale(gp) == [1.0]
    gp = abo.HipStandardGP(abo.with_lengthscale(abo.SqExponentialKernel(), 2.0), 0.1)
    assert abo.get_lengthscale(gp) == [2.0] and abo.get_scale(gp) == [1.0]
    gp = abo.HipStandardGP(4.0 * abo.with_lengthscale(abo.Matern52Kernel(), 2.0), 0.1)
    assert abo.get_lengthscale(gp) == [2.0] and abo.get_scale(gp) == [4.0]
    gp = abo.HipStandardGP(3.0 * abo.Matern52Kernel(), 0.1, mean=abo.ConstMean(1.5))
    assert abo.get_lengthscale(gp) == [1.0] and abo.get_scale(gp) == [3.0] and gp.mean.c == 1.5
    assert gp.gpx is None and gp.noise_var == 0.1
    assert abo.get_kernel_constructor(gp) == abo.Matern52Kernel()
    assert abo.prep_input(gp, [1.0]) == [1.0] and abo.prep_output(gp, [2.0]) == [2.0]


def test_rescale_model_and_mean_std():
    # StandardGP.jl:164-232
    gp = abo.HipStandardGP(2.0 * abo.with_lengthscale(abo.SqExponentialKernel(), 0.5), 0.4, mean=abo.ConstMean(3.0))
    r = abo.rescale_model(gp, 2.0)
    assert abo.get_scale(r) == [0.5] and abo.get_lengthscale(r) == [0.5] and r.noise_var == 0.1 and r.mean.c == 1.5
    y = [1.0, 2.0, 4.0]
    m, s = abo.get_mean_std(gp, y, "mean_scale")
    assert abs(m - 7 / 3) < 1e-15 and abs(s - np.std(y, ddof=1)) < 1e-15
    assert abo.get_mean_std(gp, y, "scale_only")[0] == 0.0 and abo.get_mean_std(gp, y, "mean_only")[1] == 1.0
    np.testing.assert_allclose(abo.std_y(gp, y, m, s), (np.array(y) - m) / s)
    assert abo._get_minimum(gp, y) == 1.0


def test_acquisition_update_semantics():
    # test/test_acquisition.jl:45-63,:97-113: EI/PI take best_y = min(ys); UCB unchanged
    gp = abo.HipStandardGP(abo.SqExponentialKernel(), 0.1)
    ei = abo.update(abo.ExpectedImprovement(0.01, 10.0), [2.0, 1.0, 0.5], gp)
    assert ei.best_y == 0.5 and ei.xi == 0.01
    pi = abo.update(abo.ProbabilityImprovement(0.02, 10.0), [2.0, 1.0, 0.5], gp)
    assert pi.best_y == 0.5 and pi.xi == 0.02
    ucb = abo.UpperConfidenceBound(2.0)
    assert abo.update(ucb, [1.0], gp) == ucb
    assert abo.copy(ei) == ei and abo.copy(ei) is not ei


def test_continuous_domain_validation():
    # test/test_domains.jl:6-54
    d = abo.ContinuousDomain([0.0, -1.0], [1.0, 1.0])
    assert d.bounds == [(0.0, 1.0), (-1.0, 1.0)]
    with pytest.raises(ValueError):
        abo.ContinuousDomain([0.0], [1.0, 2.0])
    with pytest.raises(ValueError):
        abo.ContinuousDomain([2.0], [1.0])
    abo.ContinuousDomain([1.0], [1.0])          # degenerate box is allowed


def test_latin_hypercube_one_point_per_stratum():
    rng = np.random.default_rng(0)
    g = abo.latin_hypercube(1000, [0.0, -2.0, 5.0], [1.0, 2.0, 6.0], rng)
    assert g.shape == (1000, 3)
    for c, (lo, hi) in enumerate([(0.0, 1.0), (-2.0, 2.0), (5.0, 6.0)]):
        strata = np.floor((g[:, c] - lo) / (hi - lo) * 1000).astype(int)
        assert sorted(strata.tolist()) == list(range(1000))


def test_synth_is_counter_based():
    a = synth.points(2, 1000, 8)
    b = synth.points(2, 300, 8, first=200)
    np.testing.assert_array_equal(a[200:500], b)
    assert 0.0 <= a.min() and a.max() < 1.0 and abs(a.mean() - 0.5) < 0.02
    X, y = synth.standardized_problem(512, 4, 0.05)
    assert abs(y.mean()) < 1e-12 and abs(y.std(ddof=1) - 1.0) < 1e-12
    n = synth.normal(3, 0, 20000)
    assert abs(n.mean()) < 0.03 and abs(n.std() - 1.0) < 0.03


def test_shard_ranges_cover_exactly():
    for M in (0, 1, 7, 8, 1000, 8388608):
        for w in (1, 2, 3, 8):
            r = [D.shard_range(M, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == M
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_merge_topk_equals_global_stable_order():
    rng = np.random.default_rng(5)
    s = np.round(rng.normal(size=5000), 1)          # many ties
    s[[10, 4000]] = np.nan
    s[77] = np.inf
    k = 100
    vals, idx = [], []
    for r in range(4):
        lo, hi = D.shard_range(len(s), r, 4)
        v, i = O.top_k(s[lo:hi], k)
        vals.append(v); idx.append(i + lo)
    mv, mi = D.merge_topk(np.stack(vals), np.stack(idx), k)
    ov, oi = O.top_k(s, k)
    np.testing.assert_array_equal(mi, oi)
    np.testing.assert_array_equal(mv, ov)
    # short shards padded with (NaN, −1) are dropped
    mv, mi = D.merge_topk(np.array([[1.0, np.nan], [3.0, 2.0]]), np.array([[5, -1], [9, 7]]), 4)
    assert mi.tolist() == [9, 7, 5] and mv.tolist() == [3.0, 2.0, 1.0]


def test_input_containers():
    from abstractbayesopt.jl_amd.surrogate import as_points
    _, m, d, space, _ = as_points([0.0, 0.5, 1.0])
    assert (m, d, space) == (3, 1, 0)
    _, m, d, _, _ = as_points([[0.0, 1.0], [2.0, 3.0]])
    assert (m, d) == (2, 2)
    with pytest.raises(abo.DimensionMismatch):
        as_points([[0.0, 1.0], [2.0]])
    import torch
    _, m, d, space, _ = as_points(torch.zeros(4, 3, dtype=torch.float64))
    assert (m, d, space) == (4, 3, 0)
    with pytest.raises(TypeError):
        as_points(torch.zeros(4, 3, dtype=torch.float32))


def test_ensemble_acquisition_construction_and_update():
    # test/test_acquisition.jl:223-253 and :255-279
    ei, ucb = abo.ExpectedImprovement(0.01, 1.0), abo.UpperConfidenceBound(2.0)
    ens = abo.EnsembleAcquisition([0.5, 0.5], [ei, ucb])
    np.testing.assert_array_equal(ens.weights, [0.5, 0.5])
    ens = abo.EnsembleAcquisition([1.0, 3.0], [ei, ucb])
    np.testing.assert_allclose(ens.weights, [0.25, 0.75])
    with pytest.raises(AssertionError):
        abo.EnsembleAcquisition([1.0], [ei, ucb])
    with pytest.raises(AssertionError):
        abo.EnsembleAcquisition([-1.0, 2.0], [ei, ucb])
    with pytest.raises(AssertionError):
        abo.EnsembleAcquisition([0.0, 0.0], [ei, ucb])
    gp = abo.HipStandardGP(abo.SqExponentialKernel(), 0.1)
    up = abo.update(ens, [2.0, 1.5, 0.8], gp)
    np.testing.assert_array_equal(up.weights, ens.weights)
    assert up.acquisitions[0].best_y == 0.8 and up.acquisitions[1] is ucb
    c = abo.copy(ens)
    assert c == ens and c is not ens and c.acquisitions[0] is not ei


def test_lengthscale_bounds_reference_cases():
    # test/test_bayesian_opt.jl:419-456
    dom = abo.ContinuousDomain([-2.0], [2.0])
    lo, hi = abo.lengthscale_bounds([-1.0, 1.0], dom, min_frac=0.1, max_frac=2.0)
    assert len(lo) == 1 and len(hi) == 1
    assert abs(lo[0] - 0.1 * 2.0) < 1e-8 and abs(hi[0] - 2.0 * 4.0) < 1e-8
    dom2 = abo.ContinuousDomain([-2.0, -2.0], [2.0, 2.0])
    lo, hi = abo.lengthscale_bounds([[-1.0, -1.0], [1.0, 1.0]], dom2, min_frac=0.1, max_frac=1.0, n_samples=100_000,
                                    rng=np.random.default_rng(0))
    assert len(lo) == 2 and len(hi) == 2
    assert np.all(np.abs(lo - 0.1 * np.sqrt(10.0)) < 1e-2)
    with pytest.raises(abo.DimensionMismatch):
        abo.lengthscale_bounds([[0.0, 0.0, 0.0]], dom2)


def test_gradient_gp_host_helpers():
    """Host-side mirror of the GradientGP helpers (GradientGP.jl:617-639 ctor, :734-794 standardisation, :893-895
    prep_output) — no device needed for any of it."""
    from abstractbayesopt.jl_amd import gradient_gp as G
    gp = abo.GradientGP(2.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.5), 3, 0.04)
    assert gp.p == 3 and gp.gpx is None and np.array_equal(gp.mean.c, np.zeros(3))
    assert abo.get_lengthscale(gp) == [0.5] and abo.get_scale(gp) == [2.0]
    with pytest.raises(abo.DimensionMismatch):
        abo.GradientGP(abo.SqExponentialKernel(), 3, 0.1, mean=abo.gradConstMean([0.0, 0.0]))
    ys = np.array([[1.0, 0.1, -0.2], [3.0, 0.3, 0.4], [5.0, -0.5, 0.0]])
    # prep_output: all function values, then all ∂₁f, then all ∂₂f (MOInputIsotopicByOutputs order)
    np.testing.assert_array_equal(G.prep_output(gp, ys), [1.0, 3.0, 5.0, 0.1, 0.3, -0.5, -0.2, 0.4, 0.0])
    with pytest.raises(abo.DimensionMismatch):
        G.prep_output(gp, ys[:, :2])
    mu, sd = abo.get_mean_std(gp, ys, "mean_scale")
    np.testing.assert_allclose(mu, [3.0, 0.0, 0.0])                  # only the function value is centred
    np.testing.assert_allclose(sd, [2.0, 2.0, 2.0])                  # gradients share the function's scale
    mu_s, sd_s = abo.get_mean_std(gp, ys, "scale_only")
    np.testing.assert_allclose(mu_s, 0.0); np.testing.assert_allclose(sd_s, 2.0)
    mu_m, sd_m = abo.get_mean_std(gp, ys, "mean_only")
    np.testing.assert_allclose(mu_m, [3.0, 0.0, 0.0]); np.testing.assert_allclose(sd_m, 1.0)
    z = abo.std_y(gp, ys, mu, sd)
    np.testing.assert_allclose(z[:, 0], [-1.0, 0.0, 1.0]); np.testing.assert_allclose(z[:, 1], ys[:, 1] / 2.0)
    r = abo.rescale_model(abo.GradientGP(2.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.5), 3, 0.04,
                                         mean=abo.gradConstMean([1.0, 0.2, 0.0])), sd)
    assert isinstance(r, abo.GradientGP) and abo.get_scale(r) == [0.5] and abo.get_lengthscale(r) == [0.5]
    assert r.noise_var == pytest.approx(0.01) and np.allclose(r.mean.c, [0.5, 0.1, 0.0])
    assert abo._get_minimum(gp, ys) == 1.0
    u = abo._update_model_parameters(gp, 3.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.25))
    assert isinstance(u, abo.GradientGP) and abo.get_scale(u) == [3.0] and u.p == 3 and u.noise_var == 0.04
    acq = abo.GradientNormUCB(1.5)
    assert abo.update(acq, ys, gp) is acq and abo.copy(acq) == acq and abo.copy(acq) is not acq


def test_committed_bench_lines_keep_the_driver_contract():
    """The JSON lines under profiles/ are what bench.py printed on the GPU box: every key the driver and the judge
    read must be there with the right type (the contract in the task statement)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, bound in (("r03_bench_default.json", "mfma"), ("r03_c2_bench.json", "mfma"), ("r03_c5_bench.json", "hbm"),
                        ("r02_bench_default.json", "mfma"), ("r01_bench_default.json", "mfma")):
        j = json.load(open(os.path.join(root, "profiles", name)))
        assert j["metric"].startswith("GP-update+acq-eval ms per BO step") and j["unit"] == "ms"
        assert j["higher_is_better"] is False and j["scaling"] in ("weak", "strong") and j["vs_baseline"] is None
        assert j["dtype"] == "f64" and j["data"] == "synthetic" and j["n_gpus"] == 1
        assert isinstance(j["steps"], int) and isinstance(j["warmup"], int)
        assert j["value"] == j["ms_per_step"] and j["value"] > 0
        assert isinstance(j["config"]["workload"], str) and "model" not in j["config"]
        r = j["roofline"]
        assert r["bound"] == bound and r["unit"] in ("GB/s", "TFLOP/s", "TOP/s") and 0 < r["frac"] <= 1
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    j = json.load(open(os.path.join(root, "profiles", "r03_bench_default.json")))
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str) and "unit" in c
    assert c["repetitions"] >= 3 and len(c["all_values"]) == c["repetitions"]
    # the default line times all three single-GPU configurations: C3 (headline) + C2 and C5 as secondary entries, C5 with
    # its own roofline
    sec = j["secondary"]
    assert len(sec) == 3 and sec[0]["workload"].startswith("C2") and sec[1]["workload"].startswith("C5")
    assert sec[1]["roofline"]["bound"] == "hbm" and 0 < sec[1]["roofline"]["frac"] <= 1
    # round 3: the reference's own loop size and one whole optimize_acquisition in one call
    assert sec[2]["workload"].startswith("C1 shape") and 0 < sec[2]["fused_call_ms"] and sec[2]["optimize_acquisition"]["value"] > 0


def test_pmc_traffic_is_dropped_when_the_kernel_source_changed(tmp_path, monkeypatch):
    """bench.py reports the committed PMC traffic figure only while the kernel's source file still has the hash the PMC pass
    recorded; a stale figure becomes null instead of silently describing another kernel."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    traffic, src = b.pmc_traffic("c3", 16384)
    rec = json.load(open(os.path.join(root, src["file"])))                     # the newest committed pass (r03, r02, r01)
    # the guard hashes EVERY file the kernel is compiled from (bench.PMC_SOURCES), not just its translation unit
    assert b.PMC_SOURCES["c3"] == ["gemm.hip", "abo_kernels.h"] and "abo_oz_dev.h" in b.PMC_SOURCES["c3_int8"]
    if rec["kernel_source_sha"] == b.source_sha(b.PMC_SOURCES["c3"]):
        assert traffic == pytest.approx(rec["traffic_bytes_per_candidate"] * 16384) and not src.get("stale")
    else:
        assert traffic is None and src["stale"] is True
    monkeypatch.setattr(b, "source_sha", lambda names: "0" * 16)
    traffic, src = b.pmc_traffic("c3", 16384)
    assert traffic is None and src["stale"] is True


def test_flattening_an_objective_leaves_no_reference_cycle_on_the_surrogate():
    """`flatten_terms` runs on every acquisition call and refinement: if it closed over the surrogate in a recursive nested function
    (a function ↔ closure-cell cycle), a dropped model — hundreds of MB of device memory — would live on until the cyclic garbage
    collector happened to run (tools/soak.py caught exactly that as device memory growing to 12 GB)."""
    import gc
    import weakref

    import abstractbayesopt.jl_amd as abo
    from abstractbayesopt.jl_amd.acquisition import flatten_terms

    class Model:
        p = 3

    ens = abo.EnsembleAcquisition([1.0, 3.0], [abo.UpperConfidenceBound(2.0),
                                               abo.EnsembleAcquisition([1.0, 1.0], [abo.ExpectedImprovement(0.01, 0.5), abo.GradientNormUCB(1.5)])])
    was = gc.isenabled()
    gc.disable()
    try:
        m = Model()
        r = weakref.ref(m)
        terms = flatten_terms(ens, m)
        del m
        assert r() is None, "flatten_terms keeps the surrogate alive through a reference cycle"
    finally:
        if was:
            gc.enable()
    assert terms == [(1, 2.0, 0.0, 0.25), (0, 0.01, 0.5, 0.375), (4, 1.5, 0.0, 0.375)]
    assert flatten_terms(abo.GradientNormUCB(1.0), object()) is None              # needs a model with gradient outputs
    assert flatten_terms(abo.EnsembleAcquisition([1.0] * 9, [abo.UpperConfidenceBound(float(b)) for b in range(9)])) is None   # > 8 terms


def test_build_staleness_is_decided_by_content_not_by_time_stamps(monkeypatch, tmp_path):
    """A library file copied over lib/libabo_hip.so (an A/B build put back, older sources checked out) is newer than every source:
    a time-stamp test would keep it.  build() compares a manifest: sha256 over csrc/, the public header and the flags, and the sha256
    of the linked library file itself (round 5: a copied-in library with the old manifest beside it was still trusted)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_abo_build_t", os.path.join(root, "abstractbayesopt.jl_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    m0 = b._manifest()
    assert len(m0) == 64 and m0 == b._manifest()
    monkeypatch.setattr(b, "FLAGS", b.FLAGS + ["-DX"])
    assert b._manifest() != m0                                   # flags are part of what a library is built from
    monkeypatch.undo()
    lib = tmp_path / "libabo_hip.so"
    lib.write_bytes(b"not a library")
    monkeypatch.setattr(b, "LIB", str(lib))
    monkeypatch.setattr(b, "MANIFEST", str(tmp_path / "libabo_hip.manifest"))
    assert b._stale()                                            # a library without a manifest is not trusted
    (tmp_path / "libabo_hip.manifest").write_text("0" * 64 + "\n")
    assert b._stale()                                            # … nor one whose manifest names other sources
    (tmp_path / "libabo_hip.manifest").write_text(m0 + "\n")
    assert b._stale()                                            # … nor one that does not say which library file it describes
    (tmp_path / "libabo_hip.manifest").write_text(m0 + "\n" + b._lib_digest() + "\n")
    assert not b._stale()
    lib.write_bytes(b"another library copied over it")           # the motivating case: the file is replaced, the manifest stays
    assert b._stale()
