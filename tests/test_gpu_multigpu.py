"""GPU tests (-m gpu) of the in-library multi-device handle (abo_mgpu_*): sharding, merge and the
greedy q-EI exchange must equal the single-device path bit for bit.  The test box has one GPU: two or three shards
on device 0 exercise the sharding, the worker threads and the host exchange; one shard with ABO_MGPU_EXCHANGE=rccl
exercises ncclCommInitAll + ncclAllGather (RCCL refuses a communicator that lists one device twice)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import multigpu, synth
from oracle import gp_oracle as O

from tests.test_gpu_parity import FAMS, make_model


def sharded(family, ell, sf2, noise, devices, **kw):
    return abo.HipShardedGP(sf2 * abo.with_lengthscale(FAMS[family](), ell), noise, devices=devices, **kw)


@pytest.mark.parametrize("devices", [(0, 0), (0, 0, 0)])
@pytest.mark.parametrize("M", [5000, 2, 1])
def test_sharded_acquisition_equals_single_device_bit_for_bit(devices, M):
    d, N = 4, 300
    X, y = synth.standardized_problem(N, d, 0.02)
    Z = synth.points(2, M, d)
    one = abo.update(make_model(O.MATERN52, 0.6, 1.0, 1e-3), X, y)
    grp = abo.update(sharded(O.MATERN52, 0.6, 1.0, 1e-3, devices), X, y)
    assert grp.exchange() == "host"
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    k = 100
    s1, tv1, ti1 = abo.evaluate(acq, one, Z, k=k)
    s2, tv2, ti2 = abo.evaluate(acq, grp, Z, k=k)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(tv1, tv2)
    np.testing.assert_array_equal(ti1, ti2)
    ov, oi = O.top_k(s1, k)
    np.testing.assert_array_equal(ti2[:min(k, M)], oi)
    mu1, var1 = abo.mean_and_var(one, Z)
    mu2, var2 = abo.mean_and_var(grp, Z)
    np.testing.assert_array_equal(mu1, mu2)
    np.testing.assert_array_equal(var1, var2)
    # every shard holds the same factor (deterministic replicated fit)
    L0 = abo.get_factor(one)[0]
    for i in range(len(devices)):
        h = abo.surrogate._Handle(None)
        m = one._clone(h)
        h.ptr = grp.shard(i)
        try:
            np.testing.assert_array_equal(abo.get_factor(m)[0], L0)
        finally:
            h.ptr = None            # borrowed


def test_ties_resolve_to_the_lowest_global_index_across_shards():
    m = abo.update(sharded(O.SE, 0.01, 1.0, 1e-2, (0, 0, 0)), [[0.0, 0.0]], [0.0])
    Z = np.full((5000, 2), 50.0) + np.arange(5000)[:, None]
    s, tv, ti = abo.evaluate(abo.UpperConfidenceBound(2.0), m, Z, k=64)
    assert np.all(s == s[0])
    np.testing.assert_array_equal(ti, np.arange(64))
    Zn = Z[:300].copy()
    Zn[217, 0] = np.nan                     # NaN lives in the last shard and still sorts first
    s, tv, ti = abo.evaluate(abo.UpperConfidenceBound(2.0), m, Zn, k=4)
    assert ti[0] == 217 and np.isnan(tv[0]) and ti[1] == 0


def test_copy_and_refit_keep_value_semantics():
    d = 3
    X, y = synth.standardized_problem(80, d)
    Z = synth.points(2, 100, d)
    a = abo.update(sharded(O.SE, 0.7, 1.0, 1e-3, (0, 0)), X, y)
    mu_a = abo.posterior_mean(a, Z)
    c = abo.copy(a)
    b = abo.update(a, X[:40], y[:40])                 # a new model; `a` and its copy are untouched
    np.testing.assert_array_equal(abo.posterior_mean(a, Z), mu_a)
    np.testing.assert_array_equal(abo.posterior_mean(c, Z), mu_a)
    assert np.max(np.abs(abo.posterior_mean(b, Z) - mu_a)) > 1e-6
    with pytest.raises(abo.PosDefException) as e:
        abo.update(sharded(O.SE, 1.0, 1.0, 0.0, (0, 0)), [[-1.0, -1.0], [5.0, -5.0], [-1.0 + 1e-12, -1.0 + 1e-12]], [1.0, 2.0, 1.0])
    assert e.value.info == 3
    with pytest.raises(abo.DimensionMismatch):
        abo.posterior_mean(a, [[0.5]])


def test_device_generated_grid_stage_equals_single_device():
    d, N, n_grid, k = 3, 120, 10_000, 100
    X, y = synth.standardized_problem(N, d, 0.02)
    lower, upper = np.zeros(d) - 0.5, np.ones(d) * 1.5
    one = abo.update(make_model(O.MATERN52, 0.5, 1.0, 1e-3), X, y)
    grp = abo.update(sharded(O.MATERN52, 0.5, 1.0, 1e-3, (0, 0, 0)), X, y)
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    Zd = abo.device_latin_hypercube(n_grid, lower, upper, seed=11)
    _, tv1, ti1 = abo.evaluate(acq, one, Zd, k=k)
    tv2, ti2, tx2 = multigpu.grid_stage(acq, grp, lower, upper, n_grid=n_grid, n_local=k, seed=11)
    np.testing.assert_array_equal(tv1.cpu().numpy(), tv2)
    np.testing.assert_array_equal(ti1.cpu().numpy(), ti2)
    np.testing.assert_array_equal(Zd.cpu().numpy()[ti2], tx2)


def test_sharded_greedy_qei_and_append_equal_single_device():
    d, N0, M, q = 4, 200, 3000, 5
    X, y = synth.standardized_problem(N0, d, 0.05)
    Z = synth.points(2, M, d)
    best = float(y.min())
    fam, ell, sf2, noise = O.MATERN52, 0.6, 1.0, 1e-4
    one = abo.update(make_model(fam, ell, sf2, noise, n_max=N0 + 32), X, y)
    c1 = abo.ResidentCandidates(one, Z)
    c1.save()
    pts1, idx1, val1, mq = abo.greedy_qei(one, c1, q, 0.01, best)
    del mq
    c1.restore()
    grp = abo.update(sharded(fam, ell, sf2, noise, (0, 0, 0), n_max=N0 + 32), X, y)
    cg = abo.ShardedCandidates(grp, Z)
    pts2, idx2, val2 = cg.greedy_qei(grp, q, 0.01, best)
    np.testing.assert_array_equal(idx1, idx2)
    np.testing.assert_array_equal(val1, val2)
    np.testing.assert_array_equal(pts1, pts2)
    # model and stored posterior are unchanged by the exploration; then the real observation goes in on every device
    acq = abo.ExpectedImprovement(0.01, best)
    tv_b, ti_b = cg.evaluate(grp, acq, 10)
    _, tv_s, ti_s = c1.evaluate(acq, k=10)
    np.testing.assert_array_equal(ti_b, ti_s)
    np.testing.assert_array_equal(tv_b, tv_s)
    y_real = 0.37
    one2 = abo.append(one, pts1[0], y_real)
    c1.downdate(one2)
    grp2 = multigpu.append(grp, pts2[0], y_real, cg)
    _, tv_s, ti_s = c1.evaluate(acq, k=10)
    tv_b, ti_b = cg.evaluate(grp2, acq, 10)
    np.testing.assert_array_equal(ti_b, ti_s)
    np.testing.assert_array_equal(tv_b, tv_s)
    st = O.fit(fam, ell, sf2, noise, 0.0, np.vstack([X, pts1[0]]), np.append(y, y_real))
    mu_o, var_o = O.predict(st, Z[:500])
    mu_g, var_g = abo.mean_and_var(grp2, Z[:500])
    assert np.max(np.abs(mu_g - mu_o)) < 1e-9 and np.max(np.abs(var_g - var_o)) < 1e-9
    # the pre-append group is still the N0-point model (rollback is free)
    np.testing.assert_array_equal(abo.mean_and_var(grp, Z[:500])[0], abo.mean_and_var(one, Z[:500])[0])


def test_rccl_transport_at_world_size_one(monkeypatch):
    """ncclCommInitAll + ncclAllGather through the dlopen'ed librccl.so.1, one rank: same selection as abo_acq."""
    monkeypatch.setenv("ABO_MGPU_EXCHANGE", "rccl")
    d, N, M, k = 4, 300, 20000, 100
    X, y = synth.standardized_problem(N, d, 0.02)
    Z = synth.points(2, M, d)
    one = abo.update(make_model(O.SE, 0.5, 1.0, 1e-4), X, y)
    grp = abo.update(sharded(O.SE, 0.5, 1.0, 1e-4, (0,)), X, y)
    assert grp.exchange() == "rccl", grp.exchange_note()
    acq = abo.UpperConfidenceBound(2.0)
    s1, tv1, ti1 = abo.evaluate(acq, one, Z, k=k)
    s2, tv2, ti2 = abo.evaluate(acq, grp, Z, k=k)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(ti1, ti2)
    np.testing.assert_array_equal(tv1, tv2)
    cg = abo.ShardedCandidates(grp, Z[:3000])
    c1 = abo.ResidentCandidates(abo.update(make_model(O.SE, 0.5, 1.0, 1e-4, n_max=N + 16), X, y), Z[:3000])
    grp_n = abo.update(sharded(O.SE, 0.5, 1.0, 1e-4, (0,), n_max=N + 16), X, y)
    cg = abo.ShardedCandidates(grp_n, Z[:3000])
    p2, i2, v2 = cg.greedy_qei(grp_n, 3, 0.01, float(y.min()))
    c1.save()
    p1, i1, v1, _ = abo.greedy_qei(c1.model, c1, 3, 0.01, float(y.min()))
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_array_equal(v1, v2)


def test_eight_shards_config4_shape_at_reduced_size():
    """BASELINE config 4's structure — 8 shards, Matérn-5/2 d = 8, EI, top-100 — at a size the one-GPU box runs in a second
    (N = 1024, M = 100 000 not divisible by 8): merged selection and every score equal the single handle bit for bit."""
    d, N, M, k = 8, 1024, 100_003, 100
    X, y = synth.standardized_problem(N, d, 0.03)
    Z = synth.points(2, M, d)
    one = abo.update(make_model(O.MATERN52, 1.0, 1.0, 1e-3), X, y)
    grp = abo.update(sharded(O.MATERN52, 1.0, 1.0, 1e-3, (0,) * 8), X, y)
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    s1, tv1, ti1 = abo.evaluate(acq, one, Z, k=k)
    s2, tv2, ti2 = abo.evaluate(acq, grp, Z, k=k)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(ti1, ti2)
    np.testing.assert_array_equal(tv1, tv2)
    # selection only (no scores back), k larger than a shard's share of the winners
    _, tv3, ti3 = abo.evaluate(acq, grp, Z, k=2000, return_scores=False)
    ov, oi = O.top_k(s1, 2000)
    np.testing.assert_array_equal(ti3, oi)
    np.testing.assert_array_equal(tv3, ov)


def test_full_size_c4_eight_shards_on_one_device():
    """BASELINE config 4 at ITS OWN size through the 8-shard path (d = 8 Matérn-5/2, N = 8192, M = 8 388 608 = 8 × 2²⁰, EI,
    top-100; acq_utils.jl:50-52 reproduced globally): eight shards of the in-library multi-device handle, all on the one GPU
    of the test box (host exchange; > 1 RCCL rank needs more devices), against ONE handle scoring all 2²³ candidates — merged
    top-100 and every score bit for bit — and a 3 × 768 slice (first / a middle / the last shard) against an independent
    oracle refit (host LAPACK; shared with test_full_size_parity_c3 through tests.test_gpu_parity.c3_oracle)."""
    from tests.parity_record import check
    from tests.test_gpu_parity import c3_oracle
    N, d, M, k = 8192, 8, 1 << 23, 100
    ell, sf2, noise = 1.0, 1.0, 1e-3
    X, y, st = c3_oracle()
    Z = synth.points(2, M, d)                                    # the first 2²⁰ rows are config 3's candidates
    best = float(y.min())
    acq = abo.ExpectedImprovement(0.01, best)
    one = abo.update(make_model(O.MATERN52, ell, sf2, noise), X, y)
    s1, tv1, ti1 = abo.evaluate(acq, one, Z, k=k)
    t = one.timings()
    assert t["contraction_engine"] == abo._lib.CONTRACT_INT8 and t["oz_nmod"] == 14, t
    del one
    grp = abo.update(sharded(O.MATERN52, ell, sf2, noise, (0,) * 8), X, y)
    assert grp.exchange() == "host" and "listed twice" in grp.exchange_note()
    s8, tv8, ti8 = abo.evaluate(acq, grp, Z, k=k)
    for i in range(8):                                           # every shard ran the engine the metric is quoted on
        tt = abo._lib.AboTimings()
        abo._lib.check(abo._lib.lib().abo_get_timings(grp.shard(i), tt))
        assert tt.contraction_engine == abo._lib.CONTRACT_INT8 and tt.oz_nmod == 14
    np.testing.assert_array_equal(ti8, ti1)
    np.testing.assert_array_equal(tv8, tv1)
    np.testing.assert_array_equal(s8, s1)
    ov, oi = O.top_k(s8, k)                                      # the stable reverse sort of the scores themselves
    np.testing.assert_array_equal(ti8, oi)
    np.testing.assert_array_equal(tv8, ov)
    assert len(set(int(i) >> 20 for i in ti8)) > 1               # the winners come from more than one shard
    # selection only, no scores back (the benchmarked shape)
    _, tv9, ti9 = abo.evaluate(acq, grp, Z, k=k, return_scores=False)
    np.testing.assert_array_equal(ti9, ti1)
    np.testing.assert_array_equal(tv9, tv1)
    sl = np.concatenate([np.arange(0, 768), np.arange(M // 2 - 384, M // 2 + 384), np.arange(M - 768, M)])
    mu_o, var_o = O.predict(st, Z[sl])
    ei_o = O.expected_improvement(mu_o, var_o, best, 0.01)
    case = "c4/N8192_d8_M8388608_8shards"
    check(case, "ei_abs", np.max(np.abs(s8[sl] - ei_o)), 1e-9)
    mu8, var8 = abo.mean_and_var(grp, Z[sl])
    check(case, "mu", np.max(np.abs(mu8 - mu_o)) / max(1.0, np.max(np.abs(mu_o))), 1e-8)
    check(case, "var", np.max(np.abs(var8 - var_o)) / sf2, 1e-8)


def test_full_size_c5_sharded_greedy_qei(monkeypatch):
    """BASELINE config 5 across shards at its own size: d = 16, N = 16384, noisy, a resident grid of 131 072 candidates PER
    SHARD, greedy q-EI with q = 8 (per pick: EI + arg-max per shard, one exchange of the shards' pick records, the same
    fantasy append + O(N·M) down-date on every shard), then the real observation appended with the grid's down-date.  Four
    shards share the one GPU of the test box (eight would need 8 × (17 GB K_ZX + 6 GB factor + 15 GB int8 scratch) > 288 GB);
    the reference is ONE handle over all 524 288 candidates: same picks, same EI values, same grid posterior, bit for bit."""
    monkeypatch.setenv("ABO_CAND_KZX_GIB", "80")                 # the single handle's K_ZX (69 GB) stays resident like the shards'
    d, N, G, q = 16, 16384, 4, 8
    M = G * 131072
    ell, sf2, noise = 2.0, 1.0, 1e-2
    X = synth.points(1, N, d)
    y = synth.objective(X, 0.1)
    y = (y - y.mean()) / y.std(ddof=1)
    Z = synth.points(2, M, d)
    best = float(y.min())
    one = abo.update(make_model(O.MATERN52, ell, sf2, noise, n_max=N + 64), X, y)
    c1 = abo.ResidentCandidates(one, Z)
    c1.save()
    pts1, idx1, val1, mq = abo.greedy_qei(one, c1, q, 0.01, best)
    del mq
    c1.restore()
    y_real = 0.25
    one2 = abo.append(one, pts1[0], y_real)
    c1.downdate(one2)
    mu1, var1 = c1.mean_and_var()
    acq = abo.ExpectedImprovement(0.01, best)
    _, tv1, ti1 = c1.evaluate(acq, k=100)
    del c1, one, one2
    abo._lib.lib().abo_pool_trim(0)
    grp = abo.update(sharded(O.MATERN52, ell, sf2, noise, (0,) * G, n_max=N + 64), X, y)
    cg = abo.ShardedCandidates(grp, Z)
    pts2, idx2, val2 = cg.greedy_qei(grp, q, 0.01, best)
    np.testing.assert_array_equal(idx2, idx1)
    np.testing.assert_array_equal(val2, val1)
    np.testing.assert_array_equal(pts2, pts1)
    grp2 = multigpu.append(grp, pts2[0], y_real, cg)
    tv2, ti2 = cg.evaluate(grp2, acq, 100)
    np.testing.assert_array_equal(ti2, ti1)
    np.testing.assert_array_equal(tv2, tv1)
    assert len(set(int(i) // 131072 for i in ti2)) > 1           # the merged top-100 draws on more than one shard
    # the down-dated posterior of every shard's grid equals the single handle's
    mu2, var2 = cg.mean_and_var(grp2)
    np.testing.assert_array_equal(mu2, mu1)
    np.testing.assert_array_equal(var2, var1)


@pytest.mark.parametrize("mode", ["collective", "stall"])
def test_collective_faults_end_in_a_status_or_the_host_exchange_never_in_a_hang(mode):
    """csrc/mgpu.hip: exchange() = vote → bounded all-gather → fall-back.  A shard that is not ready makes the call return a
    status before anything is enqueued; a shard that fails AFTER the vote (or whose collective never completes: `stall`, a
    stand-in kernel that waits for a peer that never arrives) raises the abort word, every shard leaves its bounded wait,
    the communicators are aborted and the SAME call completes through the host exchange with the right selection.  Run as a
    child process with a time limit: the regression this guards against is a hang (and an aborted communicator set stays
    aborted for the rest of its process)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("ABO_MGPU_FAULT", "ABO_MGPU_TIMEOUT_MS")}
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "mgpu_fault_child.py"), mode], env=env, capture_output=True,
                       text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["transport_before"] == "rccl", out
    assert out["clean_call_equal"]
    assert "not ready" in out["ready_fault"] and out["ready_fault_s"] < 10.0
    assert out["transport_after_ready_fault"] == "rccl" and out["call_after_ready_fault_equal"]
    assert out.get("faulted_call_equal") is True, out           # the faulted call itself returned the right selection …
    assert out["faulted_call_s"] < 30.0                          # … in bounded time (stall: ABO_MGPU_TIMEOUT_MS = 1500)
    assert out["transport_after_fault"] == "host"
    assert ("timed out" if mode == "stall" else "aborted") in out["note_after_fault"], out["note_after_fault"]
    assert out["call_after_fault_equal"] and len(out["qei_after_fault"]) == 3
    assert out["new_group_transport"] == "host"


def test_group_errors_are_statuses_and_leave_the_group_usable():
    d = 2
    X = np.array([[-1.0, -1.0], [5.0, -5.0]])
    grp = abo.update(sharded(O.SE, 1.0, 1.0, 0.0, (0, 0), n_max=16), X, [1.0, 2.0])
    Z = synth.points(2, 300, d)
    cg = abo.ShardedCandidates(grp, Z)
    mu0 = abo.posterior_mean(grp, Z)
    with pytest.raises(abo.PosDefException) as e:                       # every shard refuses the duplicate point
        multigpu.append(grp, [-1.0 + 1e-12, -1.0 + 1e-12], 1.0, cg)
    assert e.value.info == 3
    with pytest.raises(abo.DimensionMismatch):
        multigpu.append(grp, [1.0, 2.0, 3.0], 0.0)
    with pytest.raises(abo.DimensionMismatch):
        abo.evaluate(abo.UpperConfidenceBound(2.0), grp, np.zeros((5, 3)), k=2)
    with pytest.raises(ValueError):
        abo.evaluate(abo.UpperConfidenceBound(2.0), sharded(O.SE, 1.0, 1.0, 0.1, (0, 0)), Z, k=2)     # gpx === nothing
    np.testing.assert_array_equal(abo.posterior_mean(grp, Z), mu0)      # the group and its grid are as before
    tv, ti = cg.evaluate(grp, abo.UpperConfidenceBound(2.0), 5)
    s, tv1, ti1 = abo.evaluate(abo.UpperConfidenceBound(2.0), grp, Z, k=5)
    np.testing.assert_array_equal(ti, ti1)
    ok = multigpu.append(grp, [2.0, 2.0], 0.5, cg)
    assert abs(abo.posterior_mean(ok, [[2.0, 2.0]])[0] - 0.5) < 1e-9


@pytest.mark.parametrize("config", ["c2", "c5"])
def test_bench_plain_multi_gpu_launch_runs_the_library_path(config):
    """`python bench.py --gpus 2` exactly as the driver launches N = 1 (no torchrun): one process, two shards through
    abo_mgpu_*.  On the one-GPU test box the shards share device 0 (--share-device), so the exchange is the host one
    and the line must say so; on a multi-GPU node the same command without the flag runs RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", config,
           "--share-device"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["devices"] == [0, 0] and cfg["exchange"] == "host" and "listed twice" in cfg["exchange_note"]
    assert cfg["rccl_ranks"] == 0 and "rehearsal" in out
    if config == "c2":
        assert len(out["per_device_hip_event_ms_per_step"]) == 2 and out["roofline"]["frac"] > 0
        assert cfg["M_total"] == 2 * cfg["M_per_gpu"]


def _grad_problem(N, d, seed=1):
    X = synth.points(seed, N, d)
    f = np.sin(2 * np.pi * X).sum(axis=1) / np.sqrt(d)
    g = 2 * np.pi * np.cos(2 * np.pi * X) / np.sqrt(d)
    return X, np.column_stack([f, g])


@pytest.mark.parametrize("devices", [(0, 0), (0, 0, 0)])
def test_gradient_enhanced_group_equals_the_single_device_model(devices):
    """abo_mgpu_create_grad / abo_mgpu_append_grad / abo_mgpu_cand_qei on a gradient-enhanced model (GradientGP.jl:617-668):
    fit, function-value posterior, acquisition + merged top-k, an appended observation with its grid down-date and the greedy
    q-EI picks (fantasy = posterior mean of all p outputs) equal the single-device HipGradientGP bit for bit; the q-EI model
    equals a from-scratch refit on the picked points with those fantasy observations (oracle/grad_oracle.py)."""
    from oracle import grad_oracle as G
    from tests.test_gpu_gradient_gp import make_grad
    d, N, M, q = 2, 60, 1500, 3
    p = d + 1
    X, Y = _grad_problem(N + 1, d)
    Z = synth.points(2, M, d)
    fam, ell, sf2, noise = O.MATERN52, 0.45, 1.2, 1e-3
    one = abo.update(make_grad(fam, ell, sf2, noise, p, n_max=N + 8), X[:N], Y[:N])
    grp = abo.update(abo.HipShardedGradientGP(sf2 * abo.with_lengthscale(FAMS[fam](), ell), p, noise, devices=devices, n_max=N + 8),
                     X[:N], Y[:N])
    mu1, var1 = abo.mean_and_var(one, Z)
    mu2, var2 = abo.mean_and_var(grp, Z)
    np.testing.assert_array_equal(mu1, mu2)
    np.testing.assert_array_equal(var1, var2)
    st = G.fit(fam, ell, sf2, noise, np.zeros(p), X[:N], Y[:N])
    mo, vo = G.predict(st, Z[:200])
    assert np.max(np.abs(mu2[:200] - mo)) < 1e-8 and np.max(np.abs(var2[:200] - vo)) < 1e-8
    best = float(Y[:N, 0].min())
    acq = abo.ExpectedImprovement(0.01, best)
    s1, tv1, ti1 = abo.evaluate(acq, one, Z, k=32)
    s2, tv2, ti2 = abo.evaluate(acq, grp, Z, k=32)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(ti1, ti2)
    # one real observation appended on every device, the sharded grid down-dated in the same call
    c1 = abo.ResidentCandidates(one, Z)
    c2 = abo.ShardedCandidates(grp, Z)
    one2 = abo.append(one, X[N], Y[N])
    c1.downdate(one2)
    grp2 = multigpu.append(grp, X[N], Y[N], c2)
    np.testing.assert_array_equal(abo.mean_and_var(one2, Z[:300])[1], abo.mean_and_var(grp2, Z[:300])[1])
    tvc1 = c1.evaluate(acq, k=8)
    tvc2 = c2.evaluate(grp2, acq, 8)
    np.testing.assert_array_equal(tvc1[1], tvc2[0])
    np.testing.assert_array_equal(tvc1[2], tvc2[1])
    with pytest.raises(ValueError):                       # the scalar append is refused for a gradient-enhanced group
        abo._lib.check(abo._lib.lib().abo_mgpu_append(grp2._g.ptr, X[N].ctypes.data, d, 0.5, None, None))
    # greedy q-EI: the library's sharded loop against the host loop on the single-device model
    c1.save()
    pts1, idx1, val1, mq = abo.greedy_qei(one2, c1, q, 0.01, best)
    pts2, idx2, val2 = c2.greedy_qei(grp2, q, 0.01, best)
    np.testing.assert_array_equal(idx1, idx2)
    np.testing.assert_array_equal(pts1, pts2)
    np.testing.assert_allclose(val1, val2, rtol=0, atol=1e-15)
    # … and against a from-scratch refit that conditions on the fantasy observations one after the other
    Xf, Yf = X.copy(), Y.copy()
    cur = abo.update(make_grad(fam, ell, sf2, noise, p), Xf, Yf)
    for j in range(q):
        yj = np.asarray(abo.posterior_grad_mean(cur, pts1[j][None, :])).reshape(-1)
        Xf, Yf = np.vstack([Xf, pts1[j]]), np.vstack([Yf, yj])
        cur = abo.update(make_grad(fam, ell, sf2, noise, p), Xf, Yf)
    mq_mu, mq_var = abo.mean_and_var(mq, Z[:300])
    rf_mu, rf_var = abo.mean_and_var(cur, Z[:300])
    assert np.max(np.abs(mq_mu - rf_mu)) < 1e-8 and np.max(np.abs(mq_var - rf_var)) < 1e-8
    # the group's model and grid are as before the q-EI call
    np.testing.assert_array_equal(c2.evaluate(grp2, acq, 8)[1], tvc2[1])
