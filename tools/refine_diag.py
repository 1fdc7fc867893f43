"""Why does a device refinement end below SciPy L-BFGS-B on the oracle's acquisition?  Per start: the device's value under the
reference's stopping rules (x_abstol 1e-4, f_abstol 2.2e-9, g_tol 1e-5), the value of a device run with the rules tightened
(x_abstol 1e-12, f_abstol 1e-300, g_tol 1e-9, 1000 iterations), SciPy's value from the same start, distances between the three
end points, iterations / evaluations.  Classes:
   same      device ≥ SciPy − tol
   early     device < SciPy − tol, the TIGHT device run reaches SciPy (≥ SciPy − tol): the stop rules ended the run early
   basin     device < SciPy − tol and so does the tight run, the end points are apart (> 1e-2): another local maximiser
   other     none of these
    python tools/refine_diag.py > gpurun_out/r05_refine_diag.txt"""
import json
import os
import sys

import numpy as np
from scipy.optimize import minimize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

import abstractbayesopt.jl_amd as abo  # noqa: E402
from abstractbayesopt.jl_amd import synth  # noqa: E402
from abstractbayesopt.jl_amd.acquisition import refine_starts  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402
from tests.test_gpu_parity import make_model  # noqa: E402
from tests.test_gpu_refine import _oracle_acq  # noqa: E402


def oracle_of(acq, st):
    if isinstance(acq, abo.EnsembleAcquisition):
        parts = [(w, oracle_of(a, st)) for w, a in zip(acq.weights, acq.acquisitions)]
        return lambda z: sum(w * f(z) for w, f in parts)
    return _oracle_acq(acq, st)


def diagnose(tag, acq, m, st, starts, lower, upper, out):
    oracle = oracle_of(acq, st)
    xr, fr, it = refine_starts(acq, m, starts, lower, upper, return_iters=True)
    xt, ft, itt = refine_starts(acq, m, starts, lower, upper, max_iter=1000, g_tol=1e-9, f_abstol=1e-300, x_abstol=1e-12, return_iters=True)
    rows = []
    for i in range(len(starts)):
        res = minimize(lambda z: -float(oracle(z[None, :])[0]), starts[i], method="L-BFGS-B", bounds=list(zip(lower, upper)),
                       options={"ftol": 1e-14, "gtol": 1e-8})
        fs = -res.fun
        tol = 1e-5 * max(1.0, abs(fs))
        d_ds = float(np.max(np.abs(xr[i] - res.x))); d_ts = float(np.max(np.abs(xt[i] - res.x)))
        if fr[i] >= fs - tol:
            cls = "same"
        elif ft[i] >= fs - tol:
            cls = "early"
        elif d_ts > 1e-2:
            cls = "basin"
        else:
            cls = "other"
        rows.append(dict(start=i, cls=cls, f_dev=float(fr[i]), f_tight=float(ft[i]), f_scipy=float(fs), gap=float(fr[i] - fs),
                         gap_tight=float(ft[i] - fs), dist_dev_scipy=d_ds, dist_tight_scipy=d_ts, iters=int(it[i, 0]), evals=int(it[i, 1]),
                         iters_tight=int(itt[i, 0]), evals_tight=int(itt[i, 1]), scipy_nit=int(res.nit)))
    cnt = {c: sum(r["cls"] == c for r in rows) for c in ("same", "early", "basin", "other")}
    print(f"== {tag}: {cnt}")
    for r in rows:
        if r["cls"] != "same":
            print("   ", json.dumps(r))
    out[tag] = {"counts": cnt, "rows": rows}


out = {}
if len(sys.argv) > 1 and sys.argv[1] == "c3":
    # the lockstep path at the headline's size (tests/test_gpu_refine.py: test_refinement_at_config_3_size_…)
    from tests.test_gpu_parity import c3_oracle
    X, y, st = c3_oracle()
    m = abo.update(make_model(O.MATERN52, 1.0, 1.0, 1e-3), X, y)
    lower, upper = np.zeros(8), np.ones(8)
    starts = synth.points(9, 8, 8)
    for acq in (abo.UpperConfidenceBound(2.0), abo.ExpectedImprovement(0.01, float(np.median(y)))):
        diagnose(f"c3/N8192_d8/{type(acq).__name__}", acq, m, st, starts, lower, upper, out)
        for r in out[f"c3/N8192_d8/{type(acq).__name__}"]["rows"]:
            print("   ", json.dumps(r))
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_refine_diag_c3.json"), "w"), indent=1)
    sys.exit(0)
for family, d, N in [(O.MATERN52, 3, 60), (O.SE, 2, 100), (O.MATERN72, 6, 400)]:
    X, y = synth.standardized_problem(N, d, 0.02)
    ell, sf2, noise = 0.5, 1.0, 0.05
    m = abo.update(make_model(family, ell, sf2, noise), X, y)
    st = O.fit(family, ell, sf2, noise, 0.0, X, y)
    lower, upper = np.full(d, -0.5), np.full(d, 1.5)
    best = float(np.median(y))
    starts = synth.points(7, 16, d) * 2.0 - 0.5
    for acq in (abo.UpperConfidenceBound(2.0), abo.ExpectedImprovement(0.01, best), abo.ProbabilityImprovement(0.01, best)):
        diagnose(f"single/fam{family}_d{d}_N{N}/{type(acq).__name__}", acq, m, st, starts, lower, upper, out)
for family, d, N in [(O.MATERN52, 3, 200), (O.SE, 2, 1100)]:
    X, y = synth.standardized_problem(N, d, 0.03)
    ell, sf2, noise = 0.7 * np.sqrt(d), 1.3, 0.1
    m = abo.update(make_model(family, ell, sf2, noise), X, y)
    st = O.fit(family, ell, sf2, noise, 0.0, X, y)
    best = float(np.median(y))
    ens = abo.EnsembleAcquisition([0.5, 0.2, 0.3], [abo.ExpectedImprovement(0.01, best), abo.UpperConfidenceBound(2.0),
                                                    abo.ProbabilityImprovement(0.05, best)])
    lower, upper = np.full(d, -0.5), np.full(d, 1.5)
    starts = synth.points(7, 12, d) * 2.0 - 0.5
    diagnose(f"ensemble/fam{family}_d{d}_N{N}", ens, m, st, starts, lower, upper, out)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_refine_diag.json"), "w"), indent=1)
