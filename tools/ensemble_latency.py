"""PI / EnsembleAcquisition on the device — what a weighted-sum objective costs beside a single acquisition.
Reference: EnsembleAcquisition evaluates every member on the surrogate — one posterior_mean + posterior_var pair PER member and candidate
(src/acquisition_functions/EnsembleAcq.jl:53-55); here abo_acq_terms runs ONE posterior pass and every member's epilogue on it.

    python tools/ensemble_latency.py
Config-2 shape (N = 1024, d = 4, RBF, M = 65 536, top-100): one refit + scoring step with EI alone, PI alone, and the ensemble
0.5 EI + 0.3 UCB + 0.2 PI; device-resident inputs, median of 30 steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth

N, d, M = 1024, 4, 65536
X, y = synth.standardized_problem(N, d, 0.03)
dev = torch.device("cuda", 0)
Xd, yd, Zd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(synth.points(2, M, d)).to(dev)
gp = abo.HipStandardGP(abo.with_lengthscale(abo.SqExponentialKernel(), 0.5), 1e-3)
best = float(y.min())
acqs = {"ExpectedImprovement": abo.ExpectedImprovement(0.01, best),
        "ProbabilityImprovement": abo.ProbabilityImprovement(0.01, best),
        "UpperConfidenceBound": abo.UpperConfidenceBound(2.0),
        "Ensemble 0.5 EI + 0.3 UCB + 0.2 PI": abo.EnsembleAcquisition([0.5, 0.3, 0.2], [abo.ExpectedImprovement(0.01, best), abo.UpperConfidenceBound(2.0),
                                                                                         abo.ProbabilityImprovement(0.01, best)])}
print(f"N = {N}, d = {d}, M = {M}, top-100: refit + scores + selection per step (ms, median of 30), and the scoring call alone")
for name, acq in acqs.items():
    step, call = [], []
    for r in range(35):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = abo.update(gp, Xd, yd)
        t1 = time.perf_counter()
        abo.evaluate(acq, m, Zd, k=100, return_scores=False)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if r >= 5:
            step.append((t2 - t0) * 1e3); call.append((t2 - t1) * 1e3)
    print(f"{name:38s} step {np.median(step):.3f}   scoring call {np.median(call):.3f}", flush=True)
