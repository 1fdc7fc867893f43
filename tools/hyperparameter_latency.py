"""Latency of the hyper-parameter path (parity-tested since round 2, never timed).

Reference: optimize_hyperparameters (src/bayesian_opt.jl:196-328) minimises nlml(params) (src/surrogates/StandardGP.jl:99-114) with
Fminbox(LBFGS) and `autodiff=:forward` — kernel matrix, Cholesky and solves on ForwardDiff duals in generic Julia, every objective
evaluation.  Here an objective evaluation is one refit (abo_fit) + abo_nlml_grad: K⁻¹ = L⁻ᵀL⁻¹ on the fp64 MFMA GEMM (N³/3 flop on the
lower tiles), then one sweep that generates ∂K/∂log ℓ tile by tile and reduces ½ tr((K⁻¹ − ααᵀ)∂K/∂θ).

    python tools/hyperparameter_latency.py [N ...]        (default 1024 4096 8192)
Prints per N: the phases of one objective evaluation (HIP events) with the K⁻¹ GEMM's fraction of the fp64 MFMA peak (78.6 TFLOP/s),
the host wall clock of value + gradient, and — N ≤ 4096 — one whole optimize_hyperparameters (2 parameters, 3 restarts).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import hyperparams, synth

PEAK = 78.6
sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096, 8192]
d = 8
print("one objective evaluation of the hyper-parameter search = refit + abo_nlml_grad (value and analytic gradient w.r.t. log ell, log sigma_f2)")
print("     N |  refit ms (K_XX  chol  L^-1  alpha) | K^-1 GEMM ms  TFLOP/s  of fp64 peak | trace sweep ms  Gpair/s | value+grad wall ms")
out = {}
for N in sizes:
    X, y = synth.standardized_problem(N, d, 0.03)
    gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 1.0), 1e-3)
    p = [0.0, 0.0]
    for _ in range(3):
        hyperparams.nlml_and_grad(gp, p, X, y)
    walls, tms = [], []
    for _ in range(7):
        t0 = time.perf_counter()
        k = abo.with_lengthscale(abo.Matern52Kernel(), 1.0)
        m = abo.update(abo.HipStandardGP(k, 1e-3), X, y)
        import ctypes as C
        v, d1, d2 = C.c_double(), C.c_double(), C.c_double()
        abo._lib.check(abo._lib.lib().abo_nlml_grad(m._require(), C.byref(v), C.byref(d1), C.byref(d2)))
        walls.append((time.perf_counter() - t0) * 1e3)
        tms.append(m.timings())
    med = {k: float(np.median([t[k] for t in tms])) for k in tms[0]}
    flop = N ** 3 / 3.0
    tf = flop / (med["nlml_kinv_ms"] * 1e-3) / 1e12 if med["nlml_kinv_ms"] > 0 else 0.0
    pairs = N * (N + 1) / 2 / (med["nlml_trace_ms"] * 1e-3) / 1e9 if med["nlml_trace_ms"] > 0 else 0.0
    print(f"{N:6d} | {med['fit_total_ms']:7.3f} ({med['fit_kernel_matrix_ms']:.3f} {med['fit_cholesky_ms']:.3f} {med['fit_inverse_ms']:.3f} "
          f"{med['fit_alpha_ms']:.3f}) | {med['nlml_kinv_ms']:9.3f}  {tf:7.1f}  {tf / PEAK:6.3f}       | {med['nlml_trace_ms']:9.3f}    {pairs:7.1f} |"
          f" {np.median(walls):8.3f}", flush=True)
    out[N] = (med, float(np.median(walls)))
print()
print("one whole optimize_hyperparameters (bayesian_opt.jl:196-328): 2 parameters (log ell, log scale), 3 restarts, L-BFGS-B on the host over "
      "(value, analytic gradient) from the device")
print("     N |  wall ms | objective evaluations | ms per evaluation | ell found  scale found")
for N in [n for n in sizes if n <= 4096]:
    X, y = synth.standardized_problem(N, d, 0.03)
    gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.7), 1e-3)
    dom = abo.ContinuousDomain(np.zeros(d), np.ones(d))
    calls = [0]
    real = hyperparams.nlml_and_grad

    def counted(*a, **k):
        calls[0] += 1
        return real(*a, **k)

    hyperparams.nlml_and_grad = counted
    try:
        for rep in range(2):                                  # first repetition warms the buffer pool
            calls[0] = 0
            t0 = time.perf_counter()
            new = hyperparams.optimize_hyperparameters(gp, X, y, [np.log(0.7), 0.0], num_restarts=3, domain=dom,
                                                       rng=np.random.default_rng(3))
            wall = (time.perf_counter() - t0) * 1e3
    finally:
        hyperparams.nlml_and_grad = real
    print(f"{N:6d} | {wall:8.1f} | {calls[0]:21d} | {wall / max(calls[0], 1):17.3f} | {abo.get_lengthscale(new)[0]:.4f}     {abo.get_scale(new)[0]:.4f}", flush=True)
