"""Latency of one whole `optimize_acquisition` (acq_utils.jl:33-73) through ONE C-ABI call (abo_optimize_acquisition: device LHS
grid → fused scores → top n_local → on-device L-BFGS refinement of every start → best point), at the reference's stock shape
(n_grid = 10 000, n_local = 100, acq_utils.jl:37-38) and a few larger ones, next to the same stage done the host-driven way
(grid stage + batched finite-difference L-BFGS, round 2's `refine_starts`) and next to the grid stage alone.
usage: python tools/optimize_acquisition_latency.py > profiles/r04_optimize_acquisition_latency.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth
from abstractbayesopt.jl_amd.acquisition import _refine_starts_fd

print("optimize_acquisition in one C-ABI call (abo_optimize_acquisition), n_grid = 10000, n_local = 100, UCB(beta = 2), Matern-5/2,")
print("objective sum_c sin(3 x_c) + noise on N seeded points (a mid-run BO state: the acquisition surface has interior maxima to climb);")
print("refinement: one launch (one workgroup per start) below 1024 factor rows, lockstep rounds batched on the MFMA tile core from there on;")
print("median of 20 calls after 3 warm-ups; device = HIP events of the library stream (grid stage | refinement launch)")
SIZES = [(25, 1), (100, 2), (500, 4), (1024, 4), (2048, 8), (8192, 8)]
if len(sys.argv) > 1:                                   # e.g. `… 8192` under rocprofv3: that size only, no host-driven comparison
    SIZES = [(n, d) for n, d in SIZES if str(n) in sys.argv[1:]]
for N, d in SIZES:
    X = synth.points(1, N, d)
    y = np.sin(3 * X).sum(axis=1) + 0.01 * synth.normal(3, 0, N)
    y = (y - y.mean()) / y.std(ddof=1)
    dom = abo.ContinuousDomain(np.zeros(d), np.ones(d))
    gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.3 * np.sqrt(d)), 1e-4)
    m = abo.update(gp, X, y)
    acq = abo.UpperConfidenceBound(2.0)
    reps = 20 if N <= 2048 else 3
    wall, grid_ms, ref_ms, evals = [], [], [], []
    for r in range(3 + reps):
        t0 = time.perf_counter()
        best, val, sx, sv, rx, rv = abo.optimize_acquisition_device(acq, m, dom, 10_000, 100, seed=r, return_all=True)
        dt = (time.perf_counter() - t0) * 1e3
        if r >= 3:
            t = m.timings()
            wall.append(dt); grid_ms.append(t["acq_total_ms"]); ref_ms.append(t["refine_ms"]); evals.append(t["refine_evals"])
    gain = float(np.median(rv - sv))
    # host-driven variant: grid stage call + lockstep finite-difference L-BFGS (one C-ABI call per stencil / line-search trial)
    t0 = time.perf_counter()
    nrep = 0 if len(sys.argv) > 1 else (3 if N <= 2048 else 1)
    for r in range(nrep):
        grid = abo.device_latin_hypercube(10_000, dom.lower, dom.upper, r)
        _, tv, ti = abo.evaluate(acq, m, grid, k=100, return_scores=False)
        xs, fs = _refine_starts_fd(acq, m, grid[ti].cpu().numpy(), dom.lower, dom.upper)
    fd_ms = (time.perf_counter() - t0) * 1e3 / max(nrep, 1)
    print(f"N={N:5d} d={d}: one call {np.median(wall):8.3f} ms wall  (device: grid {np.median(grid_ms):7.3f} | refinement {np.median(ref_ms):8.3f} ms, "
          f"{int(np.median(evals))} evaluations over 100 starts = {np.median(ref_ms) * 1e3 / max(np.median(evals), 1):6.2f} us each if serial);  "
          f"host-driven finite-difference loop {fd_ms:8.2f} ms;  median gain over the start {gain:.3e}")
