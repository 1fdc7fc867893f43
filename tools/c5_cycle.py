"""BASELINE config 5's BO loop over a whole refresh cycle: how often does a greedy q-EI batch have to rebuild its block (one pass over the
resident K_ZX), and what is the MEAN step?  (bench.py times 20 steps right behind a refresh, where every pick is still found in a
carried-over block; tools/c5_refresh_drift.py showed one rebuild per ~8 steps over 512.)

    python tools/c5_cycle.py [steps=512] [block sizes ...=16 32 64]
Same loop as bench.py's C5 leg and tools/c5_refresh_drift.py: d = 16, N = 16384, noisy Matern-5/2, 131 072 resident candidates, q = 8, the
first pick appended for real (observed with noise) each step, the grid down-dated from the batch's chain; no refresh inside the cycle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 512
blocks = [int(a) for a in sys.argv[2:]] or [16, 32, 64]
d, N, M, Q = 16, 16384, 131072, 8
ell, sf2, noise, xi = 2.0, 1.0, 1e-2, 0.01
X = synth.points(1, N, d)
y_raw = synth.objective(X, noise_std=float(np.sqrt(noise)))
y_mean, y_std = y_raw.mean(), y_raw.std(ddof=1)
y = (y_raw - y_mean) / y_std
Zd = torch.from_numpy(synth.points(2, M, d)).cuda()
print(f"C5 loop over {steps} steps without a refresh (N = {N}+, grid {M}, q = {Q}); per block size T: median / mean step, steps that rebuilt a block")
for T in blocks:
    abo._lib.check(abo._lib.lib().abo_set_qei_block(T))
    gp = abo.HipStandardGP(sf2 * abo.with_lengthscale(abo.Matern52Kernel(), ell), noise, n_max=N + steps)
    model = abo.update(gp, X, y)
    cands = abo.ResidentCandidates(model, Zd)
    best = float(y.min())
    rng = np.random.default_rng(7)
    ts, builds = [], 0
    for k in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = {}
        pts, idx, val, _ = abo.greedy_qei(model, cands, Q, xi, best, rollback=True, stats=stats)
        builds += int(stats.get('block_builds', 0))
        x_new = pts[0]
        y_new = float(((np.sin(2 * np.pi * x_new).sum() / np.sqrt(d) + np.sqrt(noise) * rng.standard_normal()) - y_mean) / y_std)
        model = abo.append(model, x_new, y_new)
        cands.downdate(model)
        ts.append((time.perf_counter() - t0) * 1e3)
        best = min(best, y_new)
    ts = np.asarray(ts)
    slow = ts > 2.5 * np.median(ts)
    print(f"T = {T:3d}: median {np.median(ts):.3f} ms, MEAN {ts.mean():.3f} ms, {int(slow.sum())} of {steps} steps rebuilt a block (abo_qei_stats: {builds} blocks built) "
          f"({np.median(ts[slow]) if slow.any() else float('nan'):.2f} ms each); first 20 steps mean {ts[:20].mean():.3f}, last 100 mean {ts[-100:].mean():.3f}", flush=True)
    del cands, model, gp
    abo._lib.lib().abo_pool_trim(0)
abo._lib.lib().abo_set_qei_block(0)
