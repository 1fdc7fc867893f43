"""Does BASELINE config 5 need its periodic full refresh for NUMERICAL reasons?  (the refresh is 95 % of the amortised
step and the cadence — 16 — was a constant nobody measured.)

The BO loop of bench.py's C5 leg at its own size (d = 16, N = 16384, noisy Matérn-5/2, a resident grid of 131 072 candidates,
greedy q-EI q = 8, the first pick appended for real each step, the grid down-dated from the batch's chain) is run for `appends`
steps with NO refresh.  Every `every` appends the incrementally maintained state is compared with
  (o) an INDEPENDENT oracle refit on the N + k points (oracle/gp_oracle.py: host LAPACK): L on sampled rows (incl. every appended
      row), α, and (μ, σ²) of 1024 grid rows (first / middle / last of the grid);
  (r) the library's own refresh (full refit + re-evaluation of the WHOLE grid): max |Δμ|, |Δσ²| over all 131 072 candidates, and
      whether the top-100 of the grid's EI is the same list.
Reference: the reference always refits (src/surrogates/StandardGP.jl:79-83); with hyper-parameter optimisation on it re-optimises
every 10 iterations (src/bayesian_opt.jl:388), which refits anyway — the refresh stays for THAT; the question here is only whether
appends alone force one.

    python tools/c5_refresh_drift.py [appends=64] [every=16] [M=131072]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth
from oracle import gp_oracle as O          # the checker (this is a measurement tool, not the product)

appends = int(sys.argv[1]) if len(sys.argv) > 1 else 64
every = int(sys.argv[2]) if len(sys.argv) > 2 else 16
M = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
d, N, Q = 16, 16384, 8
ell, sf2, noise, xi = 2.0, 1.0, 1e-2, 0.01

X = synth.points(1, N, d)
y_raw = synth.objective(X, noise_std=float(np.sqrt(noise)))
y_mean, y_std = y_raw.mean(), y_raw.std(ddof=1)
y = (y_raw - y_mean) / y_std
Z = synth.points(2, M, d)
Zd = torch.from_numpy(Z).cuda()
gp = abo.HipStandardGP(sf2 * abo.with_lengthscale(abo.Matern52Kernel(), ell), noise, n_max=N + appends)
model = abo.update(gp, X, y)
cands = abo.ResidentCandidates(model, Zd)
best = float(y.min())
rows = np.concatenate([np.arange(342), np.arange(M // 2 - 170, M // 2 + 171), np.arange(M - 341, M)])
Xa, ya = X.copy(), y.copy()
rng = np.random.default_rng(7)
print(f"C5 drift without refresh: d={d} N={N} M={M} q={Q}, {appends} real appends (pick 1 of each greedy q-EI batch, observed with noise "
      f"sigma_n^2={noise}), compared every {every}", flush=True)
print("appends |  vs INDEPENDENT oracle refit on N+k points (1024 grid rows, sampled factor rows)      |  vs the library's own refresh (whole grid)")
print("        |  max|dL|     max|dalpha|/max|alpha|  max|dmu|     max|dvar|    | max|dmu|     max|dvar|    top-100 of EI the same  | ms: step  oracle-fit")
t_steps = []
for k in range(1, appends + 1):
    t0 = time.perf_counter()
    pts, idx, val, _ = abo.greedy_qei(model, cands, Q, xi, best, rollback=True)
    x_new = pts[0]
    y_new = float(((np.sin(2 * np.pi * x_new).sum() / np.sqrt(d) + np.sqrt(noise) * rng.standard_normal()) - y_mean) / y_std)
    model = abo.append(model, x_new, y_new)
    cands.downdate(model)
    t_steps.append((time.perf_counter() - t0) * 1e3)
    Xa = np.vstack([Xa, x_new]); ya = np.append(ya, y_new)
    best = min(best, y_new)
    if k % every and k != appends:
        continue
    mu_c, var_c = cands.mean_and_var()
    acq = abo.ExpectedImprovement(xi, best)
    _, tv_c, ti_c = cands.evaluate(acq, k=100)
    L, alpha, _ = abo.get_factor(model)
    # (r) the library's own refresh: a full refit on the N + k points and the whole grid re-evaluated — on a SECOND model and set, the
    # running ones are left alone (no refresh happens in the loop)
    ref = abo.update(abo.HipStandardGP(sf2 * abo.with_lengthscale(abo.Matern52Kernel(), ell), noise), Xa, ya)
    mu_r, var_r = abo.mean_and_var(ref, Zd)
    mu_r, var_r = mu_r.cpu().numpy() if hasattr(mu_r, "cpu") else mu_r, var_r.cpu().numpy() if hasattr(var_r, "cpu") else var_r
    _, tv_r, ti_r = abo.evaluate(acq, ref, Zd, k=100, return_scores=False)
    same_top = bool(np.array_equal(ti_r.cpu().numpy(), ti_c))
    del ref
    # (o) the independent oracle
    t1 = time.perf_counter()
    st = O.fit(O.MATERN52, ell, sf2, noise, 0.0, Xa, ya)
    t_fit = time.perf_counter() - t1
    srows = np.unique(np.concatenate([[0, 1, 127, 128, 4095, 8191, 12345, N - 1], np.arange(N, N + k)]))
    dL = float(np.max(np.abs(L[srows] - st.L[srows])))
    da = float(np.max(np.abs(alpha - st.alpha)) / np.max(np.abs(st.alpha)))
    mu_o, var_o = O.predict(st, Z[rows])
    print(f"{k:7d} |  {dL:.3e}    {da:.3e}               {np.max(np.abs(mu_c[rows] - mu_o)):.3e}    {np.max(np.abs(var_c[rows] - var_o)):.3e}    |"
          f" {np.max(np.abs(mu_c - mu_r)):.3e}    {np.max(np.abs(var_c - var_r)):.3e}    {same_top!s:5}                   |"
          f" {np.median(t_steps):6.2f}   {t_fit * 1e3:8.0f}", flush=True)
    del st, L
ts = np.asarray(t_steps)
slow = ts > 3.0 * np.median(ts)
print(f"median step (q-EI batch + real append + down-date, host wall clock, Python driver): {np.median(ts):.3f} ms;  MEAN over the {appends} "
      f"steps {ts.mean():.3f} ms — {int(slow.sum())} steps rebuilt the q-EI block (the chain of {64} conditioning columns was used up, or a pick "
      f"fell outside every block: one pass over K_ZX, {np.median(ts[slow]) if slow.any() else float('nan'):.2f} ms each)")
