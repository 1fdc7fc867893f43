"""Where the int8-residue engine overtakes the fp64 MFMA engine: one acquisition call (EI + top-100) at M = 262144 candidates,
d = 8 Matérn-5/2, for a range of training sizes, both engines (median of 5 calls after 2 warm-up calls)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth

M, d = 262144, 8
Zd = torch.from_numpy(synth.points(2, M, d)).cuda()
for N in (512, 768, 1024, 1280, 1536, 2048, 3072, 4096):
    X, y = synth.standardized_problem(N, d, 0.03)
    row = []
    for eng in ("fp64", "int8"):
        gp = abo.HipStandardGP(1.0 * abo.with_lengthscale(abo.Matern52Kernel(), 1.0), 1e-3, contraction=eng)
        m = abo.update(gp, X, y)
        acq = abo.ExpectedImprovement(0.01, float(y.min()))
        ts = []
        for r in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            abo.evaluate(acq, m, Zd, k=100, return_scores=False)
            ts.append((time.perf_counter() - t0) * 1e3)
        row.append(float(np.median(ts[2:])))
    print(f"N={N:5d}  fp64 {row[0]:8.2f} ms   int8 {row[1]:8.2f} ms   ratio {row[0] / row[1]:.2f}", flush=True)
