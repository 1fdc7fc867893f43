"""Per-step latency of the path at small N (the reference's own 1-D / 2-D BO loops live here): refit + EI over M + top-100."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth
for N, d, M in [(25, 1, 10000), (100, 2, 10000), (500, 4, 65536), (1024, 4, 65536)]:
    X = synth.points(1, N, d); y = np.sin(X.sum(axis=1) * 3)
    Z = synth.points(2, M, d)
    Xd, yd, Zd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(Z).cuda()
    gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.5), 1e-6)
    acq = abo.ExpectedImprovement(0.0, float(y.min()))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            m = abo.update(gp, Xd, yd)
            abo.evaluate(acq, m, Zd, k=100, return_scores=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50 * 1e3
    t = m.timings()
    # the same step through the fused entry point (abo_fit_acq: one C-ABI call, one host synchronisation)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            m, _, _, _ = abo.update_and_evaluate(acq, gp, Xd, yd, Zd, k=100, return_scores=False, best_y=float(y.min()))
        torch.cuda.synchronize(); dtf = (time.perf_counter() - t0) / 50 * 1e3
    # host-array variant (reference API shape)
    t0 = time.perf_counter()
    for _ in range(50):
        m = abo.update(gp, X, y)
        s = acq(m, Z)
    dth = (time.perf_counter() - t0) / 50 * 1e3
    print(f"N={N} d={d} M={M}: device-resident {dt:.3f} ms/step (fit {t['fit_total_ms']:.3f} + acq {t['acq_total_ms']:.3f} on device; of the acq: "
          f"kernel values {t['acq_kxz_ms']:.3f}, contraction {t['acq_var_gemm_ms']:.3f}, epilogue {t['acq_finalize_ms']:.3f}, top-k {t['acq_topk_ms']:.3f}); "
          f"fused abo_fit_acq {dtf:.3f} ms/step; host arrays + scores back {dth:.3f} ms/step")
