"""Fit-only loop for kernel traces of the factorisation: python tools/fit_only.py N d [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth
N, d = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
X, y = synth.standardized_problem(N, d, 0.03)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
noise = 1e-3 if N <= 8192 else 1e-2
gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 1.0 if d <= 8 else 2.0), noise)
for _ in range(3):
    m = abo.update(gp, Xd, yd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m = abo.update(gp, Xd, yd)
torch.cuda.synchronize()
print(f"N={N} d={d}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per fit (wall), phases {m.timings()}")
