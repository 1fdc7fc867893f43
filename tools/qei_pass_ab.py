"""A/B timing of the block pass of greedy q-EI (the one product over the resident K_ZX) at config-5 size: variants of the kernel are
selected per launch through ABO_QEI_PASS / ABO_QEI_PASS_SKINNY; prints pass_ms (HIP events) per variant and block size.
    python tools/qei_pass_ab.py [N] [M]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

import abstractbayesopt.jl_amd as abo  # noqa: E402
from abstractbayesopt.jl_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
d = 16
X = synth.points(1, N, d)
y = synth.objective(X, 0.1)
y = (y - y.mean()) / y.std(ddof=1)
Z = synth.points(2, M, d)
gp = abo.HipStandardGP(1.0 * abo.with_lengthscale(abo.Matern52Kernel(), 2.0), 1e-2, n_max=N + 64)
m = abo.update(gp, X, y)
c = abo.ResidentCandidates(m, Z)
best = float(y.min())
variants = [("pass", {}), ("reg", {"ABO_QEI_PASS_REG": "1"}), ("skinny", {"ABO_QEI_PASS_SKINNY": "1"})]
ref = {}
only = os.environ.get("QEI_AB_VARIANTS")
if only:
    variants = [v for v in variants if v[0] in only.split(",")]
Ts = [int(t) for t in os.environ.get("QEI_AB_T", "16,32,48,64").split(",")]
for T in Ts:
    for name, env in variants:
        for k in ("ABO_QEI_PASS_REG", "ABO_QEI_PASS_SKINNY"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ts = []
        for rep in range(5):
            pts, idx, val, st = c.qei(3, 0.01, best, block=T)
            ts.append(st["pass_ms"])
        key = (T,)
        if key not in ref:
            ref[key] = val
        dv = float(np.max(np.abs(val - ref[key])))
        print(f"T={T:2d} {name:8s} pass_ms min {min(ts):.3f} med {np.median(ts):.3f}  GB/s {st['pass_bytes'] / (min(ts) * 1e-3) / 1e9:7.0f}  "
              f"TF {st['pass_flop'] / (min(ts) * 1e-3) / 1e12:5.1f}  block_ms {st['block_ms']:.3f}  max|dEI| vs pass {dv:.2e}", flush=True)
