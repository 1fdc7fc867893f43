"""Gradient-enhanced model: all-output posterior (posterior_grad_var) and per-point covariance blocks + GradientNormUCB
(posterior_grad_cov) on the fp64 kernels and on the int8-residue engine, same model, same candidates.
usage: python tools/grad_engine_latency.py [N d M]   (defaults 1000 8 4096 -> 9000 factor rows, 36864 candidate rows)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

abo = importlib.import_module("abstractbayesopt.jl_amd")
from abstractbayesopt.jl_amd import synth


def main():
    N, d, M = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1000, 8, 4096)
    p = d + 1
    X = synth.points(1, N, d)
    f = np.sin(2 * np.pi * X).sum(axis=1) / np.sqrt(d)
    g = 2 * np.pi * np.cos(2 * np.pi * X) / np.sqrt(d)
    Ys = np.column_stack([f, g])
    Z = synth.points(2, M, d)
    res = {}
    for eng in ("fp64", "int8"):
        m = abo.GradientGP(1.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.7), p, 1e-3, contraction=eng)
        m = abo.update(m, X, Ys)
        abo.posterior_grad_var(m, Z[:256]); abo.posterior_grad_cov(m, Z[:64])
        t = []
        for _ in range(3):
            t0 = time.perf_counter(); va = abo.posterior_grad_var(m, Z); t.append(time.perf_counter() - t0)
        tv = min(t) * 1e3
        t = []
        for _ in range(3):
            t0 = time.perf_counter(); mu, cv, sc = abo.posterior_grad_cov(m, Z, beta=2.0, return_all=True); t.append(time.perf_counter() - t0)
        tc = min(t) * 1e3
        res[eng] = (va, cv, sc)
        print(f"{eng}: N={N} d={d} rows={N * p} M={M} (x{p} outputs)  posterior_grad_var {tv:.1f} ms   posterior_grad_cov+GradientNormUCB {tc:.1f} ms"
              f"   engine={m.timings()['contraction_engine']}", flush=True)
    va8, cv8, sc8 = res["int8"]; va6, cv6, sc6 = res["fp64"]
    print(f"between engines: max|dvar|/prior {np.max(np.abs(va8 - va6)) / max(np.max(va6), 1e-300):.2e}  max|dcov| {np.max(np.abs(cv8 - cv6)):.2e}"
          f"  top-10 of the GradientNormUCB scores identical: {np.array_equal(np.argsort(-sc8)[:10], np.argsort(-sc6)[:10])}")


if __name__ == "__main__":
    main()
