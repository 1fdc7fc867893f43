"""Builds tests/c_abi_harness.c (plain C, gcc) against libabo_hip.so and writes its fixture.  Test infrastructure."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi_harness.c")
OUT_DIR = os.path.join(ROOT, "tests", "_build")
BIN = os.path.join(OUT_DIR, "c_abi_harness")
LIB_DIR = os.path.join(ROOT, "abstractbayesopt.jl_amd", "lib")


def build(force=False):
    lib = os.path.join(LIB_DIR, "libabo_hip.so")
    deps = [SRC, lib, os.path.join(ROOT, "include", "abo_hip.h")]
    if not force and os.path.exists(BIN) and all(os.path.getmtime(BIN) >= os.path.getmtime(p) for p in deps):
        return BIN
    os.makedirs(OUT_DIR, exist_ok=True)
    # libabo_hip.so carries RUNPATH=/opt/rocm-*/lib and NEEDED libamdhip64.so.7: the harness itself names neither
    # the HIP runtime nor anything of PyTorch — exactly what `ccall((:abo_fit, "libabo_hip.so"), …)` sees
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), SRC, "-o", BIN,
           "-L", LIB_DIR, "-labo_hip", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath-link,/opt/rocm/lib", "-lm"]
    subprocess.check_call(cmd)
    return BIN


def _rec(f, name, arr):
    a = np.asarray(arr, dtype=np.float64).reshape(-1)
    f.write(f"{name} {a.size}\n")
    f.write(" ".join(f"{v:.17g}" for v in a) + "\n")


def write_fixture(path):
    """KAT-1,3,4,5,6 from tests/golden/kat.json (the reference's closed-form cases in 60-digit mpmath) and a seeded
    d = 4, N = 300, M = 20000 problem answered by the CPU oracle (scores, top-100, posterior after one append)."""
    from abstractbayesopt.jl_amd import synth
    from oracle import gp_oracle as O
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))
    with open(path, "w") as f:
        for name in ("kat1", "kat3", "kat4", "kat5", "kat6"):
            c = kat[name]
            _rec(f, f"{name}.hyper", [c["family"], c["ell"], c["sigma_f2"], c["noise_var"], c["mean_c"]])
            _rec(f, f"{name}.X", c["X"]); _rec(f, f"{name}.y", c["y"])
            if name == "kat6":
                continue
            _rec(f, f"{name}.Z", c["Z"]); _rec(f, f"{name}.mu", c["mu"]); _rec(f, f"{name}.var", c["var"])
            _rec(f, f"{name}.nlml", [c["nlml"]])
            if "ei" in c:
                _rec(f, f"{name}.acq", [c["xi"], c["best_y"], c["beta"]])
                _rec(f, f"{name}.ei", c["ei"]); _rec(f, f"{name}.ucb", c["ucb"]); _rec(f, f"{name}.pi", c["pi"])
        N, d, M, K = 300, 4, 20000, 100
        fam, ell, sf2, noise = O.MATERN52, 0.6, 1.0, 1e-3
        X, y = synth.standardized_problem(N, d, 0.02)
        Z = synth.points(2, M, d)
        st = O.fit(fam, ell, sf2, noise, 0.0, X, y)
        mu, var = O.predict(st, Z)
        best = float(y.min())
        scores = O.expected_improvement(mu, var, best, 0.01)
        _, top_idx = O.top_k(scores, K)
        _rec(f, "acq.hyper", [fam, ell, sf2, noise, 0.0])
        _rec(f, "acq.X", X); _rec(f, "acq.y", y); _rec(f, "acq.Z", Z)
        _rec(f, "acq.acq", [O.ACQ_EI, 0.01, best])
        _rec(f, "acq.scores", scores); _rec(f, "acq.top_idx", top_idx)
        x_new, y_new = Z[int(top_idx[0])], -0.3
        st2 = O.fit(fam, ell, sf2, noise, 0.0, np.vstack([X, x_new]), np.append(y, y_new))
        mu2, var2 = O.predict(st2, Z[:256])
        _rec(f, "acq.x_new", x_new); _rec(f, "acq.y_new", [y_new])
        _rec(f, "acq.mu_appended", mu2); _rec(f, "acq.var_appended", var2)
    return path
