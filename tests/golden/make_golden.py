#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ in 60-digit mpmath arithmetic.

The reference (Julia) cannot run in the build container and ships no fixture files, so the pins
are the closed-form identities its own tests assert:

  KAT-1  test/test_surrogates.jl:67-104    μ, σ² = k_xXᵀ(K̃\\y), k(x,x) − k_xXᵀ(K̃\\k_xX)   atol 1e-10
  KAT-2  test/test_surrogates.jl:151-169   NLML = ½(yᵀK̃⁻¹y + logdet K̃ + n log 2π)        atol 1e-10
  KAT-3  test/test_acquisition.jl:27-38,81-91,133-144  EI / UCB / PI on the same GP
  KAT-4  test/test_bayesian_opt.jl:516-558 UCB ≡ −μ + β√σ²                                atol 1e-10
  KAT-5  test/test_bayesian_opt.jl:463-486 variance grows away from the data
  KAT-6  test/test_bayesian_opt.jl:759-784 near-duplicate point, σ²_n = 0 → Cholesky must fail

plus seeded random cases for every kernel family / dimension so the fp64 implementations are
compared against values that do not come from any fp64 implementation.  Expected values are the
mpmath results rounded once to fp64.  The formulas are evaluated straight from their definitions
(dense K̃, mp.cholesky_solve) — deliberately not via the L⁻¹ / W route the product takes.

Run:  python tests/golden/make_golden.py      (writes tests/golden/kat.json, random_small.json, grad_small.json)
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 60
HERE = os.path.dirname(os.path.abspath(__file__))

SE, M52, M72, M32 = 0, 1, 2, 3


def kappa(family, d2):
    if family == SE:
        return mp.exp(-d2 / 2)
    d = mp.sqrt(d2)
    if family == M52:
        s = mp.sqrt(5)
        return (1 + s * d + 5 * d2 / 3) * mp.exp(-s * d)
    if family == M72:
        s = mp.sqrt(7)
        return (1 + s * d + mp.mpf(14) / 5 * d2 + 7 * s / 15 * d2 * d) * mp.exp(-s * d)
    s = mp.sqrt(3)
    return (1 + s * d) * mp.exp(-s * d)


def kfun(family, ell, sf2, x, z):
    d2 = sum(((mp.mpf(a) - mp.mpf(b)) / mp.mpf(ell)) ** 2 for a, b in zip(x, z))
    return mp.mpf(sf2) * kappa(family, d2)


def ncdf(z):
    return mp.erfc(-z / mp.sqrt(2)) / 2


def npdf(z):
    return mp.exp(-z * z / 2) / mp.sqrt(2 * mp.pi)


def solve_case(family, ell, sf2, noise, mean_c, X, y, Z, acq=None):
    """Exact posterior at Z given (X, y); X, Z are lists of d-vectors (python floats)."""
    n = len(X)
    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            K[i, j] = kfun(family, ell, sf2, X[i], X[j]) + (mp.mpf(noise) if i == j else 0)
    delta = mp.matrix([mp.mpf(v) - mp.mpf(mean_c) for v in y])
    alpha = mp.lu_solve(K, delta)
    mu, var = [], []
    for z in Z:
        kz = mp.matrix([kfun(family, ell, sf2, x, z) for x in X])
        w = mp.lu_solve(K, kz)
        mu.append(mp.mpf(mean_c) + (kz.T * alpha)[0])
        var.append(mp.mpf(sf2) - (kz.T * w)[0] + mp.mpf("1e-18"))
    logdet = mp.log(mp.det(K))
    nlml = (n * mp.log(2 * mp.pi) + logdet + (delta.T * alpha)[0]) / 2
    out = {
        "family": family, "ell": ell, "sigma_f2": sf2, "noise_var": noise, "mean_c": mean_c,
        "X": X, "y": y, "Z": Z,
        "mu": [float(v) for v in mu], "var": [float(v) for v in var],
        "alpha": [float(v) for v in alpha], "nlml": float(nlml),
    }
    if acq is not None:
        best_y, xi, beta = acq["best_y"], acq["xi"], acq["beta"]
        ei, ucb, pi = [], [], []
        for m, v in zip(mu, var):
            d_ = (mp.mpf(best_y) - mp.mpf(xi)) - m
            if v <= mp.mpf("1e-12"):
                ei.append(max(d_, 0)); pi.append(max(d_, 0))
            else:
                s = mp.sqrt(v); zz = d_ / s
                ei.append(d_ * ncdf(zz) + s * npdf(zz)); pi.append(ncdf(zz))
            ucb.append(-m + mp.mpf(beta) * mp.sqrt(max(v, 0)))
        out.update({"best_y": best_y, "xi": xi, "beta": beta,
                    "ei": [float(v) for v in ei], "ucb": [float(v) for v in ucb],
                    "pi": [float(v) for v in pi]})
    return out


def kats():
    c = {}
    # KAT-1/2: test/test_surrogates.jl:62-104, :147-169
    c["kat1"] = solve_case(SE, 1.0, 1.0, 0.1, 0.0, [[0.0], [0.5], [1.0]], [0.0, 0.25, 1.0], [[0.25]])
    # KAT-3: test/test_acquisition.jl:22-38 (EI ξ=0.01 best=min y), :76-91 (UCB β=2), :128-144 (PI)
    c["kat3"] = solve_case(SE, 1.0, 1.0, 0.1, 0.0, [[0.0], [0.5], [1.0]], [2.0, 1.0, 0.5], [[0.25]],
                           acq={"best_y": 0.5, "xi": 0.01, "beta": 2.0})
    # KAT-4: test/test_bayesian_opt.jl:516-558
    c["kat4"] = solve_case(SE, 1.0, 1.0, 0.01, 0.0, [[-1.0], [0.0], [1.0]], [1.0, 0.25, 1.0], [[0.5]],
                           acq={"best_y": 0.25, "xi": 0.0, "beta": 2.0})
    # KAT-5: test/test_bayesian_opt.jl:463-486
    c["kat5"] = solve_case(SE, 1.0, 1.0, 0.01, 0.0, [[-1.0, -1.0], [0.0, 0.0], [1.0, 1.0]],
                           [1.5, 0.0, 1.5], [[0.1, 0.1], [1.5, 1.5], [3.0, 3.0]])
    # KAT-6: test/test_bayesian_opt.jl:759-779 — inputs only; expectation: factorisation fails
    c["kat6"] = {"family": SE, "ell": 1.0, "sigma_f2": 1.0, "noise_var": 0.0, "mean_c": 0.0,
                 "X": [[-1.0, -1.0], [5.0, -5.0], [-1.0 + 1e-12, -1.0 + 1e-12]],
                 "y": [1.0, 2.0, 1.0], "expect": "not_positive_definite"}
    return c


def random_cases():
    rng = np.random.default_rng(20251205)
    cases = []
    for family in (SE, M52, M72, M32):
        for d, n, m in ((1, 7, 9), (2, 12, 8), (4, 20, 6), (8, 24, 5)):
            X = rng.uniform(0, 1, (n, d)).tolist()
            Z = rng.uniform(-0.1, 1.1, (m, d)).tolist()
            Z[0] = list(X[3])                       # a candidate exactly on a training point
            y = rng.normal(size=n).tolist()
            ell = float(rng.uniform(0.3, 1.5))
            sf2 = float(rng.uniform(0.5, 3.0))
            noise = float(10 ** rng.uniform(-4, -1))
            mean_c = float(rng.normal()) if d % 2 == 0 else 0.0
            acq = {"best_y": float(min(y)), "xi": 0.01, "beta": 2.0}
            cases.append(solve_case(family, ell, sf2, noise, mean_c, X, y, Z, acq=acq))
    return cases


def grad_case(family, ell, sf2, noise, mean_c, X, Ys, Z):
    """Gradient-enhanced GP (GradientGP.jl): the multi-output kernel is differentiated numerically in mpmath
    (mp.diff on the base kernel — independent of the analytic φ', φ'' the implementations use); rows by
    outputs (q·N + i).  Expected: posterior mean and full covariance of all p outputs at every z in Z."""
    n, d = len(X), len(X[0])
    p = d + 1

    def kbase(*args):
        x, z = args[:d], args[d:]
        d2 = sum(((a - b) / mp.mpf(ell)) ** 2 for a, b in zip(x, z))
        return mp.mpf(sf2) * kappa(family, d2)

    def kq(x, q, z, q2):
        orders = [0] * (2 * d)
        if q > 0:
            orders[q - 1] += 1
        if q2 > 0:
            orders[d + q2 - 1] += 1
        pt = [mp.mpf(v) for v in x] + [mp.mpf(v) for v in z]
        if sum(orders) == 0:
            return kbase(*pt)
        return mp.diff(kbase, tuple(pt), tuple(orders))

    R = p * n
    rows = [(X[i], q) for q in range(p) for i in range(n)]
    K = mp.matrix(R, R)
    for a, (xa, qa) in enumerate(rows):
        for b, (xb, qb) in enumerate(rows):
            K[a, b] = kq(xa, qa, xb, qb) + (mp.mpf(noise) if a == b else 0)
    y = mp.matrix([mp.mpf(Ys[i][q]) - mp.mpf(mean_c[q]) for q in range(p) for i in range(n)])
    alpha = mp.lu_solve(K, y)
    mus, covs = [], []
    for z in Z:
        kz = mp.matrix(R, p)
        for a, (xa, qa) in enumerate(rows):
            for q in range(p):
                kz[a, q] = kq(xa, qa, z, q)          # cov(train row, output q at z)
        w = mp.matrix(R, p)
        for q in range(p):                              # mp.lu_solve takes one right-hand side at a time
            col = mp.lu_solve(K, kz[:, q])
            for a in range(R):
                w[a, q] = col[a]
        mu = [mp.mpf(mean_c[q]) + sum(kz[a, q] * alpha[a] for a in range(R)) for q in range(p)]
        C = mp.matrix(p, p)
        for q in range(p):
            for q2 in range(p):
                C[q, q2] = kq(z, q, z, q2) - sum(kz[a, q] * w[a, q2] for a in range(R)) + (mp.mpf("1e-18") if q == q2 else 0)
        mus.append([float(v) for v in mu])
        covs.append([[float(C[a, b]) for b in range(p)] for a in range(p)])
    nl = (R * mp.log(2 * mp.pi) + mp.log(mp.det(K)) + (y.T * alpha)[0]) / 2
    return {"family": family, "ell": ell, "sigma_f2": sf2, "noise_var": noise, "mean_c": mean_c, "X": X, "Ys": Ys, "Z": Z,
            "mu": mus, "cov": covs, "nlml": float(nl)}


def grad_cases():
    mp.mp.dps = 40
    out = []
    # reference KAT: test/test_surrogates.jl:291-348 (SE, p = 3, noise 0.1; grad mean and 3×3 cov at 1e-10)
    out.append(grad_case(SE, 1.0, 1.0, 0.1, [0.0, 0.0, 0.0], [[0.0, 0.0], [0.5, 0.5], [1.0, 1.0]],
                         [[1.0, 0.1, 0.1], [0.5, 0.0, 0.0], [0.0, -0.1, -0.1]], [[0.25, 0.25]]))
    rng = np.random.default_rng(7)
    for family, d, n in ((SE, 1, 4), (M52, 2, 4), (M72, 2, 3), (M52, 3, 3)):
        X = rng.uniform(0, 1, (n, d)).tolist()
        Ys = rng.normal(size=(n, d + 1)).tolist()
        Z = rng.uniform(0, 1, (2, d)).tolist()
        mean_c = [float(rng.normal())] + [0.0] * d
        out.append(grad_case(family, float(rng.uniform(0.4, 1.2)), float(rng.uniform(0.5, 2.0)), 0.05, mean_c, X, Ys, Z))
    mp.mp.dps = 60
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "grad_small.json"), "w") as f:
        json.dump(grad_cases(), f)
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kats(), f, indent=1)
    with open(os.path.join(HERE, "random_small.json"), "w") as f:
        json.dump(random_cases(), f)
    print("wrote kat.json, random_small.json, grad_small.json")
