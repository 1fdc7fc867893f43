"""CPU oracle for the gradient-enhanced GP (GradientGP) — fp64, NumPy/SciPy.  TEST INFRASTRUCTURE ONLY
(same rules as gp_oracle.py: only tests/, smoke() and bench.py's cpu_baseline may import it).

Restates src/surrogates/GradientGP.jl of the reference: the multi-output kernel `gradKernel` (:573-606)
evaluated with *analytic* derivatives instead of nested ForwardDiff (the reference's own tests pin
gradKernel to ForwardDiff derivatives of the base kernel at 1e-10, test/test_surrogates.jl:236-287, and this
oracle is pinned the same way — against finite differences and 60-digit mpmath derivatives in
tests/test_oracle.py), the by-outputs ordering of MOInputIsotopicByOutputs (:659-668, prep_output :893-895),
`update` (:659-668), posterior_grad_mean / _var / _cov (:936-971), posterior_mean / _var (:985-1003).

Row ordering everywhere: r = q·N + i  (output q = 0 is f, q = c+1 is ∂f/∂x_c; all N points per output).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import scipy.linalg as sla

from .gp_oracle import FINITE_GP_JITTER, MATERN52, MATERN72, SE, NotPositiveDefinite, _as_points


def phi_derivs(family: int, u: np.ndarray):
    """φ(u), φ'(u), φ''(u) for k = σ_f²·φ(u), u = ‖x−z‖²/ℓ² (src/surrogates/GradientGP.jl:176-182 for
    Matérn-5/2, :400-407 for Matérn-7/2; SE: φ = e^{−u/2})."""
    u = np.asarray(u, dtype=np.float64)
    if family == SE:
        p = np.exp(-0.5 * u)
        return p, -0.5 * p, 0.25 * p
    r = np.sqrt(u)
    if family == MATERN52:
        a = math.sqrt(5.0)
        e = np.exp(-a * r)
        return (1 + a * r + 5 * u / 3) * e, -(5.0 / 6.0) * (1 + a * r) * e, (25.0 / 12.0) * e
    if family == MATERN72:
        a = math.sqrt(7.0)
        e = np.exp(-a * r)
        return ((1 + a * r + 14.0 / 5.0 * u + 7 * a / 15.0 * u * r) * e,
                -(7.0 / 10.0) * (1 + a * r + 7 * u / 3) * e, (49.0 / 60.0) * (1 + a * r) * e)
    raise ValueError("GradientGP needs a twice-differentiable kernel (SE, Matérn-5/2, Matérn-7/2)")


def grad_kernel_matrix(family, ell, sigma_f2, X, Z=None) -> np.ndarray:
    """Full multi-output kernel matrix, rows (q, i) over X, columns (q', k) over Z, by outputs:
        q=0,q'=0 : k           q=0,q'=c'+1 : ∂k/∂z_c'          q=c+1,q'=0 : ∂k/∂x_c
        q=c+1,q'=c'+1 : ∂²k/∂x_c∂z_c'
    with e = (x − z)/ℓ:  ∂k/∂x_c = σ_f²φ'·2e_c/ℓ,  ∂k/∂z_c' = −σ_f²φ'·2e_c'/ℓ,
    ∂²k/∂x_c∂z_c' = −σ_f²(4φ''e_c e_c' + 2φ'δ_cc')/ℓ²."""
    X = _as_points(X)
    Z = X if Z is None else _as_points(Z)
    n, d = X.shape
    m = Z.shape[0]
    E = (X[:, None, :] * (1.0 / ell) - Z[None, :, :] * (1.0 / ell))          # (n, m, d) scaled differences
    u = np.sum(E * E, axis=2)
    p0, p1, p2 = phi_derivs(family, u)
    K = np.empty(((d + 1) * n, (d + 1) * m))
    K[:n, :m] = sigma_f2 * p0
    for c in range(d):
        g = sigma_f2 * p1 * (2.0 * E[:, :, c] / ell)
        K[(c + 1) * n:(c + 2) * n, :m] = g                    # ∂/∂x_c
        K[:n, (c + 1) * m:(c + 2) * m] = -g                   # ∂/∂z_c
        for c2 in range(d):
            h = -sigma_f2 * (4.0 * p2 * E[:, :, c] * E[:, :, c2] + (2.0 * p1 if c == c2 else 0.0)) / (ell * ell)
            K[(c + 1) * n:(c + 2) * n, (c2 + 1) * m:(c2 + 2) * m] = h
    return K


def prior_var(family, ell, sigma_f2, d) -> np.ndarray:
    """k((z,q),(z,q)): σ_f² for q = 0, −2σ_f²φ'(0)/ℓ² for the gradient outputs."""
    _, p1, _ = phi_derivs(family, np.zeros(1))
    return np.concatenate([[sigma_f2], np.full(d, -2.0 * sigma_f2 * p1[0] / (ell * ell))])


@dataclass
class GradGPState:
    family: int
    ell: float
    sigma_f2: float
    noise_var: float
    mean_c: np.ndarray      # (p,) gradConstMean
    X: np.ndarray           # (N, d)
    L: np.ndarray           # (pN, pN)
    alpha: np.ndarray       # (pN,)
    delta: np.ndarray


def prep_output(ys) -> np.ndarray:
    """vec(permutedims(reduce(hcat, ys))) (GradientGP.jl:893-895): all f values, then all ∂₁, …"""
    return np.asarray(ys, dtype=np.float64).T.reshape(-1)


def fit(family, ell, sigma_f2, noise_var, mean_c, X, ys) -> GradGPState:
    """update(model::GradientGP, xs, ys) (GradientGP.jl:659-668): K̃ = K_grad + σ²I over all (d+1)N rows."""
    X = _as_points(X)
    n, d = X.shape
    y = prep_output(ys)
    mean_c = np.asarray(mean_c, dtype=np.float64)
    K = grad_kernel_matrix(family, ell, sigma_f2, X)
    K[np.diag_indices_from(K)] += noise_var
    L, info = sla.lapack.dpotrf(K, lower=1, clean=1)
    if info > 0:
        raise NotPositiveDefinite(info)
    delta = y - np.repeat(mean_c, n)
    alpha = sla.cho_solve((L, True), delta)
    return GradGPState(family, ell, sigma_f2, noise_var, mean_c, X, L, alpha, delta)


def predict_grad(st: GradGPState, Z, cov: bool = False):
    """posterior_grad_mean / posterior_grad_var / posterior_grad_cov (GradientGP.jl:936-971) at all outputs of
    the points Z, by outputs (length p·M); the +1e-18 FiniteGP jitter is on the diagonal."""
    Z = _as_points(Z)
    m, d = Z.shape
    Kxz = grad_kernel_matrix(st.family, st.ell, st.sigma_f2, st.X, Z)        # (pN, pM)
    mu = np.repeat(st.mean_c, m) + Kxz.T @ st.alpha
    V = sla.solve_triangular(st.L, Kxz, lower=True, check_finite=False)
    if cov:
        Kzz = grad_kernel_matrix(st.family, st.ell, st.sigma_f2, Z, Z)
        C = Kzz - V.T @ V
        C[np.diag_indices_from(C)] += FINITE_GP_JITTER
        return mu, C
    var = np.repeat(prior_var(st.family, st.ell, st.sigma_f2, d), m) - np.einsum("ij,ij->j", V, V) + FINITE_GP_JITTER
    return mu, var


def predict(st: GradGPState, Z):
    """posterior_mean / posterior_var of the function output only (GradientGP.jl:985-1003)."""
    m = _as_points(Z).shape[0]
    mu, var = predict_grad(st, Z)
    return mu[:m], var[:m]


def nlml(st: GradGPState) -> float:
    n = st.L.shape[0]
    return 0.5 * (n * math.log(2 * math.pi) + 2.0 * np.sum(np.log(np.diag(st.L))) + float(st.delta @ st.alpha))


def grad_norm_ucb(st: GradGPState, Z, beta: float) -> np.ndarray:
    """GradientNormUCB (src/acquisition_functions/gradNormUCB.jl:39-51), one point at a time:
    −(mᵀm + trΣ) + β·sqrt(max(4mᵀΣm + 2‖Σ‖_F², 1e-12)) on the gradient block of the posterior."""
    Z = _as_points(Z)
    out = np.empty(Z.shape[0])
    for j in range(Z.shape[0]):
        mu, C = predict_grad(st, Z[j:j + 1], cov=True)
        g, S = mu[1:], C[1:, 1:]
        out[j] = -(g @ g + np.trace(S)) + beta * math.sqrt(max(4 * g @ S @ g + 2 * np.sum(S * S), 1e-12))
    return out
