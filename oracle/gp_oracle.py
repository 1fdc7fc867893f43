"""CPU oracle for the StandardGP update/posterior/acquisition hot path (fp64, NumPy/SciPy).

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement of the reference's algorithm and is
used solely as the *checker* by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``abstractbayesopt.jl_amd/`` (the product) imports it; the
product path fails loudly when the HIP library is missing.

Parity status: **pinned** by the closed-form identities the reference's own tests assert
(SURVEY.md §8(c) KAT-1..KAT-6), re-derived independently in 60-digit ``mpmath`` arithmetic by
``tests/golden/make_golden.py`` and committed under ``tests/golden/``.  The reference itself is
Julia and cannot run in the build container (no ``julia`` binary; nothing was denied — the tool
simply does not exist), and its arithmetic lives in un-vendored packages (AbstractGPs 0.5,
KernelFunctions 0.10, Distances 0.10, Distributions 0.25 — Project.toml:23-38) whose published
algorithms are restated below.

Every function cites the reference ``file:line`` (relative to /root/reference) it follows.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import scipy.linalg as sla
from scipy.special import erfc

# kernel family ids — shared with include/abo_hip.h (ABO_KERNEL_*)
SE = 0          # SqExponentialKernel  [upstream KernelFunctions]: kappa(d2) = exp(-d2/2)
MATERN52 = 1    # Matern52Kernel [upstream] / ApproxMatern52Kernel (src/surrogates/GradientGP.jl:94-101)
MATERN72 = 2    # ApproxMatern72Kernel (src/surrogates/GradientGP.jl:320-327)
MATERN32 = 3    # Matern32Kernel [upstream]: (1 + sqrt3 d) exp(-sqrt3 d)

# acquisition ids — shared with include/abo_hip.h (ABO_ACQ_*)
ACQ_EI = 0
ACQ_UCB = 1
ACQ_PI = 2
ACQ_MEAN = 3    # score = -mu (pure exploitation; used by tests of the top-k path)

FINITE_GP_JITTER = 1e-18  # AbstractGPs default `(f::AbstractGP)(x) = FiniteGP(f, x, 1e-18)` [upstream]


class NotPositiveDefinite(Exception):
    """Mirror of LinearAlgebra.PosDefException(info) (caught at src/bayesian_opt.jl:126-141)."""

    def __init__(self, info: int):
        super().__init__(f"matrix is not positive definite; Cholesky failed at leading minor {info}")
        self.info = int(info)


def kappa(family: int, d2: np.ndarray) -> np.ndarray:
    """Scalar kernel profile on the *squared* scaled distance d2 = ||x-z||^2 / ell^2.

    SE: exp(-d2/2) [upstream SqExponentialKernel, SqEuclidean metric].
    Matern-5/2: (1 + sqrt5 d + 5 d2/3) exp(-sqrt5 d)  (src/surrogates/GradientGP.jl:94-101; the
    reference's Taylor branch d2<1e-10 -> 1-(5/6)d2 differs from the closed form by <=1e-16, and
    the upstream Matern52Kernel has no branch at all — test/test_kernels.jl:42-56 pins both to
    1e-12 of each other, so one formula serves both).
    Matern-7/2: (1 + sqrt7 d + 14/5 d2 + 7 sqrt7/15 d^3) exp(-sqrt7 d) (GradientGP.jl:320-327).
    """
    d2 = np.asarray(d2, dtype=np.float64)
    if family == SE:
        return np.exp(-0.5 * d2)
    d = np.sqrt(d2)
    if family == MATERN52:
        s5 = math.sqrt(5.0)
        return (1.0 + s5 * d + 5.0 * d2 / 3.0) * np.exp(-s5 * d)
    if family == MATERN72:
        s7 = math.sqrt(7.0)
        return (1.0 + s7 * d + 14.0 / 5.0 * d2 + 7.0 * s7 / 15.0 * d2 * d) * np.exp(-s7 * d)
    if family == MATERN32:
        s3 = math.sqrt(3.0)
        return (1.0 + s3 * d) * np.exp(-s3 * d)
    raise ValueError(f"unknown kernel family {family}")


def _as_points(X) -> np.ndarray:
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:                      # Vector{Float64}: N scalar inputs (d = 1)
        X = X[:, None]
    return np.ascontiguousarray(X)


def sqdist(X: np.ndarray, Z: np.ndarray, ell: float) -> np.ndarray:
    """Pairwise squared distance of the ScaleTransform(1/ell)-ed inputs.

    with_lengthscale(k, ell) = k ∘ ScaleTransform(1/ell) (src/surrogates/StandardGP.jl:51-59):
    inputs are multiplied by s = 1/ell *first*, then the metric is evaluated, and we sum the
    squared differences directly (no ||x||²+||z||²-2x·z expansion).
    """
    s = 1.0 / ell
    Xs = X * s
    Zs = Z * s
    out = np.zeros((Xs.shape[0], Zs.shape[0]))
    for c in range(Xs.shape[1]):
        diff = Xs[:, c][:, None] - Zs[:, c][None, :]
        out += diff * diff
    return out


def kernel_matrix(family, ell, sigma_f2, X, Z=None) -> np.ndarray:
    """sigma_f2 * kappa(||x-z||/ell): ScaledKernel(inner ∘ ScaleTransform(1/ell), sigma_f2)
    (src/surrogates/StandardGP.jl:41-64, surrogates_utils.jl:28-47)."""
    X = _as_points(X)
    Z = X if Z is None else _as_points(Z)
    return sigma_f2 * kappa(family, sqdist(X, Z, ell))


@dataclass
class GPState:
    family: int
    ell: float
    sigma_f2: float
    noise_var: float
    mean_c: float
    X: np.ndarray          # (N, d)
    L: np.ndarray          # (N, N) lower Cholesky factor of K + noise_var I
    alpha: np.ndarray      # (N,)
    delta: np.ndarray      # (N,)  y - m(X)


def fit(family, ell, sigma_f2, noise_var, mean_c, X, y) -> GPState:
    """update(model::StandardGP, xs, ys) (src/surrogates/StandardGP.jl:79-83) →
    AbstractGPs.posterior(FiniteGP(prior, X, noise_var), y) [upstream]:
    C = cholesky(K + σ²I); δ = y − m(X); α = C \\ δ.  No jitter is ever added
    (test/test_bayesian_opt.jl:759-784 requires failure on a singular K)."""
    X = _as_points(X)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    if y.shape[0] != X.shape[0]:
        raise ValueError("DimensionMismatch: xs and ys differ in length")
    K = kernel_matrix(family, ell, sigma_f2, X)
    K[np.diag_indices_from(K)] += noise_var
    # K is symmetric: its transpose view is the same matrix in Fortran order, which LAPACK factors in place without a copy — and the
    # host OpenBLAS runs its LOWER Fortran-order dpotrf 4 × faster than what a C-ordered argument makes it do (tools/host_potrf_probe.py)
    L, info = sla.lapack.dpotrf(K.T, lower=1, clean=1, overwrite_a=1)
    if info > 0:
        raise NotPositiveDefinite(info)
    if info < 0:
        raise ValueError(f"dpotrf illegal argument {-info}")
    delta = y - mean_c
    alpha = sla.cho_solve((L, True), delta)
    return GPState(family, ell, sigma_f2, noise_var, mean_c, X, L, alpha, delta)


def predict(st: GPState, Z, chunk: int = 8192):
    """posterior_mean / posterior_var (src/surrogates/StandardGP.jl:361-363, :377-379):
    μ = m(Z) + K_ZX α;  σ² = k(z,z) − colsum((L⁻¹K_XZ)²) + 1e-18 (latent variance; the 1e-18 is
    AbstractGPs' default FiniteGP noise).  Chunked over M so that the N×M block stays small."""
    Z = _as_points(Z)
    M = Z.shape[0]
    mu = np.empty(M)
    var = np.empty(M)
    for a in range(0, M, chunk):
        b = min(M, a + chunk)
        Kxz = kernel_matrix(st.family, st.ell, st.sigma_f2, st.X, Z[a:b])     # (N, m)
        mu[a:b] = st.mean_c + Kxz.T @ st.alpha
        V = sla.solve_triangular(st.L, Kxz, lower=True, check_finite=False)
        var[a:b] = st.sigma_f2 - np.einsum("ij,ij->j", V, V) + FINITE_GP_JITTER
    return mu, var


def nlml(st: GPState) -> float:
    """−logpdf(FiniteGP, y) = ½(N log2π + logdet C + δᵀC⁻¹δ) (src/surrogates/StandardGP.jl:99-114;
    closed form asserted at test/test_surrogates.jl:151-169)."""
    n = st.X.shape[0]
    logdet = 2.0 * np.sum(np.log(np.diag(st.L)))
    return 0.5 * (n * math.log(2.0 * math.pi) + logdet + float(st.delta @ st.alpha))


def norm_cdf(z):
    """Distributions.Normal(0,1) cdf = erfc(−z/√2)/2 [upstream StatsFuns.normcdf]."""
    return 0.5 * erfc(-np.asarray(z) / math.sqrt(2.0))


def norm_pdf(z):
    z = np.asarray(z)
    return np.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)


def expected_improvement(mu, var, best_y, xi):
    """(EI)(surrogate, x) + _single_input_ei (src/acquisition_functions/ExpectedImprovement.jl:40-66)."""
    mu = np.asarray(mu, dtype=np.float64)
    var = np.asarray(var, dtype=np.float64)
    delta = (best_y - xi) - mu
    small = var <= 1e-12
    sig = np.sqrt(np.where(small, 1.0, var))
    z = delta / sig
    ei = delta * norm_cdf(z) + sig * norm_pdf(z)
    return np.where(small, np.maximum(delta, 0.0), ei)


def upper_confidence_bound(mu, var, beta):
    """(UCB)(surrogate, x) (src/acquisition_functions/UpperConfidenceBound.jl:38-45)."""
    return -np.asarray(mu) + beta * np.sqrt(np.maximum(np.asarray(var), 0.0))


def probability_improvement(mu, var, best_y, xi):
    """(PI)(surrogate, x) (src/acquisition_functions/ProbabilityImprovement.jl:38-63), including the
    reference's σ²≤1e-12 → max(Δ,0) quirk."""
    mu = np.asarray(mu, dtype=np.float64)
    var = np.asarray(var, dtype=np.float64)
    delta = (best_y - xi) - mu
    small = var <= 1e-12
    sig = np.sqrt(np.where(small, 1.0, var))
    return np.where(small, np.maximum(delta, 0.0), norm_cdf(delta / sig))


def acquisition(kind, mu, var, p0, best_y):
    if kind == ACQ_EI:
        return expected_improvement(mu, var, best_y, p0)
    if kind == ACQ_UCB:
        return upper_confidence_bound(mu, var, p0)
    if kind == ACQ_PI:
        return probability_improvement(mu, var, best_y, p0)
    if kind == ACQ_MEAN:
        return -np.asarray(mu, dtype=np.float64)
    raise ValueError(f"unknown acquisition kind {kind}")


def top_k(scores, k):
    """`sortperm(scores; rev=true)[1:min(k,end)]` (src/acquisition_functions/acq_utils.jl:51-52).
    Julia's sortperm is stable, so equal scores keep the lowest index first; NaN sorts as the
    largest value under isless, i.e. first under rev=true.  Returns 0-based indices."""
    scores = np.asarray(scores, dtype=np.float64)
    key = np.where(np.isnan(scores), np.inf, scores)
    nan_rank = np.isnan(scores).astype(np.int8)           # NaNs ahead of +Inf
    order = np.lexsort((np.arange(scores.shape[0]), -key, -nan_rank))
    idx = order[: min(k, scores.shape[0])]
    return scores[idx], idx.astype(np.int64)


def standardize(y, choice="mean_scale"):
    """get_mean_std + std_y (src/surrogates/StandardGP.jl:164-199): Statistics.std is the
    corrected (n−1) sample standard deviation."""
    y = np.asarray(y, dtype=np.float64)
    mu = float(np.mean(y))
    sd = float(np.std(y, ddof=1))
    if choice == "scale_only":
        mu = 0.0
    elif choice == "mean_only":
        sd = 1.0
    return (y - mu) / sd, mu, sd
