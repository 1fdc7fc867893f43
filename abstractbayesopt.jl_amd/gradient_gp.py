"""HipGradientGP — MI355X-resident mirror of the reference's gradient-enhanced surrogate `GradientGP`
(src/surrogates/GradientGP.jl) and of `GradientNormUCB` (src/acquisition_functions/gradNormUCB.jl).

The (d+1)N × (d+1)N multi-output system (function value + gradient at every training point, rows ordered by
outputs like MOInputIsotopicByOutputs) is assembled on the GPU with analytic kernel derivatives — the reference
evaluates every entry with nested ForwardDiff.derivative calls (GradientGP.jl:573-606) — and then runs through
exactly the same factorisation / contraction kernels as the standard GP."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import DEVICE, HOST
from .acquisition import AbstractAcquisition
from .kernels import Kernel, extract_scale_and_lengthscale, with_lengthscale
from .surrogate import HipStandardGP, _Handle, _is_torch, as_points


class gradConstMean:
    """gradConstMean(c) (GradientGP.jl:505-515): constant prior mean per output, c[0] for f, c[1:] for ∇f."""

    def __init__(self, c):
        self.c = np.asarray(c, dtype=np.float64).reshape(-1)

    def __eq__(self, other):
        return isinstance(other, gradConstMean) and np.array_equal(self.c, other.c)

    def __repr__(self):
        return f"gradConstMean(c={self.c.tolist()})"


class HipGradientGP(HipStandardGP):
    """GradientGP(kernel, p, noise_var; mean=gradConstMean(zeros(p))) (GradientGP.jl:617-639)."""

    def __init__(self, kernel: Kernel, p: int, noise_var: float, mean=None, device: int | None = None, jitter: float = 0.0,
                 chunk: int = 0, n_max: int = 0, contraction: str | None = None):
        super().__init__(kernel, noise_var, mean=None, device=device, jitter=jitter, chunk=chunk, n_max=n_max,
                         contraction=contraction)
        self.p = int(p)
        self.mean = gradConstMean(np.zeros(self.p)) if mean is None else mean
        if len(self.mean.c) != self.p:
            raise _lib.DimensionMismatch(f"mean has {len(self.mean.c)} entries, the model p = {self.p} outputs")

    def _params(self):
        from ._lib import AboParams
        return AboParams(family=self.kernel.family, device=self.device, ell=float(self.kernel.lengthscale),
                         sigma_f2=float(self.kernel.scale), noise_var=float(self.noise_var), mean_c=float(self.mean.c[0]),
                         jitter=self.jitter, n_max=self.n_max, chunk=self.chunk)

    def _clone(self, handle):
        m = object.__new__(HipGradientGP)
        m.__dict__.update(self.__dict__)
        m._h = handle
        return m

    def __repr__(self):
        return f"HipGradientGP({self.kernel!r}, p={self.p}, noise_var={self.noise_var}, mean={self.mean}, fitted={self._h is not None})"


def prep_output(model: HipGradientGP, ys) -> np.ndarray:
    """prep_output (GradientGP.jl:893-895): vec(permutedims(reduce(hcat, ys))) — all f values, then all ∂₁f, …"""
    Y = np.asarray(ys, dtype=np.float64)
    if Y.ndim != 2 or Y.shape[1] != model.p:
        raise _lib.DimensionMismatch(f"each observation must hold p = {model.p} values (f and its gradient)")
    return np.ascontiguousarray(Y.T).reshape(-1)


def update(model: HipGradientGP, xs, ys) -> HipGradientGP:
    """update(model::GradientGP, xs, ys) (GradientGP.jl:659-668): ys[i] = [f(x_i), ∇f(x_i)…]."""
    L = _lib.lib()
    xp, n, d, xspace, xkeep = as_points(xs)
    if xspace != HOST:
        xkeep = xkeep.cpu().numpy(); xp = xkeep.ctypes.data
    y = prep_output(model, ys.cpu().numpy() if _is_torch(ys) else ys)
    if y.shape[0] != n * model.p:
        raise _lib.DimensionMismatch(f"xs has {n} points but ys has {y.shape[0] // model.p} observations")
    hp = C.c_void_p()
    prm = model._params()
    prm.mean_c = float(model.mean.c[0])
    mean = np.ascontiguousarray(model.mean.c)
    _lib.check(L.abo_create_grad(C.byref(prm), model.p, mean.ctypes.data, C.byref(hp)))
    h = _Handle(hp.value)
    if getattr(model, "contraction", None) is not None:
        from .surrogate import parse_contraction
        _lib.check(L.abo_set_contraction(h.ptr, *parse_contraction(model.contraction)))
    info = C.c_int64(0)
    _lib.check(L.abo_fit(h.ptr, xp, n, d, y.ctypes.data, HOST, C.byref(info)), info.value)
    return model._clone(h)


def _grad_predict(model: HipGradientGP, x, want_mu=True, want_var=True):
    L = _lib.lib()
    if np.isscalar(x):
        x = [float(x)]
    zp, m, d, zspace, keep = as_points(x)
    n = m * model.p
    if zspace == DEVICE:
        import torch
        mu = torch.empty(n, dtype=torch.float64, device=keep.device) if want_mu else None
        var = torch.empty(n, dtype=torch.float64, device=keep.device) if want_var else None
        ptr = lambda t: t.data_ptr() if t is not None else None
    else:
        mu = np.empty(n) if want_mu else None
        var = np.empty(n) if want_var else None
        ptr = lambda a: a.ctypes.data if a is not None else None
    _lib.check(L.abo_predict_grad(model._require(), zp, m, d, zspace, ptr(mu), ptr(var), zspace))
    return mu, var


def posterior_grad_mean(model: HipGradientGP, x):
    """posterior_grad_mean (GradientGP.jl:936-938): all p outputs, ordered by outputs (length p·M)."""
    return _grad_predict(model, x, True, False)[0]


def posterior_grad_var(model: HipGradientGP, x):
    """posterior_grad_var (GradientGP.jl:951-953)."""
    return _grad_predict(model, x, False, True)[1]


def posterior_grad_cov(model: HipGradientGP, x, beta: float = 0.0, return_all: bool = False):
    """posterior_grad_cov (GradientGP.jl:966-971) for the outputs of ONE point → (p, p); for M points the
    per-point blocks (M, p, p) (cross-point covariances are not formed).  return_all=True also gives the
    point-major means (M, p) and the GradientNormUCB(β) scores (M,) computed in the same pass."""
    L = _lib.lib()
    zp, m, d, zspace, keep = as_points(x)
    if zspace != HOST:
        keep = keep.cpu().numpy(); zp = keep.ctypes.data
    p = model.p
    mu, cov, sc = np.empty((m, p)), np.empty((m, p, p)), np.empty(m)
    _lib.check(L.abo_predict_grad_cov(model._require(), zp, m, d, HOST, float(beta), mu.ctypes.data, cov.ctypes.data,
                                      sc.ctypes.data, HOST))
    if return_all:
        return mu, cov, sc
    return cov[0] if m == 1 else cov


def unstandardized_mean_and_var(model: HipGradientGP, X, params):
    """unstandardized_mean_and_var(gp::GradientGP, X, params) (GradientGP.jl:1014-1030): (M, p) arrays."""
    mu_s, sigma = np.asarray(params[0], dtype=np.float64), float(np.asarray(params[1]).reshape(-1)[0])
    m, v = _grad_predict(model, X)
    m = np.asarray(m).reshape(model.p, -1).T
    v = np.asarray(v).reshape(model.p, -1).T
    return m * sigma + mu_s[None, :], v * sigma ** 2


def get_mean_std(model: HipGradientGP, y_train, choice: str):
    """get_mean_std(::GradientGP) (GradientGP.jl:734-746): only the function values are centred; the gradients
    share the function's scale."""
    Y = np.asarray(y_train, dtype=np.float64)
    mu = Y.mean(axis=0)
    mu[1:] = 0.0
    sd = Y.std(axis=0, ddof=1)
    sd[1:] = sd[0]
    if choice == "scale_only":
        mu[:] = 0.0
    elif choice == "mean_only":
        sd[:] = 1.0
    return mu, sd


def std_y(model: HipGradientGP, ys, mu, sigma):
    """std_y(::GradientGP) (GradientGP.jl:761-764)."""
    return (np.asarray(ys, dtype=np.float64) - np.asarray(mu)[None, :]) / float(np.asarray(sigma).reshape(-1)[0])


def rescale_model(model: HipGradientGP, sigma):
    """rescale_model(::GradientGP) (GradientGP.jl:779-794)."""
    s1 = float(np.asarray(sigma).reshape(-1)[0])
    inner, scale, ell = extract_scale_and_lengthscale(model.kernel)
    k = (scale / s1 ** 2) * with_lengthscale(inner, ell)
    if hasattr(model, "devices"):                        # a sharded group keeps its device list
        from .multigpu import HipShardedGradientGP
        return HipShardedGradientGP(k, model.p, model.noise_var / s1 ** 2, mean=gradConstMean(model.mean.c / s1),
                                    devices=model.devices, jitter=model.jitter, chunk=model.chunk, n_max=model.n_max)
    return HipGradientGP(k, model.p, model.noise_var / s1 ** 2, mean=gradConstMean(model.mean.c / s1), device=model.device,
                         jitter=model.jitter, chunk=model.chunk, n_max=model.n_max, contraction=model.contraction)


def _update_model_parameters(model: HipGradientGP, kernel: Kernel):
    if hasattr(model, "devices"):
        from .multigpu import HipShardedGradientGP
        return HipShardedGradientGP(kernel, model.p, model.noise_var, mean=model.mean, devices=model.devices, jitter=model.jitter,
                                    chunk=model.chunk, n_max=model.n_max)
    return HipGradientGP(kernel, model.p, model.noise_var, mean=model.mean, device=model.device, jitter=model.jitter,
                         chunk=model.chunk, n_max=model.n_max, contraction=model.contraction)


def nlml(model: HipGradientGP, params, xs, ys) -> float:
    """nlml(model::GradientGP, params, xs, ys) (GradientGP.jl:684-698): params = [log ℓ, log scale]; the kernel is
    rebuilt with exp.(params), noise and prior mean are kept; −logpdf of the (d+1)N-row system."""
    from .surrogate import nlml_fitted
    log_ell, log_scale = params
    if hasattr(log_ell, "partials") or hasattr(log_scale, "partials"):     # dual-number parameters: see surrogate.nlml
        from .hyperparams import nlml_dual
        return nlml_dual(model, (log_ell, log_scale), xs, ys, value_and_grad=nlml_and_grad)
    inner = extract_scale_and_lengthscale(model.kernel)[0]
    k = math.exp(log_scale) * with_lengthscale(inner, math.exp(log_ell))
    return nlml_fitted(update(_update_model_parameters(model, k), xs, ys))


def nlml_ls(model: HipGradientGP, log_ell, log_scale, xs, ys) -> float:
    """nlml_ls(model::GradientGP, …) (GradientGP.jl:719-739)."""
    return nlml(model, (log_ell, log_scale), xs, ys)


def nlml_and_grad(model: HipGradientGP, params, xs, ys):
    """(nlml, [∂/∂log ℓ, ∂/∂log scale]) for the gradient-enhanced model from ONE refit: the library forms K⁻¹ on the
    MFMA GEMM, generates ∂K/∂log ℓ of the (d+1)N-row system with the analytic derivative blocks and reduces
    ½ tr((K⁻¹ − ααᵀ) ∂K/∂θ) (abo_nlml_grad).  The reference differentiates this objective with ForwardDiff
    (bayesian_opt.jl:284)."""
    log_ell, log_scale = params
    inner = extract_scale_and_lengthscale(model.kernel)[0]
    k = math.exp(log_scale) * with_lengthscale(inner, math.exp(log_ell))
    fitted = update(_update_model_parameters(model, k), xs, ys)
    v, d1, d2 = C.c_double(), C.c_double(), C.c_double()
    _lib.check(_lib.lib().abo_nlml_grad(fitted._require(), C.byref(v), C.byref(d1), C.byref(d2)))
    return v.value, np.array([d1.value, d2.value])


def nlml_and_grad_fd(model: HipGradientGP, params, xs, ys, h: float = 1e-4):
    """The same pair with fourth-order central differences of the device NLML (8 more refits) — cross-check of the
    analytic gradient."""
    p = np.asarray(params, dtype=np.float64)
    v = nlml(model, p, xs, ys)
    g = np.zeros(2)
    for c in range(2):
        e = np.zeros(2); e[c] = h
        g[c] = (8.0 * (nlml(model, p + e, xs, ys) - nlml(model, p - e, xs, ys))
                - (nlml(model, p + 2 * e, xs, ys) - nlml(model, p - 2 * e, xs, ys))) / (12.0 * h)
    return v, g


def _get_minimum(model: HipGradientGP, ys):
    """_get_minimum(::GradientGP) (GradientGP.jl:1043-1044): minimum over the function values."""
    return float(np.min(np.asarray(ys, dtype=np.float64)[:, 0]))


class GradientNormUCB(AbstractAcquisition):
    """GradientNormUCB(β) (gradNormUCB.jl:12-51): UCB on the squared gradient norm, one point at a time in the
    reference; here all points in one call (per-point p×p covariance blocks on the device)."""

    kind = 4                                  # ABO_ACQ_GRADNORM_UCB: a term of the *_terms entry points (refinement, ensembles)

    def __init__(self, beta: float):
        self.beta = float(beta)

    def _p0(self):
        return self.beta

    def __call__(self, surrogate: HipGradientGP, x):
        if not isinstance(surrogate, HipGradientGP):
            raise TypeError("GradientNormUCB needs a gradient-enhanced surrogate")
        return posterior_grad_cov(surrogate, x, beta=self.beta, return_all=True)[2]

    def __eq__(self, other):
        return isinstance(other, GradientNormUCB) and other.beta == self.beta
