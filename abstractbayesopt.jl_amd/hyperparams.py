"""Hyper-parameter MLE on the GPU path: mirror of optimize_hyperparameters (src/bayesian_opt.jl:196-328)
and lengthscale_bounds (src/BO_utils.jl:87-159).

The reference minimises nlml(params) with Fminbox(LBFGS) and `autodiff=:forward`: kernel matrix, Cholesky
and solve all run on ForwardDiff duals in generic Julia.  Duals cannot cross a C-ABI, so the library
returns the value *and the analytic gradient* (abo_nlml_grad: K⁻¹ formed on the fp64 MFMA GEMM, ∂K/∂log ℓ
generated on the fly) and the box-constrained quasi-Newton loop (SciPy L-BFGS-B, the same family of
method) stays on the host — one refit + one gradient call per objective evaluation."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from .kernels import with_lengthscale
from .surrogate import (HipStandardGP, _update_model_parameters, get_kernel_constructor, get_scale, prep_input,
                        prep_output, update)


def nlml_and_grad(model: HipStandardGP, params, xs, ys):
    """(nlml, [∂/∂log ℓ, ∂/∂log scale]) at params = [log ℓ, log scale] (value as StandardGP.jl:99-114)."""
    log_ell, log_scale = params
    k = math.exp(log_scale) * with_lengthscale(get_kernel_constructor(model), math.exp(log_ell))
    g = HipStandardGP(k, model.noise_var, mean=model.mean, device=model.device, jitter=model.jitter, contraction=model.contraction)
    fitted = update(g, xs, ys)
    v, d1, d2 = C.c_double(), C.c_double(), C.c_double()
    _lib.check(_lib.lib().abo_nlml_grad(fitted._require(), C.byref(v), C.byref(d1), C.byref(d2)))
    return v.value, np.array([d1.value, d2.value])


class Dual:
    """value + partials — the shape of a ForwardDiff.Dual, which is what the reference's stock driver evaluates the objective
    on (`autodiff=:forward`, bayesian_opt.jl:276-285).  Only what the chain rule of `nlml_dual` needs: the Julia shim
    (integration/julia/HipStandardGP.jl) does the same with ForwardDiff.value / ForwardDiff.partials."""

    def __init__(self, value, partials):
        self.value = float(value)
        self.partials = np.asarray(partials, dtype=np.float64)

    def __add__(self, o):
        return Dual(self.value + o.value, self.partials + o.partials) if isinstance(o, Dual) else Dual(self.value + o, self.partials)

    __radd__ = __add__

    def __mul__(self, o):
        if isinstance(o, Dual):
            return Dual(self.value * o.value, self.value * o.partials + o.value * self.partials)
        return Dual(self.value * o, self.partials * o)

    __rmul__ = __mul__


def nlml_dual(model, params, xs, ys, value_and_grad=None):
    """nlml on Dual parameters [log ℓ, log scale] (either may be a plain float: nlml_ls keeps the scale fixed): the
    library evaluates value and analytic gradient at the VALUES (abo_nlml_grad, one refit) and the callers' partials go
    through the chain rule — Dual(v, g₁·∂p₁ + g₂·∂p₂) — so no dual number crosses the C-ABI."""
    value_and_grad = value_and_grad or nlml_and_grad
    vals = [p.value if isinstance(p, Dual) else float(p) for p in params]
    v, g = value_and_grad(model, vals, xs, ys)
    part = None
    for gi, p in zip(g, params):
        if isinstance(p, Dual):
            part = gi * p.partials if part is None else part + gi * p.partials
    return Dual(v, part) if part is not None else v


def monte_carlo_fill_distance(X_train, domain, n_samples: int = 10_000, rng=None, device=None) -> float:
    """BO_utils.jl:140-159: sup over random x in the box of the distance to the nearest training point.  The samples are drawn on
    the host (the reference draws them with its own RNG); device = a GPU ordinal: the N × n_samples scan runs there (abo_fill_distance —
    on the host it is a second per call at N = 1024, most of a whole optimize_hyperparameters: profiles/r06_hyperparameter_latency.txt);
    device = None: the NumPy scan below (host logic without a GPU, tests/test_host_logic_cpu.py)."""
    rng = np.random.default_rng() if rng is None else rng
    X = np.ascontiguousarray(np.asarray(X_train, dtype=np.float64))
    xs = np.ascontiguousarray(domain.lower + rng.random((n_samples, X.shape[1])) * (domain.upper - domain.lower))
    if device is not None:
        out = C.c_double()
        _lib.check(_lib.lib().abo_fill_distance(int(device), X.ctypes.data, X.shape[0], X.shape[1], _lib.HOST, xs.ctypes.data,
                                                n_samples, _lib.HOST, C.byref(out)))
        return out.value
    h = 0.0
    for a in range(0, n_samples, 1024):
        blk = xs[a:a + 1024]
        d2 = ((blk[:, None, :] - X[None, :, :]) ** 2).sum(-1)
        h = max(h, float(np.sqrt(d2.min(axis=1)).max()))
    return h


def lengthscale_bounds(X_train, domain, min_frac: float = 0.1, max_frac: float = 1.0, n_samples: int = 10_000, rng=None, device=None):
    """BO_utils.jl:87-125: ℓ_upper = max_frac·box width; ℓ_lower = min_frac·fill distance (≥ 1e-12)."""
    d = domain.lower.shape[0]
    ell_upper = max_frac * (domain.upper - domain.lower)
    X = np.asarray(X_train, dtype=np.float64)
    if d > 1:
        if X.ndim != 2 or X.shape[1] != d:
            raise _lib.DimensionMismatch(f"All points in X_train must have dimension {d}")
        h_fill = monte_carlo_fill_distance(X, domain, n_samples=n_samples, rng=rng, device=device)
    else:
        pts = np.sort(X.reshape(-1))
        h_fill = float(np.max(np.diff(np.concatenate([[domain.lower[0]], pts, [domain.upper[0]]]))))
    return np.full(d, max(min_frac * h_fill, 1e-12)), ell_upper


def optimize_hyperparameters(model: HipStandardGP, x_train, y_train, old_params, scale_std: float = 1.0,
                             length_scale_only: bool = False, num_restarts: int = 1, domain=None, rng=None):
    """optimize_hyperparameters (bayesian_opt.jl:196-328): box bounds (:214-242), starting point clamped
    into them (:247), `num_restarts − 1` extra uniform starts in log space (:263-266), the best converged
    run wins (:275-300), the old model is returned if every restart fails (:302-304), otherwise a NEW
    un-conditioned model with the optimised kernel (:319-327)."""
    from scipy.optimize import minimize
    rng = np.random.default_rng() if rng is None else rng
    ls_lo, ls_hi = 1e-3, 1e3
    if domain is not None:
        lo_v, hi_v = lengthscale_bounds(x_train, domain, rng=rng, device=getattr(model, "device", None))
        ls_lo, ls_hi = max(float(np.min(lo_v)), 1e-6), float(np.max(hi_v))
        assert ls_lo < ls_hi
    sc_lo, sc_hi = 1e-3 / scale_std ** 2, 1e6 / scale_std ** 2
    if length_scale_only:
        lower, upper = np.log([ls_lo]), np.log([ls_hi])
    else:
        lower, upper = np.log([ls_lo, sc_lo]), np.log([ls_hi, sc_hi])
    old = np.asarray(old_params, dtype=np.float64)
    eps2 = 2 * np.finfo(np.float64).eps
    if length_scale_only:
        # `clamp.(old_params, lower_bounds .+ 2eps(), upper_bounds .- 2eps())` (bayesian_opt.jl:247) broadcasts the
        # 1-element bounds over BOTH old parameters: the fixed log-scale is clamped into the length-scale box too
        start_full = np.clip(old, lower[0] + eps2, upper[0] - eps2)
    else:
        start_full = np.clip(old, lower + eps2, upper - eps2)
    from . import gradient_gp as G
    grad_model = isinstance(model, G.HipGradientGP)
    if grad_model:                                   # GradientGP: xs stay points, ys stay (N, p) rows; update() packs them
        xs, ys = x_train, y_train
        value_and_grad, rebuild = G.nlml_and_grad, G._update_model_parameters
    else:
        xs, ys = prep_input(model, x_train), prep_output(model, y_train)
        value_and_grad, rebuild = nlml_and_grad, _update_model_parameters

    def obj(p):
        full = [p[0], start_full[1]] if length_scale_only else [p[0], p[1]]
        v, g = value_and_grad(model, full, xs, ys)
        return v, (g[:1] if length_scale_only else g)

    inits = [start_full[:len(lower)].copy()] + [rng.uniform(lower, upper) for _ in range(num_restarts - 1)]
    best_v, best_p = np.inf, None
    for x0 in inits:
        try:
            res = minimize(obj, x0, jac=True, method="L-BFGS-B", bounds=list(zip(lower, upper)),
                           options={"gtol": 1e-6, "ftol": 2.2e-9, "maxls": 20})
        except (_lib.PosDefException, _lib.AboError, FloatingPointError):
            continue                                  # a failed restart is skipped (bayesian_opt.jl:296-299)
        if res.success and res.fun < best_v:
            best_v, best_p = float(res.fun), res.x.copy()
    if best_p is None:
        return model                                  # all restarts failed (:302-304)
    ell = math.exp(best_p[0])
    scale = get_scale(model)[0] if length_scale_only else math.exp(best_p[1])
    return rebuild(model, scale * with_lengthscale(get_kernel_constructor(model), ell))
