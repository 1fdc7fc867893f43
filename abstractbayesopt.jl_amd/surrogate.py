"""HipStandardGP — the MI355X-resident drop-in for the reference's ``StandardGP <: AbstractSurrogate``
(src/surrogates/StandardGP.jl).  Same names, argument meaning and error behaviour as the reference's
methods; every floating-point operation of update / posterior / NLML runs in libabo_hip.so through
the C-ABI of include/abo_hip.h.  There is no CPU fallback.

Value semantics follow the reference: ``update`` returns a *new* model (StandardGP.jl:82), ``copy``
returns a distinct object that shares the immutable device state (StandardGP.jl:26 deep-copies
α, C, x, δ — 512 MiB per BO step at N = 8192; here it is one reference-count increment).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import AboParams, AboTimings, DEVICE, HOST
from .kernels import ConstMean, Kernel, ZeroMean, extract_scale_and_lengthscale, with_lengthscale


class AbstractSurrogate:
    """src/abstract.jl:33."""


def _is_torch(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def as_points(x, d_expected=None):
    """Accept the reference's input containers: a length-M vector of reals (d = 1,
    test/test_surrogates.jl:67), a vector of d-vectors (acq_utils.jl:47), an (M, d) array, or a
    contiguous float64 torch tensor (host or on the model's GPU).
    Returns (pointer, M, d, space, keepalive)."""
    if _is_torch(x):
        import torch
        t = x
        if t.dtype != torch.float64:
            raise TypeError("torch inputs must be float64")
        if t.dim() == 1:
            t = t[:, None]
        if t.dim() != 2:
            raise _lib.DimensionMismatch("inputs must be (M,) or (M, d)")
        t = t.contiguous()
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()   # the library runs on its own stream
        return t.data_ptr(), t.shape[0], t.shape[1], (DEVICE if t.is_cuda else HOST), t
    try:
        a = np.asarray(x, dtype=np.float64)
    except ValueError as e:   # ragged vector-of-vectors
        raise _lib.DimensionMismatch(f"input points differ in length: {e}") from None
    if a.ndim == 1:
        a = a[:, None]
    if a.ndim != 2:
        raise _lib.DimensionMismatch("inputs must be (M,) or (M, d)")
    a = np.ascontiguousarray(a)
    return a.ctypes.data, a.shape[0], a.shape[1], HOST, a


class _Handle:
    """Owns one abo_gp reference."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().abo_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class HipStandardGP(AbstractSurrogate):
    """StandardGP(kernel, noise_var; mean=nothing) (src/surrogates/StandardGP.jl:41-64)."""

    def __init__(self, kernel: Kernel, noise_var: float, mean=None, device: int | None = None, jitter: float = 0.0,
                 chunk: int = 0, n_max: int = 0, contraction: str | None = None):
        if mean is None:
            mean = ZeroMean()
        inner, scale, ell = extract_scale_and_lengthscale(kernel)
        ell = 1.0 if ell is None else ell
        self.kernel = scale * with_lengthscale(inner, ell)     # normal form
        self.noise_var = noise_var
        self.mean = mean
        self.jitter = float(jitter)
        self.chunk = int(chunk)
        self.n_max = int(n_max)          # capacity for incremental appends (0 = size to the fit)
        # engine of the variance contraction: None = the library default (auto), "fp64", "int8" or "int8:<moduli>"
        self.contraction = contraction
        parse_contraction(contraction)
        if device is None:
            device = _current_device()
        self.device = int(device)
        self._h = None            # gpx === nothing

    # -- reference field names -------------------------------------------------------------------
    @property
    def gpx(self):
        return self._h

    @property
    def gp(self):
        return (self.mean, self.kernel)

    def _params(self) -> AboParams:
        return AboParams(family=self.kernel.family, device=self.device, ell=float(self.kernel.lengthscale),
                         sigma_f2=float(self.kernel.scale), noise_var=float(self.noise_var),
                         mean_c=float(getattr(self.mean, "c", 0.0)), jitter=self.jitter, n_max=self.n_max, chunk=self.chunk)

    def _clone(self, handle):
        m = object.__new__(HipStandardGP)
        m.kernel, m.noise_var, m.mean, m.jitter, m.chunk, m.device, m.n_max = (self.kernel, self.noise_var, self.mean,
                                                                              self.jitter, self.chunk, self.device, self.n_max)
        m.contraction = getattr(self, "contraction", None)
        m._h = handle
        return m

    def _require(self):
        if self._h is None:
            raise ValueError("surrogate is not conditioned on data yet (gpx === nothing): call update first")
        return self._h.ptr

    def timings(self) -> dict:
        t = AboTimings()
        _lib.check(_lib.lib().abo_get_timings(self._require(), C.byref(t)))
        return t.as_dict()

    # Checkpoint / resume: the reference has no serialisation code, a BOStruct is rebuilt from (xs, ys, hyper-
    # parameters) (bayesian_opt.jl:81).  A pickled model is exactly that — hyper-parameters plus the training data
    # read back from the device — and unpickling refits on the current device.
    def __getstate__(self):
        st = {k: v for k, v in self.__dict__.items() if k != "_h"}
        st["_data"] = training_data(self) if self._h is not None else None
        return st

    def __setstate__(self, st):
        data = st.pop("_data", None)
        self.__dict__.update(st)
        self._h = None
        self.device = _current_device()
        if data is not None:
            import abstractbayesopt.jl_amd as _pkg
            self._h = _pkg.update(self, data[0], data[1])._h

    def __copy__(self):
        return copy(self)

    def __repr__(self):
        return f"HipStandardGP({self.kernel!r}, noise_var={self.noise_var}, mean={self.mean}, fitted={self._h is not None})"


def parse_contraction(spec):
    """"auto" | "fp64" | "int8" | "int8:<moduli>" → (engine, moduli) of abo_set_contraction (include/abo_hip.h)."""
    if spec is None:
        return _lib.CONTRACT_AUTO, 0
    name, _, n = str(spec).partition(":")
    engines = {"auto": _lib.CONTRACT_AUTO, "fp64": _lib.CONTRACT_FP64, "int8": _lib.CONTRACT_INT8}
    if name not in engines or (n and not n.isdigit()):
        raise ValueError(f"contraction must be auto, fp64, int8 or int8:<moduli>, not {spec!r}")
    return engines[name], int(n) if n else 0


def set_default_contraction(spec) -> None:
    """Process-wide default engine of the variance contraction for handles created from now on."""
    _lib.check(_lib.lib().abo_set_contraction(None, *parse_contraction(spec)))


def _current_device() -> int:
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except Exception:
        pass
    return 0


# ---- Base.copy ---------------------------------------------------------------------------------
def copy(model: HipStandardGP) -> HipStandardGP:
    """Base.copy(s::StandardGP) (StandardGP.jl:26): `copied.gp === orig.gp`, `copied.gpx !== orig.gpx`
    (test/test_surrogates.jl:139-142)."""
    if model._h is None:
        return model._clone(None)
    _lib.check(_lib.lib().abo_retain(model._h.ptr))
    return model._clone(_Handle(model._h.ptr))


# ---- update --------------------------------------------------------------------------------------
def update(model: HipStandardGP, xs, ys) -> HipStandardGP:
    """update(model::StandardGP, xs, ys) (StandardGP.jl:79-83): full refit, returns a new model.
    Raises PosDefException(info) when K + σ²I is not positive definite (no jitter unless the model
    was built with jitter > 0) and DimensionMismatch on ragged / mismatched inputs."""
    L = _lib.lib()
    xp, n, d, xspace, xkeep = as_points(xs)
    if _is_torch(ys):
        import torch
        yt = ys.reshape(-1).contiguous()
        if yt.dtype != torch.float64:
            raise TypeError("torch targets must be float64")
        yspace = DEVICE if yt.is_cuda else HOST
        yp, ny, ykeep = yt.data_ptr(), yt.shape[0], yt
    else:
        ya = np.ascontiguousarray(np.asarray(ys, dtype=np.float64).reshape(-1))
        yspace, yp, ny, ykeep = HOST, ya.ctypes.data, ya.shape[0], ya
    if ny != n:
        raise _lib.DimensionMismatch(f"xs has {n} points but ys has {ny} values")
    if xspace != yspace:
        raise ValueError("xs and ys must both be host arrays or both be tensors on the model's GPU")
    hp = C.c_void_p()
    prm = model._params()
    _lib.check(L.abo_create(C.byref(prm), C.byref(hp)))
    h = _Handle(hp.value)
    if getattr(model, "contraction", None) is not None:
        _lib.check(L.abo_set_contraction(h.ptr, *parse_contraction(model.contraction)))
    info = C.c_int64(0)
    st = L.abo_fit(h.ptr, xp, n, d, yp, xspace, C.byref(info))
    _lib.check(st, info.value)
    del xkeep, ykeep
    return model._clone(h)


# ---- posterior -----------------------------------------------------------------------------------
def _predict(model, x, want_mu, want_var):
    L = _lib.lib()
    if np.isscalar(x):                       # abstract.jl:67-69 scalar wrapper
        x = [float(x)]
    zp, m, d, zspace, keep = as_points(x)
    if hasattr(model, "devices") and zspace == HOST:          # HipShardedGP: shard the batch over the group's devices
        from . import multigpu
        mu, var = multigpu.mean_and_var(model, keep)
        return (mu if want_mu else None), (var if want_var else None)
    if zspace == DEVICE:
        import torch
        mu = torch.empty(m, dtype=torch.float64, device=keep.device) if want_mu else None
        var = torch.empty(m, dtype=torch.float64, device=keep.device) if want_var else None
        st = L.abo_predict(model._require(), zp, m, d, DEVICE, mu.data_ptr() if want_mu else None,
                           var.data_ptr() if want_var else None, DEVICE)
    else:
        mu = np.empty(m) if want_mu else None
        var = np.empty(m) if want_var else None
        st = L.abo_predict(model._require(), zp, m, d, HOST, mu.ctypes.data if want_mu else None,
                           var.ctypes.data if want_var else None, HOST)
    _lib.check(st)
    return mu, var


def posterior_mean(model: HipStandardGP, x):
    """posterior_mean (StandardGP.jl:329-331, :361-363): mean(model.gpx(x))."""
    return _predict(model, x, True, False)[0]


def posterior_var(model: HipStandardGP, x):
    """posterior_var (StandardGP.jl:345-347, :377-379): var(model.gpx(x)) — latent variance + 1e-18."""
    return _predict(model, x, False, True)[1]


def mean_and_var(model: HipStandardGP, x):
    """[upstream AbstractGPs] mean_and_var(model.gpx(x)) — one fused pass (StandardGP.jl:399)."""
    return _predict(model, x, True, True)


def unstandardized_mean_and_var(model: HipStandardGP, xs, params):
    """unstandardized_mean_and_var (StandardGP.jl:395-404)."""
    mu, sigma = params
    m, v = mean_and_var(model, xs)
    return m * sigma + mu, v * sigma ** 2


# ---- NLML ----------------------------------------------------------------------------------------
def nlml_fitted(model: HipStandardGP) -> float:
    """NLML of the model's own fitted state (no refit)."""
    out = C.c_double()
    _lib.check(_lib.lib().abo_nlml(model._require(), C.byref(out)))
    return out.value


def nlml(model: HipStandardGP, params, xs, ys) -> float:
    """nlml(model, params, xs, ys) (StandardGP.jl:99-114): params = [log ℓ, log scale]; rebuilds the
    kernel with exp.(params), keeps noise and mean, returns −logpdf (value only — ForwardDiff duals
    cannot cross a C-ABI; SURVEY.md §8(f) rank 2 tracks the analytic gradient)."""
    log_ell, log_scale = params
    if hasattr(log_ell, "partials") or hasattr(log_scale, "partials"):
        # dual-number parameters (what Optim's autodiff=:forward feeds the objective, bayesian_opt.jl:276-285): value and
        # analytic gradient at the values, partials through the chain rule
        from .hyperparams import nlml_dual
        return nlml_dual(model, (log_ell, log_scale), xs, ys)
    k = math.exp(log_scale) * with_lengthscale(get_kernel_constructor(model), math.exp(log_ell))
    g = HipStandardGP(k, model.noise_var, mean=model.mean, device=model.device, jitter=model.jitter, contraction=model.contraction)
    return nlml_fitted(update(g, xs, ys))


def nlml_ls(model: HipStandardGP, log_ell, log_scale, xs, ys) -> float:
    """nlml_ls (StandardGP.jl:133-149)."""
    return nlml(model, (log_ell, log_scale), xs, ys)


# ---- standardisation helpers (host-side scalars; StandardGP.jl:164-232) -----------------------------
def get_mean_std(model: HipStandardGP, y_train, choice: str):
    y = np.asarray(y_train, dtype=np.float64).reshape(-1)
    y_mean, y_std = float(np.mean(y)), float(np.std(y, ddof=1))
    if choice == "scale_only":
        y_mean = 0.0
    elif choice == "mean_only":
        y_std = 1.0
    return y_mean, y_std


def std_y(model: HipStandardGP, ys, mu, sigma):
    return (np.asarray(ys, dtype=np.float64) - mu) / sigma


def rescale_model(model: HipStandardGP, sigma):
    ell = get_lengthscale(model)[0]
    new_scale = get_scale(model)[0] / sigma ** 2
    new_kernel = new_scale * with_lengthscale(get_kernel_constructor(model), ell)
    mean = model.mean
    if not isinstance(mean, ZeroMean):
        mean = ConstMean(mean.c / sigma)
    return HipStandardGP(new_kernel, model.noise_var / sigma ** 2, mean=mean, device=model.device,
                         jitter=model.jitter, chunk=model.chunk, n_max=model.n_max, contraction=model.contraction)


def _update_model_parameters(model: HipStandardGP, kernel: Kernel):
    return HipStandardGP(kernel, model.noise_var, mean=model.mean, device=model.device, jitter=model.jitter,
                         chunk=model.chunk, n_max=model.n_max, contraction=model.contraction)


def get_lengthscale(model: HipStandardGP):
    """1-element vector, as the reference returns (StandardGP.jl:261; test_surrogates.jl:32-33)."""
    return [model.kernel.lengthscale]


def get_scale(model: HipStandardGP):
    return [model.kernel.scale]


def get_kernel_constructor(model: HipStandardGP) -> Kernel:
    return Kernel(model.kernel.family)


def prep_input(model: HipStandardGP, xs):
    return xs


def prep_output(model: HipStandardGP, ys):
    return ys


def _get_minimum(model: HipStandardGP, ys):
    """_get_minimum (StandardGP.jl:418)."""
    return float(np.min(np.asarray(ys, dtype=np.float64)))


def get_factor(model: HipStandardGP):
    """(L, alpha, Linv) of the fitted state as host arrays — test introspection."""
    n = C.c_int64()
    d = C.c_int32()
    Lb = _lib.lib()
    _lib.check(Lb.abo_get_n(model._require(), C.byref(n), C.byref(d)))
    N = n.value * getattr(model, "p", 1)          # factor rows: p outputs per point for a gradient-enhanced model
    Lm, Li, al = np.empty((N, N)), np.empty((N, N)), np.empty(N)
    _lib.check(Lb.abo_get_factor(model._require(), Lm.ctypes.data, al.ctypes.data, Li.ctypes.data))
    return Lm, al, Li


def training_data(model: HipStandardGP):
    """(X, y) the model is conditioned on, read back from the device: X (N, d) and y (N,) — for a gradient-enhanced
    model y is (N, p), one row [f, ∇f] per point, the shape `update` takes."""
    n = C.c_int64()
    d = C.c_int32()
    Lb = _lib.lib()
    _lib.check(Lb.abo_get_n(model._require(), C.byref(n), C.byref(d)))
    p = getattr(model, "p", 1)
    X, y = np.empty((n.value, d.value)), np.empty(n.value * p)
    _lib.check(Lb.abo_get_data(model._require(), X.ctypes.data, y.ctypes.data))
    return X, (y if p == 1 else np.ascontiguousarray(y.reshape(p, n.value).T))
