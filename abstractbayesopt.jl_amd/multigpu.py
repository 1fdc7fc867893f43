"""HipShardedGP — one host process driving several MI355X through the library's own multi-device handle
(abo_mgpu_*, include/abo_hip.h): the host-side mirror of `HipStandardGP(kernel, noise_var; devices = [...])`.

This is the path a Julia host takes (it has no process launcher): the model is fitted redundantly on every
listed device, `scores = acqf(surrogate, grid)` (src/acquisition_functions/acq_utils.jl:50) is sharded
contiguously inside the library, and `sortperm(scores; rev=true)[1:n_local]` (:51-52) is reproduced globally by
ONE RCCL all-gather of k × (score, index) per device plus the stable merge.  `distributed.py` keeps the other
shape of the same thing — one process per GPU under torch.distributed — which is what `bench.py --gpus N` runs
when the driver launches it with torchrun.  All inputs and outputs here are host arrays."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .acquisition import AbstractAcquisition, ExpectedImprovement
from .surrogate import AbstractSurrogate, HipStandardGP, _Handle, as_points


class _GroupHandle:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().abo_mgpu_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class _McandHandle:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().abo_mgpu_cand_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


def _host(a, dtype=np.float64):
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


class HipShardedGP(HipStandardGP):
    """StandardGP(kernel, noise_var; mean) replicated on `devices` (a device may be listed twice: two shards on it)."""

    def __init__(self, kernel, noise_var, mean=None, devices=(0,), jitter: float = 0.0, chunk: int = 0, n_max: int = 0):
        super().__init__(kernel, noise_var, mean=mean, device=int(devices[0]), jitter=jitter, chunk=chunk, n_max=n_max)
        self.devices = [int(v) for v in devices]
        self._g = None

    @property
    def gpx(self):
        return self._g

    def _clone_group(self, handle):
        m = object.__new__(HipShardedGP)
        m.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_g", "_h")})
        m._h = None
        m._g = handle
        return m

    def _require_group(self):
        if self._g is None:
            raise ValueError("surrogate is not conditioned on data yet (gpx === nothing): call update first")
        return self._g.ptr

    def _require(self):
        """borrowed handle of shard 0 (timings, NLML, factor introspection)"""
        hp = C.c_void_p()
        _lib.check(_lib.lib().abo_mgpu_get(self._require_group(), 0, C.byref(hp)))
        return hp.value

    def shard(self, i: int) -> int:
        hp = C.c_void_p()
        _lib.check(_lib.lib().abo_mgpu_get(self._require_group(), int(i), C.byref(hp)))
        return hp.value

    def exchange(self) -> str:
        """'rccl' or 'host': how the per-device selections reach the merge"""
        nd, ex = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().abo_mgpu_info(self._require_group(), C.byref(nd), None, C.byref(ex)))
        return "rccl" if ex.value == 1 else "host"

    def exchange_note(self) -> str:
        self.exchange()
        return _lib.last_error()

    def __getstate__(self):
        raise TypeError("HipShardedGP is not picklable: rebuild it from (xs, ys, hyper-parameters)")


class HipShardedGradientGP(HipShardedGP):
    """GradientGP(kernel, p, noise_var; mean = gradConstMean(zeros(p))) (GradientGP.jl:617-639) replicated on `devices`:
    the (d+1)N-row system is fitted on every device, candidates are sharded, acquisitions address the function output;
    `append` takes one observation [f(x), ∇f(x)…] and greedy q-EI conditions on the posterior mean of all p outputs."""

    def __init__(self, kernel, p: int, noise_var, mean=None, devices=(0,), jitter: float = 0.0, chunk: int = 0, n_max: int = 0):
        from .gradient_gp import gradConstMean
        super().__init__(kernel, noise_var, mean=None, devices=devices, jitter=jitter, chunk=chunk, n_max=n_max)
        self.p = int(p)
        self.mean = gradConstMean(np.zeros(self.p)) if mean is None else mean
        if len(self.mean.c) != self.p:
            raise _lib.DimensionMismatch(f"mean has {len(self.mean.c)} entries, the model p = {self.p} outputs")

    def _params(self):
        return _lib.AboParams(family=self.kernel.family, device=self.device, ell=float(self.kernel.lengthscale),
                              sigma_f2=float(self.kernel.scale), noise_var=float(self.noise_var), mean_c=float(self.mean.c[0]),
                              jitter=self.jitter, n_max=self.n_max, chunk=self.chunk)

    def _clone_group(self, handle):
        m = object.__new__(HipShardedGradientGP)
        m.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_g", "_h")})
        m._h = None
        m._g = handle
        return m


def update(model: HipShardedGP, xs, ys) -> HipShardedGP:
    """update(model, xs, ys) (StandardGP.jl:79-83; GradientGP.jl:659-668 for a gradient-enhanced group) on every device of the
    group, concurrently; returns a new model."""
    L = _lib.lib()
    xp, n, d, xspace, xkeep = as_points(xs)
    if xspace != _lib.HOST:
        raise TypeError("the multi-device handle takes host arrays")
    grad = isinstance(model, HipShardedGradientGP)
    if grad:
        from .gradient_gp import prep_output
        ya = prep_output(model, ys)                                   # by outputs: all f values, then all ∂₁f, …
        if ya.shape[0] != n * model.p:
            raise _lib.DimensionMismatch(f"xs has {n} points but ys has {ya.shape[0] // model.p} observations")
    else:
        ya = _host(ys).reshape(-1)
        if ya.shape[0] != n:
            raise _lib.DimensionMismatch(f"xs has {n} points but ys has {ya.shape[0]} values")
    prm = model._params()
    devs = (C.c_int32 * len(model.devices))(*model.devices)
    gp = C.c_void_p()
    if grad:
        mean = np.ascontiguousarray(model.mean.c)
        _lib.check(L.abo_mgpu_create_grad(C.byref(prm), model.p, mean.ctypes.data, len(model.devices), devs, C.byref(gp)))
    else:
        _lib.check(L.abo_mgpu_create(C.byref(prm), len(model.devices), devs, C.byref(gp)))
    h = _GroupHandle(gp.value)
    info = C.c_int64(0)
    st = L.abo_mgpu_fit(h.ptr, xp, n, d, ya.ctypes.data, C.byref(info))
    _lib.check(st, info.value)
    return model._clone_group(h)


def copy(model: HipShardedGP) -> HipShardedGP:
    if model._g is None:
        return model._clone_group(None)
    out = C.c_void_p()
    _lib.check(_lib.lib().abo_mgpu_clone(model._g.ptr, C.byref(out)))
    return model._clone_group(_GroupHandle(out.value))


def mean_and_var(model: HipShardedGP, x):
    if np.isscalar(x):
        x = [float(x)]
    zp, m, d, zspace, keep = as_points(x)
    if zspace != _lib.HOST:
        raise TypeError("the multi-device handle takes host arrays")
    mu, var = np.empty(m), np.empty(m)
    _lib.check(_lib.lib().abo_mgpu_predict(model._require_group(), zp, m, d, mu.ctypes.data, var.ctypes.data))
    return mu, var


def evaluate(acq: AbstractAcquisition, model: HipShardedGP, x, k: int = 0, return_scores: bool = True):
    """scores = acq(model, x) over all devices and the merged global top-k (values, 0-based indices into x)."""
    zp, m, d, zspace, keep = as_points(x)
    if zspace != _lib.HOST:
        raise TypeError("the multi-device handle takes host arrays")
    scores = np.empty(m) if return_scores else None
    tv = np.empty(k) if k > 0 else None
    ti = np.empty(k, dtype=np.int64) if k > 0 else None
    ptr = lambda a: a.ctypes.data if a is not None else None
    _lib.check(_lib.lib().abo_mgpu_acq(model._require_group(), zp, m, d, acq.kind, acq._p0(), acq._best(), ptr(scores), k,
                                       ptr(tv), ptr(ti)))
    return scores, tv, ti


def grid_stage(acq: AbstractAcquisition, model: HipShardedGP, lower, upper, n_grid: int = 10_000, n_local: int = 100, seed: int = 0):
    """The grid stage of optimize_acquisition (acq_utils.jl:44-52) with the Latin-hypercube grid generated shard by
    shard on the devices: returns (scores[k], global indices[k], points[k, d]) of the best min(n_local, n_grid)."""
    lo, up = _host(lower), _host(upper)
    d = lo.shape[0]
    k = int(n_local)
    tv, ti, tx = np.empty(k), np.empty(k, dtype=np.int64), np.empty((k, d))
    _lib.check(_lib.lib().abo_mgpu_acq_lhs(model._require_group(), int(n_grid), d, lo.ctypes.data, up.ctypes.data,
                                           int(seed) & (2 ** 64 - 1), acq.kind, acq._p0(), acq._best(), k, tv.ctypes.data,
                                           ti.ctypes.data, tx.ctypes.data))
    keep = ti >= 0
    return tv[keep], ti[keep], tx[keep]


class ShardedCandidates:
    """A candidate grid sharded over the group's devices, resident with its posterior (BASELINE config 5)."""

    def __init__(self, model: HipShardedGP, Z=None, lhs=None):
        L = _lib.lib()
        out = C.c_void_p()
        if Z is not None:
            zp, m, d, zspace, keep = as_points(Z)
            if zspace != _lib.HOST:
                raise TypeError("the multi-device handle takes host arrays")
            _lib.check(L.abo_mgpu_cand_create(model._require_group(), zp, m, d, C.byref(out)))
            self.M, self.d = m, d
        else:
            n, lower, upper, seed = lhs
            lo, up = _host(lower), _host(upper)
            _lib.check(L.abo_mgpu_cand_create_lhs(model._require_group(), int(n), lo.shape[0], lo.ctypes.data, up.ctypes.data,
                                                  int(seed) & (2 ** 64 - 1), C.byref(out)))
            self.M, self.d = int(n), lo.shape[0]
        self._h = _McandHandle(out.value)

    def refresh(self, model: HipShardedGP):
        _lib.check(_lib.lib().abo_mgpu_cand_refresh(model._require_group(), self._h.ptr))

    def evaluate(self, model: HipShardedGP, acq: AbstractAcquisition, k: int):
        tv, ti = np.empty(k), np.empty(k, dtype=np.int64)
        _lib.check(_lib.lib().abo_mgpu_cand_acq(model._require_group(), self._h.ptr, acq.kind, acq._p0(), acq._best(), k,
                                                tv.ctypes.data, ti.ctypes.data))
        return tv, ti

    def mean_and_var(self, model: HipShardedGP):
        """the stored posterior of all M candidates, in their global order"""
        mu, var = np.empty(self.M), np.empty(self.M)
        _lib.check(_lib.lib().abo_mgpu_cand_get(model._require_group(), self._h.ptr, mu.ctypes.data, var.ctypes.data))
        return mu, var

    def greedy_qei(self, model: HipShardedGP, q: int, xi: float, best_y: float, distinct: bool = False):
        """Greedy (Kriging-believer) q-EI inside the library: (points (q, d), global indices, EI values); the model and
        the stored posterior are unchanged on return."""
        X, idx, ei = np.empty((q, self.d)), np.empty(q, dtype=np.int64), np.empty(q)
        _lib.check(_lib.lib().abo_mgpu_cand_qei(model._require_group(), self._h.ptr, int(q), float(xi), float(best_y),
                                                int(bool(distinct)), X.ctypes.data, idx.ctypes.data, ei.ctypes.data))
        return X, idx, ei

    def qei_stats(self, model: HipShardedGP) -> dict:
        """statistics of the last greedy_qei on this set (abo_qei_stats; block = 0: the plain loop ran)"""
        st = _lib.AboQeiStats()
        _lib.check(_lib.lib().abo_mgpu_cand_qei_stats(model._require_group(), self._h.ptr, C.byref(st)))
        return st.as_dict()


def append(model: HipShardedGP, x, y, cands: ShardedCandidates | None = None) -> HipShardedGP:
    """Bordered append of one observation on every device (returns a new model; `model` stays valid); a sharded
    candidate set is down-dated in the same call.  A gradient-enhanced group takes y = [f(x), ∇f(x)…]."""
    L = _lib.lib()
    new = copy(model)
    xa = _host(x).reshape(-1)
    info = C.c_int64(0)
    cp = cands._h.ptr if cands is not None else None
    if isinstance(model, HipShardedGradientGP):
        ya = _host(y).reshape(-1)
        if ya.shape[0] != model.p:
            raise _lib.DimensionMismatch(f"the observation must hold p = {model.p} values (f and its gradient)")
        st = L.abo_mgpu_append_grad(new._g.ptr, xa.ctypes.data, xa.shape[0], ya.ctypes.data, C.byref(info), cp)
        _lib.check(st, info.value)
        return new
    st = L.abo_mgpu_append(new._g.ptr, xa.ctypes.data, xa.shape[0], float(y), C.byref(info),
                           cands._h.ptr if cands is not None else None)
    _lib.check(st, info.value)
    return new
