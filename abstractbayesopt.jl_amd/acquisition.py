"""Acquisition functions and the grid stage of ``optimize_acquisition``, mirroring
src/acquisition_functions/{ExpectedImprovement,UpperConfidenceBound,ProbabilityImprovement,acq_utils}.jl.

Calling ``acq(surrogate, x)`` on a ``HipStandardGP`` takes the fused GPU path (posterior → EI/UCB/PI
epilogue → optional top-k) in one C-ABI call — the Julia analogue is a method specialised on the
surrogate type, which wins dispatch over ExpectedImprovement.jl:40's ``::AbstractSurrogate``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, replace

import numpy as np

from . import _lib
from ._lib import DEVICE, HOST
from .surrogate import AbstractSurrogate, HipStandardGP, _get_minimum, as_points

ACQ_EI, ACQ_UCB, ACQ_PI, ACQ_MEAN, ACQ_GRADNORM_UCB = 0, 1, 2, 3, 4
MAX_TERMS = 8


class AbstractAcquisition:
    """src/abstract.jl:49."""

    kind = None

    def _p0(self):
        raise NotImplementedError

    def _best(self):
        return 0.0

    def __call__(self, surrogate: AbstractSurrogate, x):
        """acq(surrogate, x::AbstractVector) → scores; scalar x is wrapped as [x] (abstract.jl:67-69)."""
        if np.isscalar(x):
            x = [float(x)]
        scores, _, _ = evaluate(self, surrogate, x, k=0, return_scores=True)
        return scores


@dataclass(frozen=True)
class ExpectedImprovement(AbstractAcquisition):
    """ExpectedImprovement(ξ, best_y) (ExpectedImprovement.jl:12-22, :40-66)."""
    xi: float
    best_y: float
    kind = ACQ_EI

    def _p0(self):
        return float(self.xi)

    def _best(self):
        return float(self.best_y)


@dataclass(frozen=True)
class UpperConfidenceBound(AbstractAcquisition):
    """UpperConfidenceBound(β) (UpperConfidenceBound.jl:12-20, :38-45)."""
    beta: float
    kind = ACQ_UCB

    def _p0(self):
        return float(self.beta)


@dataclass(frozen=True)
class ProbabilityImprovement(AbstractAcquisition):
    """ProbabilityImprovement(ξ, best_y) (ProbabilityImprovement.jl:12-22, :38-63)."""
    xi: float
    best_y: float
    kind = ACQ_PI

    def _p0(self):
        return float(self.xi)

    def _best(self):
        return float(self.best_y)


class EnsembleAcquisition(AbstractAcquisition):
    """EnsembleAcquisition(weights, acqs) (EnsembleAcq.jl:12-27): non-negative weights normalised to sum 1;
    value = Σ wᵢ·acqᵢ(surrogate, x) (:53-55).  On a HipStandardGP the posterior is computed ONCE on the
    device and every member's epilogue (abo_score) runs on that same μ, σ²."""

    def __init__(self, weights, acqs):
        weights = np.asarray(weights, dtype=np.float64)
        assert len(weights) == len(acqs), "weights and acquisitions must align"
        assert np.all(weights >= 0), "weights must be non-negative"
        total = float(weights.sum())
        assert total > 0, "sum of weights must be positive"
        self.weights = weights / total
        self.acquisitions = list(acqs)

    def __call__(self, surrogate: AbstractSurrogate, x):
        import torch
        from .surrogate import mean_and_var
        if np.isscalar(x):
            x = [float(x)]
        terms = flatten_terms(self, surrogate)
        if terms is not None and not hasattr(surrogate, "devices"):
            return evaluate_terms(terms, surrogate, x)[0]       # ONE C-ABI call: one posterior pass, every member's epilogue on it
        zp, m, d, zspace, keep = as_points(x)
        dev = torch.device("cuda", surrogate.device)
        zt = keep if zspace == DEVICE else torch.from_numpy(keep).to(dev)
        mu, var = mean_and_var(surrogate, zt)                         # one posterior pass, stays on the device
        total = torch.zeros(m, dtype=torch.float64, device=dev)
        sc = torch.empty(m, dtype=torch.float64, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        for w, a in zip(self.weights, self.acquisitions):
            if isinstance(a, EnsembleAcquisition):
                total += w * torch.as_tensor(a(surrogate, zt), device=dev)
                continue
            _lib.check(_lib.lib().abo_score(surrogate.device, mu.data_ptr(), var.data_ptr(), m, a.kind, a._p0(), a._best(),
                                            sc.data_ptr()))
            total += w * sc
        return total if zspace == DEVICE else total.cpu().numpy()

    def __eq__(self, other):
        return (isinstance(other, EnsembleAcquisition) and np.array_equal(self.weights, other.weights)
                and self.acquisitions == other.acquisitions)


def update(acq: AbstractAcquisition, ys, surrogate: AbstractSurrogate):
    """update(acq, ys, surrogate): EI/PI take best_y = _get_minimum(surrogate, ys)
    (ExpectedImprovement.jl:81-83, ProbabilityImprovement.jl:79-82); UCB is unchanged
    (UpperConfidenceBound.jl:60-62)."""
    if isinstance(acq, EnsembleAcquisition):                            # EnsembleAcq.jl:57-62
        return EnsembleAcquisition(acq.weights, [update(a, ys, surrogate) for a in acq.acquisitions])
    if isinstance(acq, (ExpectedImprovement, ProbabilityImprovement)):
        return replace(acq, best_y=_get_minimum(surrogate, ys))
    return acq


def copy(acq: AbstractAcquisition):
    if isinstance(acq, EnsembleAcquisition):                            # EnsembleAcq.jl:36-38
        return EnsembleAcquisition(acq.weights.copy(), [copy(a) for a in acq.acquisitions])
    return replace(acq)


def evaluate(acq: AbstractAcquisition, surrogate: HipStandardGP, x, k: int = 0, idx_base: int = 0,
             return_scores: bool = True):
    """One fused C-ABI call: scores = acq(surrogate, x) and, if k > 0, the first k entries of
    `sortperm(scores; rev=true)` (acq_utils.jl:50-52) as (values, 0-based global indices).
    Host inputs give NumPy outputs; a CUDA tensor gives CUDA tensors (nothing crosses PCIe)."""
    if not isinstance(surrogate, HipStandardGP):
        raise TypeError("the fused acquisition path needs a HipStandardGP surrogate")
    if isinstance(acq, EnsembleAcquisition) or getattr(acq, "kind", None) == ACQ_GRADNORM_UCB:
        terms = flatten_terms(acq, surrogate)
        if terms is None or hasattr(surrogate, "devices"):
            raise TypeError("this objective is not served by the fused acquisition call on this surrogate")
        return evaluate_terms(terms, surrogate, x, k=k, idx_base=idx_base, return_scores=return_scores)
    L = _lib.lib()
    zp, m, d, zspace, keep = as_points(x)
    if hasattr(surrogate, "devices") and zspace == HOST and idx_base == 0:   # HipShardedGP: sharded inside the library
        from . import multigpu
        return multigpu.evaluate(acq, surrogate, keep, k=k, return_scores=return_scores)
    if zspace == DEVICE:
        import torch
        dev = keep.device
        scores = torch.empty(m, dtype=torch.float64, device=dev) if return_scores else None
        tv = torch.empty(k, dtype=torch.float64, device=dev) if k > 0 else None
        ti = torch.empty(k, dtype=torch.int64, device=dev) if k > 0 else None
        ptr = lambda t: t.data_ptr() if t is not None else None
    else:
        scores = np.empty(m) if return_scores else None
        tv = np.empty(k) if k > 0 else None
        ti = np.empty(k, dtype=np.int64) if k > 0 else None
        ptr = lambda a: a.ctypes.data if a is not None else None
    st = L.abo_acq(surrogate._require(), zp, m, d, zspace, acq.kind, acq._p0(), acq._best(), idx_base,
                   ptr(scores), k, ptr(tv), ptr(ti), zspace)
    _lib.check(st)
    return scores, tv, ti


def update_and_evaluate(acq: AbstractAcquisition, model: HipStandardGP, xs, ys, x, k: int = 0, return_scores: bool = True,
                        idx_base: int = 0, best_y: float | None = None):
    """`update(model, xs, ys)` followed by `evaluate(acq, new_model, x, k)` in ONE C-ABI call with one host synchronisation
    (abo_fit_acq): the acquisition's launches are queued right behind the fit's.  EI / PI take best_y = min(ys), as
    `update(acq, ys, surrogate)` would (ExpectedImprovement.jl:81-83) — pass `best_y` when the caller already has it (for device-resident ys the minimum otherwise
    costs a device reduction and a synchronisation of its own).  Returns (new_model, scores, top_vals, top_idx).  At the
    reference's own sizes (tens of points, 10 000 grid points) the host round trip between the two calls is a third of a step."""
    from .surrogate import _Handle, _is_torch, parse_contraction
    if isinstance(acq, EnsembleAcquisition) or hasattr(model, "devices") or hasattr(model, "p"):
        from . import update as _update                      # not fused for these: the two calls
        new = _update(model, xs, ys)
        return (new,) + tuple(evaluate(update(acq, ys, new), new, x, k=k, return_scores=return_scores, idx_base=idx_base))
    L = _lib.lib()
    xp, n, d, xspace, xkeep = as_points(xs)
    if _is_torch(ys):
        yt = ys.reshape(-1).contiguous()
        yspace, yp, ny, ykeep = (DEVICE if yt.is_cuda else HOST), yt.data_ptr(), yt.shape[0], yt
        needs_min = best_y is None and isinstance(acq, (ExpectedImprovement, ProbabilityImprovement))
        ymin = float(yt.min().item()) if needs_min else best_y
    else:
        ya = np.ascontiguousarray(np.asarray(ys, dtype=np.float64).reshape(-1))
        yspace, yp, ny, ykeep = HOST, ya.ctypes.data, ya.shape[0], ya
        ymin = best_y if best_y is not None else (float(ya.min()) if ya.size else 0.0)
    if ny != n:
        raise _lib.DimensionMismatch(f"xs has {n} points but ys has {ny} values")
    if xspace != yspace:
        raise ValueError("xs and ys must both be host arrays or both be tensors on the model's GPU")
    if isinstance(acq, (ExpectedImprovement, ProbabilityImprovement)):
        acq = replace(acq, best_y=ymin)
    zp, m, dz, zspace, zkeep = as_points(x)
    if zspace == DEVICE:
        import torch
        dev = zkeep.device
        scores = torch.empty(m, dtype=torch.float64, device=dev) if return_scores else None
        tv = torch.empty(k, dtype=torch.float64, device=dev) if k > 0 else None
        ti = torch.empty(k, dtype=torch.int64, device=dev) if k > 0 else None
        ptr = lambda t: t.data_ptr() if t is not None else None
    else:
        scores = np.empty(m) if return_scores else None
        tv = np.empty(k) if k > 0 else None
        ti = np.empty(k, dtype=np.int64) if k > 0 else None
        ptr = lambda a: a.ctypes.data if a is not None else None
    if dz != d:
        raise _lib.DimensionMismatch(f"candidate dimension {dz}, training dimension {d}")
    hp = C.c_void_p()
    prm = model._params()
    _lib.check(L.abo_create(C.byref(prm), C.byref(hp)))
    h = _Handle(hp.value)
    if getattr(model, "contraction", None) is not None:
        _lib.check(L.abo_set_contraction(h.ptr, *parse_contraction(model.contraction)))
    info = C.c_int64(0)
    st = L.abo_fit_acq(h.ptr, xp, n, d, yp, xspace, C.byref(info), zp, m, zspace, acq.kind, acq._p0(), acq._best(), idx_base,
                       ptr(scores), k, ptr(tv), ptr(ti), zspace)
    _lib.check(st, info.value)
    return model._clone(h), scores, tv, ti


def torch_index(idx, like):
    import torch
    return torch.as_tensor(idx, dtype=torch.int64, device=like.device)


def latin_hypercube(n: int, lower, upper, rng) -> np.ndarray:
    """QuasiMonteCarlo.sample(n, lower, upper, LatinHypercubeSample()) (acq_utils.jl:44-46): one
    point per stratum in every coordinate, strata permuted independently per coordinate.
    Returns (n, d) point-major (the reference's d×n column-major Matrix has the same memory)."""
    lower = np.asarray(lower, dtype=np.float64)
    upper = np.asarray(upper, dtype=np.float64)
    d = lower.shape[0]
    u = np.empty((n, d))
    for c in range(d):
        u[:, c] = (rng.permutation(n) + rng.random(n)) / n
    return lower + u * (upper - lower)


def device_latin_hypercube(n: int, lower, upper, seed: int, device: int = 0, first: int = 0, count: int | None = None):
    """Rows first .. first+count−1 of an n-point Latin-hypercube design, generated on the GPU (abo_lhs) as a
    CUDA tensor: the grid of optimize_acquisition without the host LHS and the H2D copy."""
    import torch
    lower = np.ascontiguousarray(np.asarray(lower, dtype=np.float64))
    upper = np.ascontiguousarray(np.asarray(upper, dtype=np.float64))
    count = n - first if count is None else count
    Z = torch.empty((count, lower.shape[0]), dtype=torch.float64, device=torch.device("cuda", device))
    torch.cuda.current_stream(Z.device).synchronize()
    _lib.check(_lib.lib().abo_lhs(device, n, lower.shape[0], lower.ctypes.data, upper.ctypes.data, int(seed) & (2 ** 64 - 1),
                                  first, count, Z.data_ptr()))
    return Z


def _walk_terms(a, w, grad_ok, out) -> bool:
    """(module level on purpose: a nested recursive function is a reference cycle — function ↔ its own closure cell — that would keep
    the surrogate it closes over, and with it a GPU model, alive until the cyclic garbage collector happens to run)"""
    if isinstance(a, EnsembleAcquisition):
        return all(_walk_terms(m, w * float(wi), grad_ok, out) for wi, m in zip(a.weights, a.acquisitions))
    k = getattr(a, "kind", None)
    if k not in (ACQ_EI, ACQ_UCB, ACQ_PI, ACQ_MEAN, ACQ_GRADNORM_UCB):
        return False
    if k == ACQ_GRADNORM_UCB and not grad_ok:
        return False
    out.append((int(k), float(a._p0()), float(a._best()), float(w)))
    return True


def flatten_terms(acqf, surrogate=None):
    """[(kind, p0, best_y, weight), …] of an acquisition function as a weighted-sum objective (include/abo_hip.h: abo_acq_term) —
    a plain function is one term of weight 1, an EnsembleAcquisition (nested ones included) the list of its members with the
    weights multiplied through (EnsembleAcq.jl:53-55).  None when the library cannot take it: more than 8 terms, an unknown
    member, or a GradientNormUCB member on a model without gradient outputs."""
    out = []
    grad_ok = surrogate is None or hasattr(surrogate, "p")
    if not _walk_terms(acqf, 1.0, grad_ok, out) or not 1 <= len(out) <= MAX_TERMS:
        return None
    return out


def _term_array(terms):
    arr = (_lib.AboAcqTerm * len(terms))()
    for i, (k, p0, b, w) in enumerate(terms):
        arr[i].kind, arr[i].reserved, arr[i].p0, arr[i].best_y, arr[i].weight = k, 0, p0, b, w
    return arr


def evaluate_terms(terms, surrogate, x, k: int = 0, idx_base: int = 0, return_scores: bool = True):
    """`evaluate` for a weighted-sum objective (abo_acq_terms): scores and the stable reverse sort's first k in one call"""
    L = _lib.lib()
    zp, m, d, zspace, keep = as_points(x)
    if zspace == DEVICE:
        import torch
        dev = keep.device
        scores = torch.empty(m, dtype=torch.float64, device=dev) if return_scores else None
        tv = torch.empty(k, dtype=torch.float64, device=dev) if k > 0 else None
        ti = torch.empty(k, dtype=torch.int64, device=dev) if k > 0 else None
        ptr = lambda t: t.data_ptr() if t is not None else None
    else:
        scores = np.empty(m) if return_scores else None
        tv = np.empty(k) if k > 0 else None
        ti = np.empty(k, dtype=np.int64) if k > 0 else None
        ptr = lambda a: a.ctypes.data if a is not None else None
    arr = _term_array(terms)
    _lib.check(L.abo_acq_terms(surrogate._require(), zp, m, d, zspace, arr, len(terms), idx_base, ptr(scores), k, ptr(tv), ptr(ti),
                               zspace))
    return scores, tv, ti


def _library_refinable(acqf, surrogate) -> bool:
    """the on-device refinement serves every objective that flattens into ≤ 8 weighted EI / UCB / PI / GradientNormUCB terms, on
    StandardGP and gradient-enhanced handles (single device or sharded group)"""
    return flatten_terms(acqf, surrogate) is not None


def _refine_opts(max_iter, g_tol, f_abstol, x_abstol, history):
    return _lib.AboRefineOpts(max_iter=int(max_iter), linesearch_max=20, history=int(history), reserved=0, g_tol=float(g_tol),
                              f_abstol=float(f_abstol), x_abstol=float(x_abstol))


def refine_starts(acqf: AbstractAcquisition, surrogate: HipStandardGP, starts, lower, upper, max_iter: int = 100,
                  g_tol: float = 1e-5, f_abstol: float = 2.2e-9, x_abstol: float = 1e-4, history: int = 10, return_iters: bool = False):
    """Local refinement stage of optimize_acquisition (acq_utils.jl:55-71): one box-constrained L-BFGS run per start in the
    reference (Fminbox(LBFGS(HagerZhang(linesearchmax=20))), g_tol=1e-5, f_abstol=2.2e-9, x_abstol=1e-4, central finite
    differences of M = 1 posterior calls).  Here a thin caller of `abo_refine` (include/abo_hip.h): ONE launch, one workgroup per
    start running that start's whole projected L-BFGS on the device with the analytic gradient of the acquisition function.
    Returns (points (S, d), values (S,)).  Ensemble acquisitions (weighted sums of ≤ 8 members) and gradient-enhanced models are
    served by the same call (`abo_refine_terms`); only an objective the library cannot express takes the host loop below
    (`_refine_starts_fd`: the same algorithm on batched finite-difference stencils)."""
    terms = flatten_terms(acqf, surrogate)
    if terms is None:                      # more than 8 members / a member the library does not know: the host loop
        return _refine_starts_fd(acqf, surrogate, starts, lower, upper, max_iter, g_tol, f_abstol, x_abstol, history)
    lower = np.ascontiguousarray(np.asarray(lower, dtype=np.float64))
    upper = np.ascontiguousarray(np.asarray(upper, dtype=np.float64))
    st = np.ascontiguousarray(np.asarray(starts, dtype=np.float64))
    S, d = st.shape
    x, f, it = np.empty((S, d)), np.empty(S), np.zeros((S, 2), dtype=np.int32)
    opts = _refine_opts(max_iter, g_tol, f_abstol, x_abstol, history)
    h = surrogate.shard(0) if hasattr(surrogate, "devices") else surrogate._require()
    arr = _term_array(terms)
    _lib.check(_lib.lib().abo_refine_terms(h, arr, len(terms), lower.ctypes.data, upper.ctypes.data, d,
                                           st.ctypes.data, S, C.byref(opts), x.ctypes.data, f.ctypes.data, it.ctypes.data))
    return (x, f, it) if return_iters else (x, f)


def acquisition_value_and_grad(acqf: AbstractAcquisition, surrogate: HipStandardGP, x):
    """(f (M,), ∇f (M, d)) of EI / UCB / PI at the points x: the evaluation the on-device refinement is built on
    (abo_test_acq_grad) — analytic ∇μ, ∇σ² and the closed-form partials of the acquisition function."""
    z = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if z.ndim == 1:
        z = z[:, None]
    f, g = np.empty(z.shape[0]), np.empty(z.shape)
    terms = flatten_terms(acqf, surrogate)
    if terms is None:
        raise ValueError("the objective does not flatten into at most 8 library terms")
    arr = _term_array(terms)
    _lib.check(_lib.lib().abo_test_acq_grad_terms(surrogate._require(), arr, len(terms), z.ctypes.data, z.shape[0], z.shape[1],
                                                  f.ctypes.data, g.ctypes.data))
    return f, g


def optimize_acquisition_device(acqf: AbstractAcquisition, surrogate: HipStandardGP, domain, n_grid: int = 10_000,
                                n_local: int = 100, seed: int = 0, return_all: bool = False, **opts):
    """optimize_acquisition (acq_utils.jl:33-73) in ONE C-ABI call: Latin-hypercube grid generated on the device(s), scored,
    reduced to the n_local best, every start refined on the device, the best point returned.  Serves EI / UCB / PI, GradientNormUCB
    and EnsembleAcquisitions of them (abo_acq_term) on a HipStandardGP, a HipGradientGP (abo_optimize_acquisition_terms) and their
    sharded groups (abo_mgpu_optimize_acquisition_terms: grid and starts sharded over the group)."""
    lower = np.ascontiguousarray(np.asarray(domain.lower, dtype=np.float64))
    upper = np.ascontiguousarray(np.asarray(domain.upper, dtype=np.float64))
    d, k = lower.shape[0], min(int(n_local), int(n_grid))
    best, val = np.empty(d), C.c_double()
    sx, sv, rx, rv = np.empty((k, d)), np.empty(k), np.empty((k, d)), np.empty(k)
    o = _refine_opts(opts.get("max_iter", 0), opts.get("g_tol", 0), opts.get("f_abstol", 0), opts.get("x_abstol", 0), opts.get("history", 0))
    L = _lib.lib()
    terms = flatten_terms(acqf, surrogate)
    if terms is None:
        raise ValueError("the objective does not flatten into at most 8 library terms (use optimize_acquisition)")
    arr = _term_array(terms)
    if hasattr(surrogate, "devices"):
        fn, h = L.abo_mgpu_optimize_acquisition_terms, surrogate._require_group()
    else:
        fn, h = L.abo_optimize_acquisition_terms, surrogate._require()
    _lib.check(fn(h, arr, len(terms), lower.ctypes.data, upper.ctypes.data, d, int(n_grid), int(n_local),
                  int(seed) & (2 ** 64 - 1), C.byref(o), best.ctypes.data, C.byref(val), sx.ctypes.data, sv.ctypes.data,
                  rx.ctypes.data, rv.ctypes.data))
    if return_all:
        return best, val.value, sx, sv, rx, rv
    return best


def _refine_starts_fd(acqf: AbstractAcquisition, surrogate: HipStandardGP, starts, lower, upper, max_iter: int = 100,
                      g_tol: float = 1e-5, f_abstol: float = 2.2e-9, x_abstol: float = 1e-4, history: int = 10):
    """The host-driven variant of the refinement (ensemble acquisitions, gradient-enhanced models): all S starts advance in
    lockstep, one fused acquisition call evaluates the whole central-difference stencil (2d·S points), one more per
    line-search trial (S points) — projected L-BFGS with Armijo backtracking, the same stopping rules."""
    lower = np.asarray(lower, dtype=np.float64)
    upper = np.asarray(upper, dtype=np.float64)
    x = np.clip(np.asarray(starts, dtype=np.float64).copy(), lower, upper)
    S, d = x.shape
    f = acqf(surrogate, x)
    active = np.isfinite(f)
    eps3 = np.finfo(np.float64).eps ** (1.0 / 3.0)
    Sh, Yh = [], []                                   # L-BFGS history (lists of (S, d) arrays)

    def gradient(xa):
        """central differences, one-sided where the stencil would leave the box"""
        n = xa.shape[0]
        h = eps3 * np.maximum(np.abs(xa), 1.0)                      # (n, d)
        xp = np.minimum(xa + h, upper)
        xm = np.maximum(xa - h, lower)
        pts = np.empty((n, 2 * d, d))
        pts[:] = xa[:, None, :]
        for c in range(d):
            pts[:, 2 * c, c] = xp[:, c]
            pts[:, 2 * c + 1, c] = xm[:, c]
        vals = acqf(surrogate, pts.reshape(n * 2 * d, d)).reshape(n, 2 * d)
        span = xp - xm
        g = (vals[:, 0::2] - vals[:, 1::2]) / np.where(span > 0, span, 1.0)
        return np.where(span > 0, g, 0.0)

    def project(xa, ga):
        """projected gradient: drop components that push against an active bound"""
        return np.where(((xa <= lower) & (ga < 0)) | ((xa >= upper) & (ga > 0)), 0.0, ga)

    x_prev = pg_prev = None
    x_run, f_run = x.copy(), f.copy()                  # where the current inner run of every start began
    fresh_start = np.zeros(S, dtype=bool)              # starts whose next pair spans a restart: skipped
    for _ in range(2 * max_iter + 2):
        idx = np.flatnonzero(active)
        if idx.size == 0:
            break
        pg = np.zeros((S, d))
        pg[idx] = project(x[idx], gradient(x[idx]))
        if x_prev is not None:                                 # curvature pair of the last accepted step
            keep = active & ~fresh_start                       # (a fresh inner run starts without one)
            s_k = np.where(keep[:, None], x - x_prev, 0.0)     # (maximisation: y = g_old − g_new)
            y_k = np.where(keep[:, None], pg_prev - pg, 0.0)
            fresh_start[:] = False
            Sh.append(s_k); Yh.append(y_k)
            if len(Sh) > history:
                Sh.pop(0); Yh.pop(0)
        active &= ~(np.max(np.abs(pg), axis=1) <= g_tol)
        if not active.any():
            break
        # two-loop recursion (ascent direction), batched over starts; pairs with s·y ≤ 0 are skipped per start
        q = pg.copy()
        stack = []
        for s_k, y_k in zip(reversed(Sh), reversed(Yh)):
            sy = np.sum(s_k * y_k, axis=1)
            rho = np.where(sy > 1e-300, 1.0 / np.where(sy > 1e-300, sy, 1.0), 0.0)
            a = rho * np.sum(s_k * q, axis=1)
            stack.append((a, rho, s_k, y_k))
            q = q - a[:, None] * y_k
        if Sh:
            sy = np.sum(Sh[-1] * Yh[-1], axis=1)
            yy = np.sum(Yh[-1] * Yh[-1], axis=1)
            q = q * np.where((sy > 1e-300) & (yy > 0), sy / np.where(yy > 0, yy, 1.0), 1.0)[:, None]
        for a, rho, s_k, y_k in reversed(stack):
            b = rho * np.sum(y_k * q, axis=1)
            q = q + (a - b)[:, None] * s_k
        p = q
        bad = ~(np.sum(p * pg, axis=1) > 0)                    # not an ascent direction → steepest ascent
        p[bad] = pg[bad]
        # Armijo backtracking on the projected step, all unfinished starts per trial in one fused call
        t = np.ones(S)
        if not Sh:                                             # first step: a tenth of the box at most
            width = upper - lower                                # (a degenerate side pins its coordinate, it does not limit the others)
            wmin = float(np.min(width[width > 0])) if np.any(width > 0) else 0.0
            t = np.minimum(1.0, 0.1 * wmin / np.maximum(np.max(np.abs(p), axis=1), 1e-300))
        x_new, f_new = x.copy(), f.copy()
        todo = active.copy()
        for _ls in range(20):                                  # HagerZhang(linesearchmax = 20) in the reference
            j = np.flatnonzero(todo)
            if j.size == 0:
                break
            cand = np.clip(x[j] + t[j, None] * p[j], lower, upper)
            fc = acqf(surrogate, cand)
            ok = np.isfinite(fc) & (fc >= f[j] + 1e-4 * np.sum(pg[j] * (cand - x[j]), axis=1))
            acc = j[ok]
            x_new[acc], f_new[acc] = cand[ok], fc[ok]
            todo[acc] = False
            t[j[~ok]] *= 0.5
        # Fminbox's two levels (csrc/refine.hip: outer_converged): an inner run ends when one of its steps moves x by ≤ x_abstol or f
        # by ≤ f_abstol, or when its line search finds no step; the refinement ends when a WHOLE inner run moved no further than
        # that, else a fresh inner run (that start's curvature pairs dropped) begins where the last one ended
        moved = ~todo
        run_ends = todo | (moved & ((np.max(np.abs(x_new - x), axis=1) <= x_abstol) | (np.abs(f_new - f) <= f_abstol)))
        outer = (np.max(np.abs(x_new - x_run), axis=1) <= x_abstol) | (np.abs(f_new - f_run) <= f_abstol)
        done = run_ends & outer
        fresh = run_ends & ~outer & active
        for s_k, y_k in zip(Sh, Yh):
            s_k[fresh] = 0.0; y_k[fresh] = 0.0
        x_run = np.where(run_ends[:, None], x_new, x_run)
        f_run = np.where(run_ends, f_new, f_run)
        x_prev, pg_prev = np.where(fresh[:, None], x_new, x), np.where(fresh[:, None], 0.0, pg)
        fresh_start |= fresh
        x, f = x_new, f_new
        active &= ~done
    return x, f


def optimize_acquisition(acqf: AbstractAcquisition, surrogate: HipStandardGP, domain, n_grid: int = 10_000,
                         n_local: int = 100, rng=None, return_starts: bool = False, refine: bool = True,
                         device_grid: bool = False):
    """optimize_acquisition (acq_utils.jl:33-73): LHS grid → fused scores → top `n_local` starts
    (:44-52) → local refinement of every start, best refined point returned (:55-73).  `refine=False`
    stops after the grid stage and returns the best grid point.  With return_starts=True also returns
    the (n_local, d) start points and their grid scores."""
    rng = np.random.default_rng() if rng is None else rng
    k = min(n_local, n_grid)
    if device_grid and refine and _library_refinable(acqf, surrogate):
        # grid, selection, refinement and arg-max in one C-ABI call: nothing but the result crosses PCIe
        best, _, sx, sv, _, _ = optimize_acquisition_device(acqf, surrogate, domain, n_grid, n_local,
                                                            seed=int(rng.integers(0, 2 ** 63)), return_all=True)
        return (best, sx, sv) if return_starts else best
    if device_grid and flatten_terms(acqf, surrogate) is not None and not hasattr(surrogate, "devices"):
        grid = device_latin_hypercube(n_grid, domain.lower, domain.upper, int(rng.integers(0, 2 ** 63)), surrogate.device)
        _, vals, idx = evaluate_terms(flatten_terms(acqf, surrogate), surrogate, grid, k=k, return_scores=False)
        vals, idx = vals.cpu().numpy(), idx.cpu().numpy()
        starts = grid[torch_index(idx, grid)].cpu().numpy()
    elif device_grid and not isinstance(acqf, EnsembleAcquisition):
        # the grid is generated, scored and reduced on the GPU; only the k starts come back
        grid = device_latin_hypercube(n_grid, domain.lower, domain.upper, int(rng.integers(0, 2 ** 63)), surrogate.device)
        _, vals, idx = evaluate(acqf, surrogate, grid, k=k, return_scores=False)
        vals, idx = vals.cpu().numpy(), idx.cpu().numpy()
        starts = grid[torch_index(idx, grid)].cpu().numpy()
    elif isinstance(acqf, EnsembleAcquisition):
        grid = latin_hypercube(n_grid, domain.lower, domain.upper, rng)
        scores = acqf(surrogate, grid)
        nan_first = np.isnan(scores)
        idx = np.lexsort((np.arange(n_grid), -np.where(nan_first, np.inf, scores), ~nan_first))[:k]
        vals, starts = scores[idx], grid[idx]
    else:
        grid = latin_hypercube(n_grid, domain.lower, domain.upper, rng)
        _, vals, idx = evaluate(acqf, surrogate, grid, k=k, return_scores=False)
        starts = grid[idx]
    best = starts[0].copy()
    if refine:
        xr, fr = refine_starts(acqf, surrogate, starts, domain.lower, domain.upper)
        fr = np.where(np.isfinite(fr), fr, -np.inf)
        j = int(np.argmax(fr))                                 # first maximum, as the reference's strict `>` keeps
        if fr[j] >= vals[0]:
            best = xr[j].copy()
    if return_starts:
        return best, starts, vals
    return best
