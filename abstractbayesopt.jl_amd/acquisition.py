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

ACQ_EI, ACQ_UCB, ACQ_PI, ACQ_MEAN = 0, 1, 2, 3


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


def update(acq: AbstractAcquisition, ys, surrogate: AbstractSurrogate):
    """update(acq, ys, surrogate): EI/PI take best_y = _get_minimum(surrogate, ys)
    (ExpectedImprovement.jl:81-83, ProbabilityImprovement.jl:79-82); UCB is unchanged
    (UpperConfidenceBound.jl:60-62)."""
    if isinstance(acq, (ExpectedImprovement, ProbabilityImprovement)):
        return replace(acq, best_y=_get_minimum(surrogate, ys))
    return acq


def copy(acq: AbstractAcquisition):
    return replace(acq)


def evaluate(acq: AbstractAcquisition, surrogate: HipStandardGP, x, k: int = 0, idx_base: int = 0,
             return_scores: bool = True):
    """One fused C-ABI call: scores = acq(surrogate, x) and, if k > 0, the first k entries of
    `sortperm(scores; rev=true)` (acq_utils.jl:50-52) as (values, 0-based global indices).
    Host inputs give NumPy outputs; a CUDA tensor gives CUDA tensors (nothing crosses PCIe)."""
    if not isinstance(surrogate, HipStandardGP):
        raise TypeError("the fused acquisition path needs a HipStandardGP surrogate")
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
    st = L.abo_acq(surrogate._require(), zp, m, d, zspace, acq.kind, acq._p0(), acq._best(), idx_base,
                   ptr(scores), k, ptr(tv), ptr(ti), zspace)
    _lib.check(st)
    return scores, tv, ti


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


def optimize_acquisition(acqf: AbstractAcquisition, surrogate: HipStandardGP, domain, n_grid: int = 10_000,
                         n_local: int = 100, rng=None, return_starts: bool = False):
    """Grid stage of optimize_acquisition (acq_utils.jl:33-52): LHS grid → fused scores → top
    `n_local` starts.  Returns the best grid point (the start the reference's refinement loop would
    visit first); with return_starts=True also the (n_local, d) start points and their scores.
    The per-start box-L-BFGS refinement (acq_utils.jl:55-71) is the next row of the scope table."""
    rng = np.random.default_rng() if rng is None else rng
    grid = latin_hypercube(n_grid, domain.lower, domain.upper, rng)
    k = min(n_local, n_grid)
    _, vals, idx = evaluate(acqf, surrogate, grid, k=k, return_scores=False)
    starts = grid[idx]
    if return_starts:
        return starts[0].copy(), starts, vals
    return starts[0].copy()
