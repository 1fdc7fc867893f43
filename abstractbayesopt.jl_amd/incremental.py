"""Incremental path of BASELINE config 5: bordered ("rank-1") Cholesky append, candidate sets resident
in HBM with O(N·M) posterior down-dates, and greedy (Kriging-believer) q-EI on top of them.

The reference has no counterpart — its `update` always refits (src/surrogates/StandardGP.jl:79-83) and
its EI is single-point (src/acquisition_functions/ExpectedImprovement.jl:40-66).  Every sub-step here
has exactly the semantics of those two (condition on one more point; EI over the grid), so parity is
checked against the from-scratch path on the N+j points."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DEVICE, HOST
from .acquisition import AbstractAcquisition, ExpectedImprovement
from .surrogate import HipStandardGP, _Handle, as_points


def append(model: HipStandardGP, x, y) -> HipStandardGP:
    """Condition on one more observation (x, y) in O(N²): returns a new model sharing the factor storage;
    `model` stays valid (free rollback).  Raises PosDefException(N+1) if the bordered pivot is ≤ 0.
    Give the model capacity with HipStandardGP(..., n_max=N_max) to avoid refit fallbacks."""
    L = _lib.lib()
    xa = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    hp = C.c_void_p()
    info = C.c_int64(0)
    if hasattr(model, "p"):
        # gradient-enhanced model: y = [f(x), ∇f(x)…] (one row of the ys `update` takes); p rows join the factor
        ya = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        if ya.shape[0] != model.p:
            raise _lib.DimensionMismatch(f"the observation must hold p = {model.p} values (f and its gradient)")
        st = L.abo_append_grad(model._require(), xa.ctypes.data, xa.shape[0], ya.ctypes.data, C.byref(info), C.byref(hp))
    else:
        st = L.abo_append(model._require(), xa.ctypes.data, xa.shape[0], float(y), C.byref(info), C.byref(hp))
    _lib.check(st, info.value)
    return model._clone(_Handle(hp.value))


class _CandHandle:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().abo_cand_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class ResidentCandidates:
    """M candidates kept on the GPU together with their posterior μ, σ² under `model`."""

    def __init__(self, model: HipStandardGP, Z):
        L = _lib.lib()
        zp, m, d, space, keep = as_points(Z)
        hp = C.c_void_p()
        _lib.check(L.abo_cand_create(model._require(), zp, m, d, space, C.byref(hp)))
        self._h = _CandHandle(hp.value)
        self.M, self.d = m, d
        self.model = model

    def refresh(self, model: HipStandardGP):
        """Full re-evaluation (after a refit or a hyper-parameter change)."""
        _lib.check(_lib.lib().abo_cand_refresh(model._require(), self._h.ptr))
        self.model = model

    def exclude(self, idx: int):
        """Take candidate `idx` (local index) out of the running until the next refresh / restore."""
        _lib.check(_lib.lib().abo_cand_exclude(self.model._require(), self._h.ptr, int(idx)))

    def downdate(self, model: HipStandardGP):
        """`model` must be `append(self.model, x, y)`: O(N·M) update of the stored μ, σ²."""
        _lib.check(_lib.lib().abo_cand_downdate(model._require(), self._h.ptr))
        self.model = model

    def save(self):
        """Snapshot the stored posterior (before exploring fantasy appends)."""
        _lib.check(_lib.lib().abo_cand_save(self.model._require(), self._h.ptr))
        self._saved_model = self.model

    def restore(self):
        """Roll back to the last snapshot."""
        _lib.check(_lib.lib().abo_cand_restore(self._saved_model._require(), self._h.ptr))
        self.model = self._saved_model

    def mean_and_var(self):
        mu, var = np.empty(self.M), np.empty(self.M)
        _lib.check(_lib.lib().abo_cand_get(self.model._require(), self._h.ptr, mu.ctypes.data, var.ctypes.data, HOST))
        return mu, var

    def point(self, idx: int):
        x = np.empty(self.d)
        mu, var = C.c_double(), C.c_double()
        _lib.check(_lib.lib().abo_cand_point(self.model._require(), self._h.ptr, int(idx), x.ctypes.data,
                                             C.byref(mu), C.byref(var)))
        return x, mu.value, var.value

    def qei(self, q: int, xi: float, best_y: float, distinct: bool = False, idx_base: int = 0, block: int = 0):
        """abo_cand_qei: the whole greedy q-EI batch in ONE library call (block form: T = `block` points per block, 0 = the
        library default, < 0 = the plain loop); model and stored posterior are unchanged on return.
        Returns (points (q, d), global indices, EI values, statistics)."""
        X, idx, ei = np.empty((q, self.d)), np.empty(q, dtype=np.int64), np.empty(q)
        st = _lib.AboQeiStats()
        _lib.check(_lib.lib().abo_cand_qei(self.model._require(), self._h.ptr, int(q), float(xi), float(best_y), int(bool(distinct)),
                                           int(idx_base), int(block), X.ctypes.data, idx.ctypes.data, ei.ctypes.data, C.byref(st)))
        return X, idx, ei, st.as_dict()

    def evaluate(self, acq: AbstractAcquisition, k: int = 0, idx_base: int = 0, return_scores: bool = False,
                 device_out: bool = False):
        """Acquisition epilogue + top-k on the stored posterior (no kernel evaluations)."""
        L = _lib.lib()
        if device_out:
            import torch
            dev = torch.device("cuda", self.model.device)
            scores = torch.empty(self.M, dtype=torch.float64, device=dev) if return_scores else None
            tv = torch.empty(k, dtype=torch.float64, device=dev) if k > 0 else None
            ti = torch.empty(k, dtype=torch.int64, device=dev) if k > 0 else None
            ptr = lambda t: t.data_ptr() if t is not None else None
            space = DEVICE
        else:
            scores = np.empty(self.M) if return_scores else None
            tv = np.empty(k) if k > 0 else None
            ti = np.empty(k, dtype=np.int64) if k > 0 else None
            ptr = lambda a: a.ctypes.data if a is not None else None
            space = HOST
        _lib.check(L.abo_cand_acq(self.model._require(), self._h.ptr, acq.kind, acq._p0(), acq._best(), idx_base,
                                  ptr(scores), k, ptr(tv), ptr(ti), space))
        return scores, tv, ti


def _qei_block_batch(model, cands, q, n_cond, xi, best_y, idx_base, group, distinct, block):
    """The picks of a greedy q-EI batch in the BLOCK form (include/abo_hip.h "block form"; csrc/qei.hip): the posterior covariances
    of the T best candidates to every candidate from ONE pass over the resident K_ZX, rank-1 corrections between picks, no fantasy
    appends.  n_cond picks are conditioned on (q − 1 suffice for the picks themselves).  The set's stored posterior is rolled back
    before this returns; the chain of down-date columns stays with the set, so that appending the picks afterwards — as fantasies or
    with their real observations, in order — costs no further pass (abo_cand_downdate).  Sharded (torch.distributed): every rank makes
    the same calls with the same all-gathered records, so the picks equal the single-handle batch bit for bit.
    Returns (points, global indices, EI values, μ at the picks, statistics)."""
    L = _lib.lib()
    d, h, c = cands.d, model._require(), cands._h.ptr
    dist, world = None, 1
    if group is not None or _dist_ready():
        import torch.distributed as dist
        world = dist.get_world_size(group)
    m_tot = cands.M
    if world > 1:
        m_tot = int(_allgather_rows(np.array([[float(cands.M)]]), dist, group).sum())
    if m_tot < 1:
        raise ValueError("q-EI: the candidate set is empty")
    _lib.check(L.abo_cand_qei_begin(h, c, q, int(block or 0)))
    st = _lib.AboQeiStats()
    try:
        _lib.check(L.abo_cand_qei_stats(h, c, C.byref(st)))
        T = min(int(st.block), m_tot)
        picks, idxs, vals, mus = [], [], [], []
        nch, has = C.c_int32(0), C.c_int32(0)
        _lib.check(L.abo_cand_qei_has(h, c, -1, None, C.byref(nch)))
        n_chain = int(nch.value)                       # real appends the set's state carries over from earlier batches
        for j in range(q):
            words = 4 + d + n_chain
            rec = np.empty((1, words))
            _lib.check(L.abo_cand_qei_top(h, c, xi, best_y, idx_base, 1, rec.ctypes.data, rec.size))
            recs = _allgather_rows(rec, dist, group) if world > 1 else rec
            recs = _sorted_valid(recs)
            if len(recs) == 0:
                raise ValueError("q-EI: the candidate set is empty")
            w = recs[0]
            gidx = int(w[1])
            picks.append(w[4:4 + d].copy()); idxs.append(gidx); vals.append(w[0]); mus.append(w[2])
            if j >= n_cond:
                break
            _lib.check(L.abo_cand_qei_has(h, c, gidx, C.byref(has), None))
            if not has.value:
                rt = np.empty((T, words))
                _lib.check(L.abo_cand_qei_top(h, c, xi, best_y, idx_base, T, rt.ctypes.data, rt.size))
                rt = _sorted_valid(_allgather_rows(rt, dist, group) if world > 1 else rt)[:T]
                pts = np.ascontiguousarray(rt[:, 4:4 + d])
                gix = np.ascontiguousarray(rt[:, 1].astype(np.int64))
                if gidx not in set(gix.tolist()):
                    raise _lib.AboError("q-EI: internal error: the pick is not among the block's points")
                _lib.check(L.abo_cand_qei_block(h, c, pts.ctypes.data, gix.ctypes.data, len(gix)))
            cx = np.ascontiguousarray(w[4 + d:4 + d + n_chain]) if n_chain else np.zeros(1)
            excl = gidx - idx_base if (distinct and idx_base <= gidx < idx_base + cands.M) else -1
            info = C.c_int64(0)
            _lib.check(L.abo_cand_qei_pick(h, c, gidx, float(w[3]), cx.ctypes.data, n_chain, excl, C.byref(info)), info.value)
            n_chain += 1
    finally:
        _lib.check(L.abo_cand_qei_end(h, c))
    _lib.check(L.abo_cand_qei_stats(h, c, C.byref(st)))
    return np.array(picks), np.array(idxs, dtype=np.int64), np.array(vals), np.array(mus), st.as_dict()


def _local_block_eligible(model, cands, q, block=0) -> bool:
    """abo_cand_qei_eligible: can THIS rank's shard run the block form (K_ZX resident, model in sync, q ≤ 64, …)?  No side effects."""
    ok = C.c_int32(0)
    _lib.check(_lib.lib().abo_cand_qei_eligible(model._require(), cands._h.ptr, int(q), int(block or 0), C.byref(ok)))
    return bool(ok.value)


def _block_form_agreed(model, cands, q, group, block=0) -> bool:
    """The block form and the plain loop exchange records of different widths in different collectives: every rank must take the
    same one.  Residency of a shard's K_ZX depends on that rank's allocations, so the ranks all-gather their own answer and take the
    block form only if ALL can (the C multi-device path does the same: abo_mgpu_cand_qei runs qei_eligible on every shard first)."""
    mine = _local_block_eligible(model, cands, q, block)
    if not (group is not None or _dist_ready()):
        return mine
    import torch.distributed as dist
    if dist.get_world_size(group) <= 1:
        return mine
    flags = _allgather_rows(np.array([[1.0 if mine else 0.0]]), dist, group)
    return bool(np.all(flags > 0.5))


def _sorted_valid(recs: np.ndarray) -> np.ndarray:
    """records {score, global index, …} in the reference's order (score descending with NaN first, ties → lowest index); records of
    empty shards (index −1) dropped"""
    v = recs[recs[:, 1] >= 0]
    if len(v) == 0:
        return v
    b = v[:, 0].copy().view(np.uint64)
    neg = (b >> np.uint64(63)).astype(bool)
    key = np.where(neg, ~b, b | np.uint64(0x8000000000000000))
    key = np.where(np.isnan(v[:, 0]), np.uint64(0xFFFFFFFFFFFFFFFF), key)
    order = np.lexsort((v[:, 1], np.iinfo(np.uint64).max - key))
    return v[order]


def _allgather_rows(rows: np.ndarray, dist, group) -> np.ndarray:
    """all_gather a (k, words) block of records per rank → (world·k, words)"""
    import torch
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(rows))
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.cat(out).cpu().numpy()


def greedy_qei(model: HipStandardGP, cands: ResidentCandidates, q: int, xi: float, best_y: float, idx_base: int = 0,
               group=None, distinct: bool = False, condition_last: bool = True, block=None, rollback: bool = False,
               stats: dict = None):
    """Greedy q-EI (Kriging believer): for j = 1..q  pick argmax EI over the resident grid, condition on the
    fantasy (z_j, μ(z_j)), repeat.
    With torch.distributed initialised (`group`), every rank holds a shard of the grid: the arg-max is the
    only exchange (one all_gather of a pick record per pick) and every rank applies the same conditioning.
    Returns (batch points (q, d), their global indices, their EI values, the final model).
    `cands` must be in sync with `model`; on return it is in sync with the returned model.
    distinct=True takes every picked candidate out of the running (abo_cand_exclude): with observation noise the
    fantasy does not collapse the variance at a picked point, and the plain rule may return it again.
    condition_last=False leaves the q-th pick unconditioned (the batch does not depend on it): model and `cands` then hold
    q − 1 fantasies on return.
    rollback=True: the batch only — model and `cands` are as before on return (what a BO step wants: the points go to the real
    objective, and the real observations are appended afterwards; in the block form those appends find their down-date columns in
    the chain the batch left with the set).
    block: None = the library's default (block form with T = 16 when the model and the set qualify), 0 / False = the plain loop
    (one bordered append and one O(N·M) pass over the resident K_ZX per pick), T = block form with T points per block.
    stats (a dict, optional) receives the batch's statistics (abo_qei_stats)."""
    use_block = (block is None or bool(block)) and not hasattr(model, "p") and 1 <= q <= 64
    sharded = False
    if group is not None or _dist_ready():
        import torch.distributed as _d
        sharded = _d.get_world_size(group) > 1
    if use_block and sharded:
        # agreed BEFORE the paths diverge: a rank that cannot run the block form would leave the others' all-gathers without a partner
        use_block = _block_form_agreed(model, cands, q, group, 0 if block is None or block is True else int(block))
    if use_block and rollback and not sharded:
        # one shard, the batch only: the library's one-call driver (abo_cand_qei — the same steps, same bits, no interpreter between them)
        pts, idxs, vals, st = cands.qei(q, xi, best_y, distinct=distinct, idx_base=idx_base,
                                        block=0 if block is None or block is True else int(block))
        if stats is not None:
            stats.update(st)
        return pts, idxs, vals, model
    if use_block:
        n_cond = q - 1 if (rollback or not condition_last) else q
        try:
            pts, idxs, vals, mus, st = _qei_block_batch(model, cands, q, n_cond, xi, best_y, idx_base, group, distinct,
                                                        0 if block is None or block is True else int(block))
        except ValueError as e:
            if sharded or ("block form" not in str(e) and "block size 0" not in str(e)):
                raise                          # (sharded: the ranks agreed on the block form — falling back alone would desynchronise them)
            use_block = False                  # the set or the model does not qualify: the plain loop below
    if use_block:
        if stats is not None:
            stats.update(st)
        if rollback:
            return pts, idxs, vals, model
        for j in range(n_cond):                # materialise the fantasies: each down-date finds its column in the chain
            model = append(model, pts[j], float(mus[j]))
            cands.downdate(model)
            if distinct and idx_base <= idxs[j] < idx_base + cands.M:
                cands.exclude(int(idxs[j]) - idx_base)
        return pts, idxs, vals, model
    if stats is not None:
        stats.update({"block": 0, "picks": q})
    if rollback:
        cands.save()
        try:
            pts, idxs, vals, _ = greedy_qei(model, cands, q, xi, best_y, idx_base, group, distinct, False, 0, False)
        finally:
            cands.restore()
        return pts, idxs, vals, model
    acq = ExpectedImprovement(xi, best_y)
    picks, idxs, vals = [], [], []
    dist = None
    if group is not None or _dist_ready():
        import torch.distributed as dist
    for _ in range(q):
        _, tv, ti = cands.evaluate(acq, k=1, idx_base=idx_base)
        if ti[0] >= 0:
            x, mu, _ = cands.point(int(ti[0]) - idx_base)
            rec = np.concatenate([[tv[0], float(ti[0]), mu], x])
        else:                                  # empty shard
            rec = np.concatenate([[np.nan, -1.0, 0.0], np.zeros(cands.d)])
        if dist is not None and dist.get_world_size(group) > 1:
            rec = _allgather_best(rec, dist, group)
        score, gidx, mu, x = rec[0], int(rec[1]), rec[2], rec[3:]
        picks.append(x.copy()); idxs.append(gidx); vals.append(score)
        if not condition_last and len(picks) == q:
            break
        if hasattr(model, "p"):
            # gradient-enhanced model: the fantasy observation is the posterior mean of all p outputs at x (every rank
            # evaluates it on its own identical model; its first entry is the μ of the record)
            from .gradient_gp import posterior_grad_mean
            mu = np.asarray(posterior_grad_mean(model, x[None, :]), dtype=np.float64).reshape(-1)
        model = append(model, x, mu)           # fantasy observation y = μ(x): β = 0, only σ² changes
        cands.downdate(model)
        if distinct and idx_base <= gidx < idx_base + cands.M:
            cands.exclude(gidx - idx_base)     # the rank that owns the candidate masks it
    return np.array(picks), np.array(idxs, dtype=np.int64), np.array(vals), model


def _dist_ready():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:
        return False


def _allgather_best(rec: np.ndarray, dist, group):
    """all_gather one record per rank, keep the winner in the reference's order (score desc, ties → lowest
    global index, NaN scores of empty shards ignored)."""
    import torch
    world = dist.get_world_size(group)
    t = torch.from_numpy(rec.copy())
    use_cuda = dist.get_backend(group) == "nccl"
    if use_cuda:
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    recs = torch.stack(out).cpu().numpy()
    valid = recs[recs[:, 1] >= 0]
    order = np.lexsort((valid[:, 1], -valid[:, 0]))
    return valid[order[0]]
