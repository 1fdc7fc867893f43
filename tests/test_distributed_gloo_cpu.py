"""world_size-2 gloo test of the only exchange step of the multi-GPU path: every rank contributes
its shard's local top-k, all_gather, identical merge — must equal the single-process selection over
the whole candidate batch (stable reverse sortperm, acq_utils.jl:51-52).  The local scores here are
synthetic (seeded) numbers standing in for what the HIP path returns per shard."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from abstractbayesopt.jl_amd import distributed as D
from oracle import gp_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(11)
    scores = np.round(rng.normal(size=M), 2)        # ties across shards
    scores[M // 2] = np.nan
    lo, hi = D.shard_range(M, rank, world)
    v, i = O.top_k(scores[lo:hi], k)
    v = np.concatenate([v, np.full(k - len(v), np.nan)])
    i = np.concatenate([i + lo, np.full(k - len(i), -1, dtype=np.int64)])
    mv, mi = D.all_gather_topk(torch.from_numpy(v), torch.from_numpy(i), k)
    q.put((rank, mv.numpy().copy(), mi.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, M, k):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_rank_topk_merge_matches_global():
    M, k = 10001, 100
    res = _run(2, M, k)
    rng = np.random.default_rng(11)
    scores = np.round(rng.normal(size=M), 2)
    scores[M // 2] = np.nan
    ov, oi = O.top_k(scores, k)
    for rank, mv, mi in res:
        np.testing.assert_array_equal(mi, oi)
        np.testing.assert_array_equal(mv, ov)


def test_two_rank_short_shards():
    # fewer candidates than k on each rank: padding entries (NaN, −1) must vanish in the merge
    res = _run(2, 7, 16)
    for rank, mv, mi in res:
        assert len(mi) == 7 and sorted(mi.tolist()) == list(range(7))


def _worker_best(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from abstractbayesopt.jl_amd.incremental import _allgather_best
    # per-rank best record of a greedy q-EI pick: (score, global index, mu, x[3]); rank 1 ties rank 0's score
    # with a higher index, rank 2 (if any) has an empty shard
    recs = {0: [0.7, 41.0, -0.3, 0.1, 0.2, 0.3], 1: [0.7, 99.0, 0.5, 0.4, 0.5, 0.6], 2: [float("nan"), -1.0, 0.0, 0, 0, 0]}
    out = _allgather_best(np.array(recs[rank]), dist, None)
    q.put((rank, out.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_qei_pick_exchange_prefers_lowest_index_on_ties():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_worker_best, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rec in res:
        assert rec[1] == 41.0 and rec[0] == 0.7 and rec[2] == -0.3 and rec[3:].tolist() == [0.1, 0.2, 0.3]


def test_bounded_gather_protocol_with_a_stubbed_transport(tmp_path):
    """The part of the in-library exchange that must never hang — every shard enqueues its all-gather, polls its stream, an abort
    word and a deadline, and takes its OWN communicator down when any of them says so (csrc/abo_exchange.h, used by mgpu.hip) —
    driven on the CPU with a stubbed transport (tests/exchange_stub.cpp): n = 2, 3, 8 shards on one thread each; a peer that never
    enqueues (the others are released by their own aborts, at once), a collective that never completes (released at the deadline),
    a stream error.  The one-GPU box can only ever run this at one rank (tests/test_gpu_multigpu.py: fault injection)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "exchange_stub")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-I", os.path.join(root, "abstractbayesopt.jl_amd", "csrc"),
                           os.path.join(root, "tests", "exchange_stub.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout


def _worker_agree(rank, world, port, flags, q):
    """greedy_qei under torch.distributed with a per-rank eligibility flag (stubbed: no GPU here) and both batch forms stubbed to
    record which one this rank takes — and to run the collective that form runs, so a disagreement would hang (the parent's timeout)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from abstractbayesopt.jl_amd import incremental as I

    class _Cands:
        M, d = 10, 3

        def save(self): pass

        def restore(self): pass

        def evaluate(self, acq, k=0, idx_base=0, **kw):
            return None, np.array([0.5 + rank]), np.array([idx_base + 1], dtype=np.int64)

        def point(self, idx):
            return np.full(3, float(rank)), 0.25, 0.1

    class _Model:
        pass

    taken = []
    I._local_block_eligible = lambda model, cands, q_, block=0: bool(flags[rank])

    def block_batch(model, cands, q_, n_cond, xi, best_y, idx_base, group, distinct, block):
        taken.append("block")
        recs = I._allgather_rows(np.full((1, 7), float(rank)), dist, group)          # the block form's record width (4 + d)
        assert recs.shape == (world, 7)
        return np.zeros((q_, 3)), np.zeros(q_, dtype=np.int64), np.zeros(q_), np.zeros(q_), {"block": 16}

    I._qei_block_batch = block_batch
    real_best = I._allgather_best

    def best(rec, d_, group):
        taken.append("plain")
        return real_best(rec, d_, group)

    I._allgather_best = best
    I.append = lambda model, x, y: model
    _Cands.downdate = lambda self, model: None
    pts, idxs, vals, _ = I.greedy_qei(_Model(), _Cands(), 2, 0.01, 0.0, idx_base=10 * rank, rollback=True)
    q.put((rank, sorted(set(taken)), idxs.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _run_agree(flags):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = len(flags)
    procs = [ctx.Process(target=_worker_agree, args=(r, world, port, flags, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_ranks_agree_on_the_qei_form_before_their_collectives_diverge():
    """a rank whose K_ZX is not resident cannot run the block form; before the fix it fell back to the plain loop ALONE
    (3 + d-word records through _allgather_best) while the others all-gathered (k, 4 + d + n)-word records — a hang or garbage.  Now
    the ranks exchange abo_cand_qei_eligible's answer first and all take the same form."""
    res = _run_agree([1, 0])                       # rank 1 cannot: BOTH take the plain loop
    assert [r[1] for r in res] == [["plain"], ["plain"]], res
    assert res[0][2] == res[1][2] == [11, 11]      # rank 1's record wins (score 1.5), global index 10·1 + 1
    res = _run_agree([1, 1])                       # all can: the block form everywhere
    assert [r[1] for r in res] == [["block"], ["block"]], res
    res = _run_agree([0, 1, 1])
    assert [r[1] for r in res] == [["plain"]] * 3, res


def _worker_xchg(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rec = bench.exchange_record(argparse.Namespace(backend="gloo"), True, rank)
    q.put((rank, rec))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_line_records_what_the_collective_layer_saw():
    """bench.py's config.exchange: backend, world size and the device of every rank as the process group reports
    them — the first multi-GPU line can then be checked for N ranks on N devices without reading logs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_xchg, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rec in res:
        assert rec["backend"] == "gloo" and rec["world_size"] == 2
        assert [r["rank"] for r in rec["ranks"]] == [0, 1] and [r["device"] for r in rec["ranks"]] == [0, 1]
        assert rec["distinct_devices"] == 2
