"""Child process of tests/test_gpu_multigpu.py::test_collective_faults_*: drives the in-library RCCL exchange (one rank: the
test box has one GPU) through the fault-injection hook of csrc/mgpu.hip (ABO_MGPU_FAULT) and prints one JSON line per stage.
Run as a child because an aborted communicator set stays aborted for the rest of the process (that is the behaviour under
test), and because a regression here is a HANG: the parent gives this process a time limit instead of the test suite one."""
import json
import os
import sys
import time

import numpy as np

os.environ["ABO_MGPU_EXCHANGE"] = "rccl"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import abstractbayesopt.jl_amd as abo  # noqa: E402
from abstractbayesopt.jl_amd import synth  # noqa: E402


def main():
    d, N, M, k = 3, 200, 4000, 50
    X, y = synth.standardized_problem(N, d, 0.02)
    Z = synth.points(2, M, d)
    ker = abo.with_lengthscale(abo.Matern52Kernel(), 0.6)
    one = abo.update(abo.HipStandardGP(ker, 1e-3, device=0), X, y)
    grp = abo.update(abo.HipShardedGP(ker, 1e-3, devices=(0,), n_max=N + 16), X, y)
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    _, tv1, ti1 = abo.evaluate(acq, one, Z, k=k)
    out = {"transport_before": grp.exchange(), "note_before": grp.exchange_note()}

    def same():
        _, tv, ti = abo.evaluate(acq, grp, Z, k=k)
        return bool(np.array_equal(tv, tv1) and np.array_equal(ti, ti1))

    out["clean_call_equal"] = same()

    # 1. a shard says "not ready" in the vote: a status, nothing enqueued, the communicators stay
    os.environ["ABO_MGPU_FAULT"] = "ready:0"
    t0 = time.time()
    try:
        same()
        out["ready_fault"] = "no error"
    except abo.AboError as e:
        out["ready_fault"] = str(e)
    out["ready_fault_s"] = time.time() - t0
    del os.environ["ABO_MGPU_FAULT"]
    out["transport_after_ready_fault"] = grp.exchange()
    out["call_after_ready_fault_equal"] = same()

    # 2. (mode = collective) a shard fails between the vote and its all-gather, or (mode = stall) its collective never
    #    completes: bounded wait → ncclCommAbort → host exchange, and THIS call still returns the right selection
    mode = sys.argv[1] if len(sys.argv) > 1 else "collective"
    os.environ["ABO_MGPU_FAULT"] = f"{mode}:0"
    if mode == "stall":
        os.environ["ABO_MGPU_TIMEOUT_MS"] = "1500"
    t0 = time.time()
    try:
        out["faulted_call_equal"] = same()
    except abo.AboError as e:
        out["faulted_call_error"] = str(e)
    out["faulted_call_s"] = time.time() - t0
    del os.environ["ABO_MGPU_FAULT"]
    out["transport_after_fault"] = grp.exchange()
    out["note_after_fault"] = grp.exchange_note()
    out["call_after_fault_equal"] = same()
    # greedy q-EI exchanges once per pick: it runs on after the fall-back
    cg = abo.ShardedCandidates(grp, Z[:1500])
    _, idx, _ = cg.greedy_qei(grp, 3, 0.01, float(y.min()))
    out["qei_after_fault"] = [int(v) for v in idx]
    # a NEW group on the same device list inherits the fall-back (the list's communicators are gone for this process)
    grp2 = abo.update(abo.HipShardedGP(ker, 1e-3, devices=(0,)), X, y)
    out["new_group_transport"] = grp2.exchange()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
