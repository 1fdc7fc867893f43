"""Child process of tests/test_gpu_threads.py: the threading contract of include/abo_hip.h ("distinct handles may be used from
different threads"; SURVEY.md §8(b) Threading) exercised the way a threaded Julia host would — the BO driver copies the model
every step (src/bayesian_opt.jl:116), and `copy` is a shared reference to the same device state:
  thread A   loops  copy(m) → append(copy, x_i, y_i) → mean_and_var(appended)        (one lineage: in-place appends to shared storage)
  thread B   loops  mean_and_var(m) / EI + top-k on the parent, and update() of an unrelated model
  thread C   (rounds 2 and 3) loops  copy(m) → append(copy, x'_i, y'_i) → mean_and_var   — a second lineage racing A for the same
             factor rows: the loser of the row claim refits into storage of its own (copy-on-write)
Round 1 (A and B): every result must equal the serial run's bit for bit.  Rounds 2 and 3 (A, B, C): B still bit for bit; an
appended model equals the serial one bit for bit when its append took the rows in place and to 1e-9 when it lost the claim and
refitted (another summation order).  No call may fail, and device memory must be flat between rounds 2 and 3 (the buffer pool neither
leaks nor grows).  Run as a child with a time limit: a regression here can be a dead-lock."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import abstractbayesopt.jl_amd as abo  # noqa: E402
from abstractbayesopt.jl_amd import synth  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    d, N, M = 4, 700, 3000
    X, y = synth.standardized_problem(N + 16, d, 0.05)
    Z = synth.points(2, M, d)
    X2, y2 = synth.standardized_problem(300, 3, 0.02)
    ker = abo.with_lengthscale(abo.Matern52Kernel(), 0.7)
    gp = abo.HipStandardGP(ker, 1e-3, n_max=N + 64)
    m = abo.update(gp, X[:N], y[:N])
    gp2 = abo.HipStandardGP(abo.with_lengthscale(abo.SqExponentialKernel(), 0.5), 1e-2)
    acq = abo.ExpectedImprovement(0.01, float(y[:N].min()))
    xa, ya = X[N:N + 8], y[N:N + 8]
    xc, yc = X[N + 8:N + 16], y[N + 8:N + 16]

    # serial reference
    ref_a = [abo.mean_and_var(abo.append(abo.copy(m), xa[i], ya[i]), Z[:500]) for i in range(8)]
    ref_c = [abo.mean_and_var(abo.append(abo.copy(m), xc[i], yc[i]), Z[:500]) for i in range(8)]
    ref_b = abo.mean_and_var(m, Z)
    _, ref_tv, ref_ti = abo.evaluate(acq, m, Z, k=50)
    ref_u = abo.mean_and_var(abo.update(gp2, X2, y2), X2[:64])

    errors, counts = [], {"A": 0, "B": 0, "C": 0}
    stop = threading.Event()

    def same(a, b):
        return all(np.array_equal(p, q) for p, q in zip(a, b))

    strict = {"on": True}
    refits = {"A": 0, "C": 0}

    def loop_append(tag, xs, ys, ref):
        try:
            i = 0
            while not stop.is_set():
                c = abo.copy(m)
                n = abo.append(c, xs[i % 8], ys[i % 8])
                got = abo.mean_and_var(n, Z[:500])
                if not same(got, ref[i % 8]):
                    close = all(np.max(np.abs(p - q)) <= 1e-9 for p, q in zip(got, ref[i % 8]))
                    refits[tag] += 1
                    if strict["on"] or not close:
                        errors.append(f"{tag}: appended model {i % 8} differs from the serial run" + ("" if close else " by more than 1e-9"))
                del n, c
                i += 1
                counts[tag] += 1
        except Exception as e:                       # a status != 0 surfaces here
            errors.append(f"{tag}: {type(e).__name__}: {e}")

    def loop_parent():
        try:
            while not stop.is_set():
                if not same(abo.mean_and_var(m, Z), ref_b):
                    errors.append("B: parent posterior differs")
                _, tv, ti = abo.evaluate(acq, m, Z, k=50)
                if not (np.array_equal(tv, ref_tv) and np.array_equal(ti, ref_ti)):
                    errors.append("B: parent selection differs")
                u = abo.update(gp2, X2, y2)
                if not same(abo.mean_and_var(u, X2[:64]), ref_u):
                    errors.append("B: unrelated model differs")
                del u
                counts["B"] += 1
        except Exception as e:
            errors.append(f"B: {type(e).__name__}: {e}")

    def round_(secs, racing):
        stop.clear()
        strict["on"] = not racing
        ts = [threading.Thread(target=loop_append, args=("A", xa, ya, ref_a)), threading.Thread(target=loop_parent)]
        if racing:
            ts.append(threading.Thread(target=loop_append, args=("C", xc, yc, ref_c)))
        for t in ts:
            t.start()
        time.sleep(secs)
        stop.set()
        for t in ts:
            t.join(timeout=60)
        return not any(t.is_alive() for t in ts)

    joined1 = round_(seconds, False)
    c1 = dict(counts)
    joined2 = round_(seconds, True)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    c2 = dict(counts)
    joined3 = round_(seconds, True)
    torch.cuda.synchronize()
    free2 = torch.cuda.mem_get_info()[0]
    joined1 = joined1 and joined3
    # the parent is untouched by all of it
    parent_ok = same(abo.mean_and_var(m, Z), ref_b)
    print(json.dumps({"errors": errors[:10], "n_errors": len(errors), "joined": bool(joined1 and joined2), "iterations_round1": c1,
                      "iterations_round2": {k: c2[k] - c1[k] for k in c2}, "iterations": counts, "appends_that_refitted": refits,
                      "free_after_round2": free1, "free_after_round3": free2, "parent_ok": bool(parent_ok)}))


if __name__ == "__main__":
    main()
