"""Race / determinism screen of the int8-residue engine: the same posterior variance over and over at several sizes (every call must
return the same bits), printed as one hash per size."""
import hashlib, sys
import numpy as np
sys.path.insert(0, ".")
import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for N, d, M in ((300, 3, 20000), (1000, 4, 30000), (2304, 8, 40000), (4100, 8, 70000), (8192, 8, 70000)):
    X, y = synth.standardized_problem(N, d, 0.03)
    Z = synth.points(2, M, d)
    m = abo.update(abo.HipStandardGP(1.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.9), 1e-3, contraction="int8"), X, y)
    ref = abo.posterior_var(m, Z)
    bad = 0
    for r in range(reps):
        if r % 8 == 7:      # a fresh model now and then: new residue planes of W, recycled pool buffers
            m = abo.update(abo.HipStandardGP(1.0 * abo.with_lengthscale(abo.Matern52Kernel(), 0.9), 1e-3, contraction="int8"), X, y)
        v = abo.posterior_var(m, Z)
        bad += int(not np.array_equal(v, ref))
    print(f"N={N} M={M}: {reps} repetitions, {bad} differing, sha {hashlib.sha256(ref.tobytes()).hexdigest()[:16]}", flush=True)
    assert bad == 0
