"""Soak run: many model lifetimes (refit / append / resident grids / gradient models) in one process; device
memory in use must stay flat (pool bounded) and results must stay identical."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abstractbayesopt.jl_amd as abo
from abstractbayesopt.jl_amd import synth

def used_mb():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20

X, y = synth.standardized_problem(900, 4, 0.05)
Z = torch.from_numpy(synth.points(2, 50000, 4)).cuda()
gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.7), 1e-3, n_max=1024)
ref = None
marks = []
for it in range(300):
    m = abo.update(gp, X, y)
    acq = abo.ExpectedImprovement(0.01, float(y.min()))
    s, tv, ti = abo.evaluate(acq, m, Z, k=10)
    if ref is None:
        ref = (s.clone(), ti.clone())
    else:
        assert torch.equal(s, ref[0]) and torch.equal(ti, ref[1])
    c = abo.ResidentCandidates(m, Z)
    m2 = abo.append(m, synth.points(9, 1, 4, first=it)[0], 0.1)
    c.downdate(m2)
    if it % 10 == 0:
        g = abo.update(abo.GradientGP(abo.SqExponentialKernel(), 3, 0.1), synth.points(5, 40, 2), np.ones((40, 3)))
        abo.posterior_grad_cov(g, [[0.2, 0.3]])
        g2 = abo.append(abo.update(abo.GradientGP(abo.SqExponentialKernel(), 3, 0.1, n_max=64), synth.points(5, 40, 2), np.ones((40, 3))),
                        [0.3, 0.7], [1.0, 0.0, 0.0])
        abo.posterior_grad_mean(g2, [[0.2, 0.3]])
        # multi-device handle (three shards on this device): worker threads, exchange buffers, fantasy handles of q-EI
        grp = abo.update(abo.HipShardedGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.7), 1e-3, devices=(0, 0, 0), n_max=1024), X, y)
        Zh = synth.points(2, 3000, 4)
        cg = abo.ShardedCandidates(grp, Zh)
        cg.greedy_qei(grp, 3, 0.01, float(y.min()))
        abo.multigpu.append(grp, Zh[it % 3000], 0.2, cg)
        small = abo.update(abo.HipStandardGP(abo.Matern52Kernel(), 1e-6), synth.points(7, 30, 2), np.arange(30.0))   # fused small fit
        abo.posterior_var(small, Zh[:100, :2])
        # round 3: one-call optimize_acquisition (one-launch and lockstep refinement), the sharded one, the fused update + acquisition,
        # a gradient-enhanced group
        dom = abo.ContinuousDomain(np.zeros(4), np.ones(4))
        ucb = abo.UpperConfidenceBound(2.0)
        b1 = abo.optimize_acquisition_device(ucb, m, dom, 5000, 30, seed=it)
        os.environ["ABO_REFINE_LOCKSTEP_NP"] = "128"
        b2 = abo.optimize_acquisition_device(ucb, m, dom, 5000, 30, seed=it)
        del os.environ["ABO_REFINE_LOCKSTEP_NP"]
        b3 = abo.optimize_acquisition_device(ucb, grp, dom, 5000, 30, seed=it)
        assert np.array_equal(b1, b3) and np.all(np.isfinite(b2))
        abo.update_and_evaluate(acq, gp, X, y, Zh, k=10)
        gg = abo.update(abo.HipShardedGradientGP(abo.SqExponentialKernel(), 3, 0.1, devices=(0, 0), n_max=64), synth.points(5, 40, 2), np.ones((40, 3)))
        abo.multigpu.append(gg, [0.3, 0.7], [1.0, 0.0, 0.0])
    if it % 50 == 0:
        torch.cuda.synchronize()
        marks.append(used_mb())
        print(it, f"{marks[-1]:.0f} MiB in use", flush=True)
assert max(marks[1:]) - min(marks[1:]) < 64, marks
abo._lib.check(abo._lib.lib().abo_pool_trim(0))
print("after trim", f"{used_mb():.0f} MiB")
print("soak ok")
