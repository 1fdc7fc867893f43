"""The drop-in boundary exercised from a host that is neither Python nor PyTorch: tests/c_abi_harness.c (plain C,
host arrays, linked against libabo_hip.so and through it the system ROCm runtime) is compiled with gcc and run as a
FRESH child process — the shape of a Julia `ccall` host (integration/julia/HipStandardGP.jl)."""
import os
import subprocess

import pytest

from tests import c_harness

pytestmark = pytest.mark.gpu


def test_c_host_runs_the_reference_closed_forms_and_the_sharded_path(tmp_path):
    exe = c_harness.build()
    fixture = c_harness.write_fixture(str(tmp_path / "fixture.txt"))
    env = {k: v for k, v in os.environ.items() if not k.startswith(("PYTHON", "LD_PRELOAD"))}
    # a child process started with subprocess (fork + exec BEFORE the child touches the GPU): never an exec from a
    # GPU-initialised process
    r = subprocess.run([exe, fixture, "0"], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout)
    print(r.stderr)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "all checks passed" in r.stdout
    for name in ("kat1", "kat3", "kat4", "kat5", "kat6", "nlml_grad kat1", "nlml_grad acq", "acq", "optimize_acquisition", "objectives as terms", "gradient-enhanced model",
                 "retain / append / rollback", "greedy q-EI", "mgpu"):
        assert f"ok {name}" in r.stdout
    # the process ran on the system HIP runtime, not on PyTorch's bundled copy
    hip = [ln for ln in r.stdout.splitlines() if ln.startswith("hip_runtime=")][0]
    assert "/opt/rocm" in hip and "torch" not in hip, hip
    # RCCL (dlopen'ed librccl.so.1) initialised and carried the exchange at world size 1
    assert "exchange=rccl" in r.stdout, r.stdout


def test_c_host_exits_cleanly_with_live_handles_pool_and_communicator(tmp_path):
    """A host that never finalises its handles (a Julia session ending, a C host returning from main): live model, live
    two-shard group with its worker threads, an RCCL communicator, buffers in the library's pool.  The process must end with
    status 0 — exit handlers, the library's static state and the HIP runtime's teardown in whatever order the loader picks.
    Run ONCE, as a fresh child (round-2 record: a SIGSEGV under __cxa_finalize in an experimental build; the shipped
    library now stops touching the device once the process is exiting, csrc/abo_internal.h: exiting())."""
    exe = c_harness.build()
    fixture = c_harness.write_fixture(str(tmp_path / "fixture.txt"))
    env = {k: v for k, v in os.environ.items() if not k.startswith(("PYTHON", "LD_PRELOAD"))}
    r = subprocess.run([exe, fixture, "0", "exit_live"], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout)
    print(r.stderr)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    assert "exit_live: leaving with a live model" in r.stdout


def test_python_host_exits_cleanly_with_live_handles():
    """the same from the ctypes host: handles still referenced when the interpreter shuts down (module globals, a group, a
    resident candidate set) — their __del__ runs during finalisation or never; either way the child ends with status 0"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np, abstractbayesopt.jl_amd as abo\n"
        "from abstractbayesopt.jl_amd import synth\n"
        "X, y = synth.standardized_problem(300, 4, 0.02)\n"
        "Z = synth.points(2, 5000, 4)\n"
        "gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.7), 1e-3, n_max=320)\n"
        "m = abo.update(gp, X, y)\n"
        "c = abo.ResidentCandidates(m, Z)\n"
        "m2 = abo.append(m, Z[0], 0.1)\n"
        "g = abo.update(abo.HipShardedGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.7), 1e-3, devices=(0, 0)), X, y)\n"
        "s, tv, ti = abo.evaluate(abo.ExpectedImprovement(0.01, float(y.min())), g, Z, k=10)\n"
        "keep = [m, c, m2, g]\n"
        "print('leaving', int(ti[0]))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    assert "leaving" in r.stdout
