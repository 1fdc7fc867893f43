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
    for name in ("kat1", "kat3", "kat4", "kat5", "kat6", "acq", "retain / append / rollback", "mgpu"):
        assert f"ok {name}" in r.stdout
    # the process ran on the system HIP runtime, not on PyTorch's bundled copy
    hip = [ln for ln in r.stdout.splitlines() if ln.startswith("hip_runtime=")][0]
    assert "/opt/rocm" in hip and "torch" not in hip, hip
    # RCCL (dlopen'ed librccl.so.1) initialised and carried the exchange at world size 1
    assert "exchange=rccl" in r.stdout, r.stdout
