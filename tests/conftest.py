import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the suite drives the abo_test_* building blocks: it loads the TEST build of the library (libabo_hip_test.so = the shipped objects
# with api.hip compiled under -DABO_TEST_HOOKS; abstractbayesopt.jl_amd/_lib.py).  bench.py, smoke() and the plain-C host of
# tests/c_abi_harness.c run on the shipped libabo_hip.so.
os.environ.setdefault("ABO_LIB_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (*.so are git-ignored): compile the HIP library and the C
    oracle once (hipcc cross-compiles without a GPU) so that the ABI tests have something to load."""
    lib = os.path.join(ROOT, "abstractbayesopt.jl_amd", "lib", "libabo_hip.so")
    lib_t = os.path.join(ROOT, "abstractbayesopt.jl_amd", "lib", "libabo_hip_test.so")
    orc = os.path.join(ROOT, "oracle", "_build", "libgp_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(lib_t) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """achieved errors of the GPU parity tests → gpurun_out/parity_r06.json (tests/parity_record.py)"""
    try:
        from tests import parity_record
        parity_record.dump()
    except Exception:
        pass
