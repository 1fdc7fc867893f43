"""Threading contract (-m gpu): include/abo_hip.h "distinct handles may be used from different threads" — and `copy` (abo_retain,
a shared reference: src/bayesian_opt.jl:116 copies the model every BO step) must be as safe as a distinct handle.  The work runs in
a child process under a time limit (tests/threads_child.py): a regression can be a dead-lock."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_copies_appends_and_predictions_from_three_host_threads_equal_the_serial_run():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "threads_child.py"), "1.5"], capture_output=True, text=True,
                       timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["joined"], out
    assert out["n_errors"] == 0, out["errors"]
    assert out["parent_ok"]
    assert out["iterations_round1"]["A"] >= 3 and out["iterations_round1"]["B"] >= 3, out      # both made progress beside each other
    assert min(out["iterations_round2"].values()) >= 3, out
    # the third round allocates nothing the second one did not leave in the pool
    assert out["free_after_round3"] >= out["free_after_round2"] - (64 << 20), out
