"""CPU tests of bench.py's launch decision (no GPU is touched before it): a plain `python bench.py --gpus N` — the command
shape the driver uses for N = 1 — must reach the in-library multi-device branch for N > 1 instead of exiting, and the
driver's torchrun launch must keep the one-process-per-GPU path."""
import argparse
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    a = argparse.Namespace(gpus=1, single_process=False)
    a.__dict__.update(kw)
    return a


def test_plain_launch_with_several_gpus_takes_the_library_path():
    for n in (2, 4, 8):
        assert bench.plan_launch(_args(gpus=n), {}) == "library"
    assert bench.plan_launch(_args(gpus=1), {}) == "ranks"                       # the BENCH line: unchanged path
    assert bench.plan_launch(_args(gpus=1, single_process=True), {}) == "library"


def test_torchrun_launch_keeps_one_process_per_gpu():
    env = {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}
    assert bench.plan_launch(_args(gpus=8), env) == "ranks"
    assert bench.plan_launch(_args(gpus=1), {"WORLD_SIZE": "1", "RANK": "0"}) == "ranks"
    with pytest.raises(SystemExit):
        bench.plan_launch(_args(gpus=4), env)                                    # rank count and --gpus disagree
    with pytest.raises(SystemExit):
        bench.plan_launch(_args(gpus=8, single_process=True), env)


def test_plain_multi_gpu_argv_reaches_the_multi_device_branch_not_sys_exit():
    """the real entry point as a child process on this GPU-less machine: it must get past argument handling into
    run_single_process and stop only where the devices are counted — with the message of that branch"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a machine with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "--share-device rehearses 2 shards on GPU 0" in r.stderr, r.stderr[-800:]
    assert "torch.distributed.run" not in r.stderr
