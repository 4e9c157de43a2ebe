"""bench.py's own launch logic, without a GPU: `--gpus N` starts N ranks itself, hands each its rank
through the environment, and reports failure (no JSON line) when the ranks cannot run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_step_flops_match_the_survey_figures():
    import bench
    # SURVEY: headline 163.5 GF, C2 2.90 GF, C4 370.4 GF per step at 10 sweeps
    assert abs(bench.step_flops(1024, 256, 65536, 10) / 1e9 - 163.5) < 0.1
    assert abs(bench.step_flops(1024, 64, 8192, 10) / 1e9 - 2.90) < 0.01
    assert abs(bench.step_flops(4096, 256, 65536, 10) / 1e9 - 370.4) < 0.1


def test_gpus_n_starts_its_own_ranks_and_fails_loudly_without_gpus(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: this checks the no-GPU failure path")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                          "--warmup", "0", "--workload", "c1"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0                        # both ranks refuse to run without a GPU ...
    assert not any(line.startswith("{") for line in res.stdout.splitlines())      # ... and no line is printed
    assert "needs a GPU" in res.stderr


def test_a_process_group_of_another_size_is_refused(tmp_path):
    """`--gpus 4` inside a 2-rank environment must not report a line for another GPU count."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import bench
    args = bench.parse_args(["--gpus", "4"])
    assert args.gpus == 4
    # one rank of a 2-rank group, host backend (no GPU here): init works, the size check refuses
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "refusing" in res.stderr
    assert not any(line.startswith("{") for line in res.stdout.splitlines())
