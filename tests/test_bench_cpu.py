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
    # headline 163.5 GF, C2 2.90 GF, C4 370.4 GF per step at 10 sweeps (2 d r b for each of W^T X and H X^T, 2 r^2 b per sweep and for H H^T)
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


STUB = '''import json, os, sys, time
rank = int(os.environ["RANK"])
with open(os.path.join(sys.argv[1], "env%d.json" % rank), "w") as f:
    json.dump({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                              "MASTER_ADDR", "MASTER_PORT")}, f)
mode = sys.argv[2]
if mode == "fail" and rank == 1:
    sys.exit(7)
if mode == "fail":
    time.sleep(60)          # a rank that would wait for the failed one forever
'''


def test_launch_ranks_hands_every_child_its_rank_and_returns_the_first_failure(tmp_path):
    """bench.launch_ranks with a stub script: environment per child; when one rank fails, the others are
    terminated and its exit code becomes the launcher's."""
    import time
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    assert bench.launch_ranks(3, [str(tmp_path), "ok"], script=str(stub)) == 0
    envs = [json.load(open(tmp_path / ("env%d.json" % k))) for k in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["LOCAL_WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1
    t0 = time.time()
    assert bench.launch_ranks(3, [str(tmp_path), "fail"], script=str(stub)) == 7
    assert time.time() - t0 < 30          # the sleeping ranks were terminated, not waited for


def test_core_shares_for_rank_pinning():
    from onmf_ontf_ndl_amd import dist as odist
    assert odist.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    cores = list(range(96))
    shares = [odist.share_of_cores(cores, k, 4) for k in range(4)]
    assert [len(s) for s in shares] == [24] * 4 and sorted(sum(shares, [])) == cores
    assert odist.share_of_cores([5, 1, 3], 1, 2) == [5]                 # sorted, near-equal, never empty
    assert odist.share_of_cores([0, 1], 0, 8) == [0, 1]                 # fewer cores than ranks: all of them
    if hasattr(os, "sched_setaffinity"):
        before = os.sched_getaffinity(0)
        try:
            got = odist.pin_to_local_cores(1, 2, None)
            assert got and set(got) <= before and os.sched_getaffinity(0) == set(got)
        finally:
            os.sched_setaffinity(0, before)


def test_whole_step_traffic_record_feeds_the_bench_line():
    """`step_hbm` of the bench line: bytes per step from the committed counter passes over this run's time per
    step, for the shape the passes were taken at only; the hash over the library's sources is stable."""
    import bench
    d, r, b = bench.WORKLOADS["headline"]
    rec = bench.measured_step_hbm(d, r, b, 1.5)
    assert rec is not None, "profiles/ holds no *_step_hbm_traffic.json for the headline shape"
    assert rec["unit"] == "GB/s" and rec["peak"] == bench.PEAK_HBM_GBPS
    assert abs(rec["achieved"] - rec["bytes_per_step"] / 1.5e-3 / 1e9) < 1e-6 * rec["achieved"]
    assert abs(rec["frac"] - rec["achieved"] / rec["peak"]) < 1e-12
    # a fused step cannot move less than X once, H once and the small matrices
    assert rec["bytes_per_step"] > rec["algorithmic_bytes_per_step"] > 4.0 * d * b
    assert isinstance(rec["sources_unchanged"], bool)
    assert bench.measured_step_hbm(d, r, b // 2, 1.5) is None          # another shape: no figure
    assert bench.csrc_sha256() == bench.csrc_sha256() and len(bench.csrc_sha256()) == 64
