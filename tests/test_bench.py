"""bench.py itself: the launcher logic on CPU, and on the GPU the whole script on a reduced grid -- the CPU oracle leg on
the GPU step's own weights and Batch, the `parity_full_grid` figure it yields, and `--gpus 2` without a launcher."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_launch_command_spawns_one_rank_per_gpu():
    sys.path.insert(0, str(ROOT))
    import argparse

    import bench

    cmd = bench.launch_command(argparse.Namespace(gpus=4), ["--gpus", "4", "--steps", "3"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-5:] == [str(ROOT / "bench.py"), "--gpus", "4", "--steps", "3"]


def _run(args, env=None, timeout=900):
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                         env={**os.environ, **(env or {})}, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stderr[-3000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_with_oracle_parity_on_a_reduced_grid():
    """The default single-GPU run, on 181 x 360: one JSON line carrying roofline (plain and all-launch fractions),
    cpu_baseline from the oracle on the same weights / Batch, and the parity of the two."""
    d = _run(["--grid", "181x360", "--steps", "2", "--warmup", "1", "--cpu-budget", "600"])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "n/a"
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and 0 < r["frac_all_matrix_launches"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["full_grid"] is True
    p = d["parity_full_grid"]
    assert p["ok"] and p["grid"] == "180x360"
    assert p["fp32_engine_vs_oracle"] <= 1e-4 and p["bf16_engine_vs_oracle"] <= 5e-3


@pytest.mark.gpu
def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: spawns two ranks (here both on the one GPU, gloo-staged
    halos) and prints ONE JSON line for the sharded forecast."""
    d = _run(["--gpus", "2", "--grid", "181x360", "--steps", "2", "--warmup", "1"],
             env={"AURORA_BENCH_SAME_GPU": "1", "AURORA_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    # the transport self-test ran (the run would have ended with an `error` line otherwise), and the record explains itself:
    # per rank, the step time without a barrier and the time the launch stream stood still in the halo `wait`s
    pr = d["per_rank"]
    assert len(pr["step_ms"]) == len(pr["halo_wait_ms"]) == len(pr["compute_ms"]) == 2
    assert all(w >= 0 for w in pr["halo_wait_ms"]) and pr["max_rank_compute_ms"] == max(pr["compute_ms"]) > 0


@pytest.mark.gpu
def test_bench_falls_back_to_replicas_when_the_halo_self_test_fails():
    """A fabric on which the halo self-test fails on ANY rank: every rank learns of it (host-side agreement), the run measures
    independent replicas instead of ending without a record, and the line says so."""
    d = _run(["--gpus", "2", "--grid", "181x360", "--steps", "2", "--warmup", "1"],
             env={"AURORA_BENCH_SAME_GPU": "1", "AURORA_BENCH_BACKEND": "gloo", "AURORA_BENCH_BREAK_SELFTEST": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "per_rank" not in d
    assert "rank 1" in d["bands_error"] and "replica" in d["config"]["parallelism"]
