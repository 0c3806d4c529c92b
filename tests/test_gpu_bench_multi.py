"""GPU: `bench.py --gpus N` really runs N ranks -- rehearsed with two ranks sharing the one GPU of the box
(gloo rendezvous, `--device-index 0`); the RCCL path differs by the backend string only."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--batch", "8192", "--ppo-steps", "4", "--ppo-epochs", "1",
          "--ppo-minibatches", "2"]


def _bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *COMMON, *extra], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                   # stdout carries exactly ONE line, the JSON
    return json.loads(lines[0])


def test_bench_gpus_2_self_launches_two_ranks_and_times_the_collective():
    one = _bench("--gpus", "1", "--no-extras")
    two = _bench("--gpus", "2", "--backend", "gloo", "--device-index", "0")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["steps"] == 4 and two["scaling"] == "weak" and two["repeats"] >= 3
    # two ranks time-share ONE GPU here, so the whole-job rate is about the single-rank rate (on two GPUs it doubles):
    # the check is that the job really processed 2 x batch env-steps per step
    assert two["value"] > 0.6 * one["value"], (one["value"], two["value"])
    tr = two["train"]
    assert tr["collective"]["n_ranks_in_group"] == 2 and tr["collective"]["bytes"] == 1830089 * 4
    assert tr["collective"]["calls_timed"] == 3 * tr["optimizer_steps_per_iteration"]
    assert tr["collective"]["allreduce_us_per_optimizer_step"] > 0 and tr["value"] > 0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", *COMMON], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode != 0 and b"WORLD_SIZE=1" in p.stderr and p.stdout.strip() == b""


def test_bench_single_rank_group_runs_the_train_leg_on_rccl():
    """The RCCL calls of the N > 1 line on the one GPU this box has: a ONE-rank `nccl` process group (communicator init,
    barriers, the MAX all-reduce of the timing, the flat 7.3 MB gradient all-reduce between the update graphs), with the
    hipGraph captures running beside the process group's watchdog thread."""
    one = _bench("--gpus", "1", "--backend", "nccl", "--single-rank-group")
    assert one["n_gpus"] == 1 and "error" not in one["train"], one["train"]
    col = one["train"]["collective"]
    assert col["backend"] == "nccl" and col["n_ranks_in_group"] == 1 and col["bytes"] == 1830089 * 4
    assert col["calls_timed"] == 3 * one["train"]["optimizer_steps_per_iteration"] and col["allreduce_us_per_optimizer_step"] > 0
