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


def test_train_rate_two_ranks_share_weights_and_split_the_envs(tmp_path):
    """The REAL trainer under two ranks (what tests/test_ppo_distributed.py checks with a CPU test double): `train_rate.py` under
    `torch.distributed.run --nproc-per-node 2`, gloo, both ranks on the box's one GPU, tiny configuration, rollout graph + update
    graph captured, one flat gradient all-reduce per optimizer step.  After the curriculum's three phases every rank holds
    BIT-EQUAL parameters (same initial weights, same averaged gradients, same optimizer) although each flew a different env
    shard (different seeds => different first observations).  This is the only N >= 2 evidence a one-GPU box allows; the RCCL
    leg differs by the backend string (and is exercised in a one-rank group above)."""
    import yaml
    sys.path.insert(0, REPO)
    from hcrl_amd import train_rate
    cfg = yaml.safe_load(open(train_rate.DEFAULT_CONFIG))
    cfg["training"]["n_envs"] = 512
    cfg["ppo"].update(n_steps=8, n_epochs=2, n_minibatches=2)
    for ph in cfg["curriculum"]["phases"]:
        ph["timesteps"] = 2 * 512 * 8 * 3                    # whole-job steps: 3 iterations per phase on each of 2 ranks
    cfg["paths"]["model_save_dir"] = str(tmp_path / "ckpt")
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(cfg))
    rep = tmp_path / "ranks"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(REPO, "train_rate.py"), "--config", str(p), "--bf16", "--backend", "gloo",
           "--device-index", "0", "--rank-report", str(rep)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    a, b = (json.load(open(rep / f"rank{k}.json")) for k in (0, 1))
    assert a["world"] == b["world"] == 2 and (a["rank"], b["rank"]) == (0, 1)
    assert a["rollout_graph"] and b["rollout_graph"] and a["update_graph"] and b["update_graph"]
    assert a["num_timesteps"] == b["num_timesteps"] == 3 * 3 * 512 * 8            # per rank: three phases x three iterations
    assert a["env_seed"] != b["env_seed"] and a["first_obs_sha256"] != b["first_obs_sha256"]      # different env shards
    assert a["param_sha256"] == b["param_sha256"]                                   # ... one set of weights, bit for bit
    # and the weights moved: not the shared initialisation (a one-rank run of the same budget ends elsewhere)
    one = tmp_path / "one"
    train_rate.main(["--config", str(p), "--bf16", "--rank-report", str(one), "--timesteps-scale", "0.5"])
    c = json.load(open(one / "rank0.json"))
    assert c["num_timesteps"] == a["num_timesteps"] and c["param_sha256"] != a["param_sha256"]
