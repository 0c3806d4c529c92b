#!/usr/bin/env python3
"""Task-quality table in the shape of the reference's README.md:40-44 (mean episode reward and survival time on
easy / medium / hard): the fused rate-PID demonstrator (default ControllerConfig gains, throttle 0.6) and, optionally, a
trained policy checkpoint, each over N episodes run in parallel on the GPU.

    python scripts/eval_table.py [--episodes 4096] [--checkpoint runs/checkpoints/final_model.pt]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcrl_amd  # noqa: E402,F401
from hcrl_amd.policy import RateLSTMPolicy  # noqa: E402
from hcrl_amd.rate_env import GpuRateVecEnv  # noqa: E402


@torch.no_grad()
def evaluate(difficulty, n, policy=None, seed=0, dt=0.02):
    env = GpuRateVecEnv(n, difficulty, 10.0, dt, "step", seed=seed, precision="mixed", sampling="device")
    obs = env.reset()
    alive = torch.ones(n, dtype=torch.bool, device=env.device)
    ret = torch.zeros(n, device=env.device); length = torch.zeros(n, device=env.device)
    if policy is not None:
        st = policy.initial_state(n, env.device)
        start = torch.ones(n, device=env.device)
    for _ in range(int(10.0 / dt)):
        if policy is None:
            _, rew, term, trunc = env.step_device(None, auto_reset=False)
        else:
            a, _, _, st = policy.step(obs, st, start, deterministic=True)
            start = torch.zeros_like(start)
            obs, rew, term, trunc = env.step_device(a, auto_reset=False)
        ret += rew * alive; length += alive.float()
        alive &= ~(term | trunc).bool()
        if not bool(alive.any()):
            break
    return float(ret.mean()), float(length.mean()) * dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=4096)
    ap.add_argument("--checkpoint", default=None)
    a = ap.parse_args()
    policy = None
    if a.checkpoint:
        policy = RateLSTMPolicy(compute_dtype=torch.bfloat16).cuda()
        policy.load_state_dict(torch.load(a.checkpoint, map_location="cuda", weights_only=True)["policy"])
        policy.prepare_inference()
    print(f"| difficulty | PID reward | PID survival (s) |" + (" RL reward | RL survival (s) |" if policy else ""))
    print("|---|---|---|" + ("---|---|" if policy else ""))
    for d in ("easy", "medium", "hard"):
        r, s = evaluate(d, a.episodes)
        row = f"| {d} | {r:+.1f} | {s:.2f} |"
        if policy:
            r2, s2 = evaluate(d, a.episodes, policy)
            row += f" {r2:+.1f} | {s2:.2f} |"
        print(row)


if __name__ == "__main__":
    main()
