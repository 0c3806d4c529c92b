#!/usr/bin/env python3
"""The learned-controller building blocks one at a time, over the drop-in objects: what the reference's
learned_controllers/example_usage.py:16-181 walks through (env interaction, command generation, flight-envelope sampling,
reward behaviour, training configuration), plus the batched form of each where one exists.

    python examples/rate_env_usage.py               # needs an MI355X: there is no CPU fallback
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import hcrl_amd  # noqa: E402,F401
from hcrl_amd import rewards, samplers  # noqa: E402
from hcrl_amd.gym_env import RateControlEnv, ResidualRateControlEnv  # noqa: E402
from hcrl_amd.rate_env import GpuRateVecEnv  # noqa: E402
from hcrl_amd.training_utils import load_config, normalize_config  # noqa: E402
from hcrl_amd.train_rate import DEFAULT_CONFIG  # noqa: E402


def banner(title):
    print("\n" + "=" * 60 + f"\n{title}\n" + "=" * 60)


def env_interaction():
    banner("1. One environment, stepped by hand (RateControlEnv)")
    env = RateControlEnv(difficulty="medium", episode_length=5.0, dt=0.02, command_type="step", rng_seed=42)
    obs, info = env.reset(seed=42)
    print(f"observation {obs.shape}, action bounds {env.action_space.low} .. {env.action_space.high}")
    print(f"rate command (rad/s): {np.round(info['rate_command'], 3)}")
    total = 0.0
    for step in range(100):
        action = np.clip(np.array([2.0 * obs[6], -2.0 * obs[7], -2.0 * obs[8], 0.6], dtype=np.float32), [-1, -1, -1, 0], 1)
        obs, reward, terminated, truncated, info = env.step(action)          # a crude proportional law on the rate errors
        total += reward
        if step % 25 == 0:
            print(f"  step {step:3d}  rates {np.round(obs[0:3], 3)}  error {np.round(info['rate_error'], 3)}  reward {reward:+.3f}")
        if terminated or truncated:
            break
    print(f"return over {step + 1} steps: {total:.2f}   (altitude {info['altitude']:.1f} m, airspeed {info['airspeed']:.1f} m/s)")
    res = ResidualRateControlEnv(difficulty="medium", rng_seed=42)
    obs, info = res.reset(seed=42)
    obs, reward, *_rest, info = res.step(np.zeros(4, dtype=np.float32))      # zero residual = the PID alone
    print(f"residual env, zero correction: PID action {np.round(info['pid_action'], 3)} reward {reward:+.3f}")
    # the same thing for many environments: one launch per step
    vec = GpuRateVecEnv(4096, "medium", 5.0, 0.02, "step", seed=0, precision="mixed", sampling="device")
    vec.reset()
    for _ in range(100):
        vec.step_device(None)                                                # actions = None: the fused rate PID flies every env
    print(f"4096 envs x 100 PID steps: mean reward of the last step {float(vec.rewards_full.mean()):+.3f}")


def command_generation():
    banner("2. Rate commands (RateCommandGenerator)")
    gen = samplers.RateCommandGenerator(difficulty="medium", rng_seed=1)
    for axes in (1, 2, 3):
        cmd, desc = gen.generate_step_command(num_axes=axes)
        print(f"  {desc:28s} p={cmd[0]:+.2f} q={cmd[1]:+.2f} r={cmd[2]:+.2f} rad/s")
    freq, amps, desc = gen.generate_sine_command()
    print(f"  {desc:28s} amplitudes {np.round(amps, 2)}")
    cmd, desc = gen.generate_multi_axis_command()
    print(f"  {desc:28s} p={cmd[0]:+.2f} q={cmd[1]:+.2f} r={cmd[2]:+.2f} rad/s")


def envelope_sampling():
    banner("3. Initial conditions (FlightEnvelopeSampler)")
    sampler = samplers.FlightEnvelopeSampler(rng_seed=3)
    for k in range(4):
        ic = sampler.sample()
        print(f"  #{k}: V={ic['airspeed']:.1f} m/s  h={ic['altitude']:.0f} m  roll/pitch/yaw={np.round(np.degrees(ic['attitude']), 1)} deg")


def reward_behaviour():
    banner("4. Reward components (RateTrackingReward on the device)")
    fn = rewards.RateTrackingReward(w_tracking=1.0, w_smoothness=0.01, w_stability=0.1, w_oscillation=0.5)
    action, prev_action = np.array([0.1, 0.05, -0.02, 0.5]), np.array([0.08, 0.04, -0.03, 0.5])
    for name, errs in (("perfect tracking", (0.0, 0.0, 0.0)), ("small error", (0.1, 0.1, 0.05)), ("large error", (1.0, 0.8, 0.6)),
                       ("very large error", (3.0, 2.5, 2.0))):
        total, c = fn.compute(*errs, action, prev_action, airspeed=20.0, altitude=100.0, roll=0.1, pitch=0.05)
        print(f"  {name:17s} total {total:+8.4f} | tracking {c['tracking']:+8.4f} smoothness {c['smoothness']:+.4f} "
              f"stability {c['stability']:+.4f} oscillation {c['oscillation']:+.4f}")
    # batched: 100 000 logged steps scored in one launch
    n, T, dev = 1000, 100, "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.rand(s, device=dev, generator=g, dtype=torch.float64)          # noqa: E731
    flight = torch.stack([15 + 10 * rnd(T, n), 50 + 100 * rnd(T, n), 0.4 * rnd(T, n) - 0.2, 0.2 * rnd(T, n) - 0.1], 1)
    out = rewards.score_sequences(0.3 * (rnd(T, 3, n) - 0.5), 2 * rnd(T, 4, n) - 1, torch.zeros((4, n), dtype=torch.float64, device=dev),
                                  flight, 0.5 * (rnd(3, n) - 0.5), 0.02)
    print(f"  {T * n} random logged steps: mean total {float(out['tracking'].mean()):+.3f}, "
          f"settled fraction {float(out['settled'].float().mean()):.3f}")


def training_preview():
    banner("5. Training configuration")
    cfg = normalize_config(load_config(DEFAULT_CONFIG))
    print(f"  total timesteps {cfg['training']['total_timesteps']:,}  envs {cfg['training']['n_envs']}  curriculum {cfg['curriculum']['enabled']}")
    for ph in cfg["curriculum"]["phases"]:
        print(f"    phase {ph['name']:12s} {ph['difficulty']:7s} {ph['command_type']:7s} {ph['timesteps']:,} steps")
    print("  PPO: " + ", ".join(f"{k}={v}" for k, v in cfg["ppo"].items()))
    print("  LSTM: " + ", ".join(f"{k}={v}" for k, v in cfg["lstm"].items()))
    print("  train with: python train_rate.py --config <yaml> --bf16 [--callbacks]")


if __name__ == "__main__":
    env_interaction(); command_generation(); envelope_sampling(); reward_behaviour(); training_preview()
