"""`python train_with_imitation.py [--residual] [--n-demos N] [--bc-epochs E] [--rl-steps S]` -- the reference's
learned_controllers/train_with_imitation.py:51-199 over the device path: PID demonstrations (the rate PID fused into the env
kernel) -> behaviour cloning of an MLP policy ([128, 128], the reference's net_arch) -> PPO fine-tuning, optionally on the
residual env (`ResidualRateControlEnv`: the policy corrects the PID, residual_rate_env.py:99-157) -> quick evaluation.
Flags, defaults and the PPO hyper-parameters (:96-111) follow the reference; `--n-envs` defaults to a GPU-sized batch and
`--reward-scale` / `--n-minibatches` are this package's large-batch extensions (configs/training/cfg4_easy_16384.yaml).
"""
import argparse
import os

import torch

from .policy import RateLSTMPolicy
from .ppo import PPOConfig, RecurrentPPO
from .rate_env import GpuRateVecEnv
from .training_utils import (CallbackList, CheckpointCallback, EvalCallback, behavior_cloning_pretrain,
                             collect_pid_demonstrations, load_demonstrations, run_final_evaluation)

RESIDUAL_SCALE = 0.3                                      # ResidualRateControlEnv default (residual_rate_env.py:36)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train with imitation learning")
    ap.add_argument("--n-demos", type=int, default=200, help="Number of demo episodes")
    ap.add_argument("--bc-epochs", type=int, default=15, help="Behavior cloning epochs")
    ap.add_argument("--rl-steps", type=int, default=500000, help="RL fine-tuning steps")
    ap.add_argument("--n-envs", type=int, default=4096, help="Parallel environments (one device-resident vec-env)")
    ap.add_argument("--residual", action="store_true", help="Use residual RL")
    ap.add_argument("--difficulty", type=str, default="medium", help="Difficulty level")
    ap.add_argument("--seed", type=int, default=42, help="Random seed")
    ap.add_argument("--skip-bc", action="store_true", help="Skip behavior cloning")
    ap.add_argument("--model-dir", type=str, default="runs/imitation_trained")
    ap.add_argument("--demo-path", type=str, default="runs/pid_demos.npz")
    ap.add_argument("--reward-scale", type=float, default=0.02)
    ap.add_argument("--n-minibatches", type=int, default=4)
    ap.add_argument("--n-steps", type=int, default=64, help="rollout length per env (the reference: 1024 with 8 envs)")
    a = ap.parse_args(argv)
    print(f"imitation training: difficulty={a.difficulty} residual={a.residual} envs={a.n_envs}")
    os.makedirs(a.model_dir, exist_ok=True)
    scale = RESIDUAL_SCALE if a.residual else 0.0

    if not a.skip_bc:                                     # step 1: demonstrations (:77-90)
        if os.path.exists(a.demo_path):
            observations, actions = load_demonstrations(a.demo_path)
        else:
            observations, actions = collect_pid_demonstrations(n_episodes=a.n_demos, difficulty=a.difficulty,
                                                               save_path=a.demo_path, seed=a.seed)
        print(f"demonstrations: {len(observations)} (observation, action) pairs")
        if a.residual:
            actions = actions * 0.0                       # the residual policy starts from "no correction": clone zeros
    env = GpuRateVecEnv(a.n_envs, a.difficulty, 10.0, 0.02, "step", seed=a.seed, precision="mixed", sampling="device",
                        residual_scale=scale)             # step 2 (:92-95)
    policy = RateLSTMPolicy(use_lstm=False, mlp_net_arch=(128, 128))                      # step 3: PPO("MlpPolicy") (:96-111)
    cfg = PPOConfig(learning_rate=3e-4, n_steps=a.n_steps, n_epochs=5, gamma=0.99, gae_lambda=0.95, clip_range=0.2,
                    ent_coef=0.01, n_minibatches=a.n_minibatches, reward_scale=a.reward_scale)
    model = RecurrentPPO(env, policy, cfg, seed=a.seed)
    if not a.skip_bc:                                     # step 4 (:113-127)
        losses = behavior_cloning_pretrain(model, observations, actions, epochs=a.bc_epochs, batch_size=256, lr=1e-3)
        print("BC losses:", [round(float(l), 5) for l in losses])
        model.save(os.path.join(a.model_dir, "bc_pretrained.pt"))
    callbacks = CallbackList([                            # step 5 (:129-150); frequencies in vec-env steps
        EvalCallback(a.difficulty, os.path.join(a.model_dir, "best"), os.path.join(a.model_dir, "logs"),
                     eval_freq=max(25000 // a.n_envs, 64), n_eval_episodes=5, residual_scale=scale),
        CheckpointCallback(max(100000 // a.n_envs, 256), os.path.join(a.model_dir, "checkpoints"), "imitation_model")])
    model.learn(a.rl_steps, log_interval=10, callback=callbacks)
    model.save(os.path.join(a.model_dir, "final_model.pt"))
    ev = run_final_evaluation(model, difficulty=a.difficulty, n_episodes=256, residual_scale=scale)   # (:160-190), 256 episodes
    print(f"evaluation ({a.difficulty}{', residual' if a.residual else ''}): mean reward {ev['mean_reward']:.1f}, "
          f"mean length {ev['mean_length']:.0f} steps")
    return ev


if __name__ == "__main__":
    main()
