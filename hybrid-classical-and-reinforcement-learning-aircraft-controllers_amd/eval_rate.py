"""`python eval_rate.py --model <checkpoint.pt> [--compare-pid]` -- the reference's learned_controllers/eval_rate.py
command line (:266-345: same flags) over the device path: every episode of an evaluation flies in parallel in one
`GpuRateVecEnv`, metrics come from `fdyn_rate_metrics_*`.  `--model` takes a checkpoint written by `RecurrentPPO.save`
(train_rate.py) or a Stable-Baselines3-layout .zip (weights read with weights_only=True; the pickled parts of such an
archive are never decoded, sb3_zip.py).
Extra flags: --seed, --precision, --pid-only (no model), --pid-true-dt (hand the PID the real env dt).
"""
import argparse

import torch

from .eval_metrics import aggregate_metrics, compare_metrics, evaluate_learned_controller, evaluate_pid_controller
from .policy import RateLSTMPolicy


def load_policy(path: str, device="cuda", bf16: bool = True) -> RateLSTMPolicy:
    from .sb3_zip import is_sb3_zip, policy_kwargs_from_state_dict, read_sb3_zip
    if is_sb3_zip(path):                        # Stable-Baselines3 archive layout: policy.pth read with weights_only=True
        sd, _meta = read_sb3_zip(path, device)
    else:
        ck = torch.load(path, map_location=device, weights_only=True)
        sd = ck["policy"] if "policy" in ck else ck
    kw = policy_kwargs_from_state_dict(sd)      # network sizes from the tensor shapes
    use_lstm = kw["use_lstm"]
    pol = RateLSTMPolicy(compute_dtype=torch.bfloat16 if bf16 else None, **kw).to(device)
    pol.load_state_dict(sd)
    pol.eval()
    if use_lstm and bf16:
        pol.prepare_inference()
    return pol


def main(argv=None):
    ap = argparse.ArgumentParser(description="Evaluate learned rate controller")
    ap.add_argument("--model", type=str, default=None, help="Path to learned model (RecurrentPPO.save checkpoint)")
    ap.add_argument("--n-episodes", type=int, default=10)
    ap.add_argument("--difficulty", type=str, default="medium", choices=["easy", "medium", "hard"])
    ap.add_argument("--compare-pid", action="store_true", help="Compare with PID baseline")
    ap.add_argument("--stochastic", action="store_true", help="Use stochastic policy (default: deterministic)")
    ap.add_argument("--episode-length", type=float, default=10.0)
    ap.add_argument("--command-type", type=str, default="step", choices=["step", "ramp", "sine", "random"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--precision", default="mixed", choices=["f64", "mixed", "f32"])
    ap.add_argument("--fp32-policy", action="store_true", help="run the policy in fp32 instead of bf16")
    ap.add_argument("--pid-only", action="store_true")
    ap.add_argument("--pid-true-dt", action="store_true",
                    help="give the PID the env dt instead of ControllerConfig.rate_loop_dt (what eval_rate.py:200 does)")
    a = ap.parse_args(argv)
    if not a.pid_only and a.model is None:
        ap.error("--model is required (or --pid-only)")
    kw = dict(n_episodes=a.n_episodes, difficulty=a.difficulty, episode_length=a.episode_length,
              command_type=a.command_type, seed=a.seed, precision=a.precision)
    learned_avg = None
    if not a.pid_only:
        print(f"\nEvaluating Learned Controller: {a.model}")
        print(f"Episodes: {a.n_episodes}, Difficulty: {a.difficulty}, Command: {a.command_type}")
        pol = load_policy(a.model, bf16=not a.fp32_policy)
        _, learned_avg = evaluate_learned_controller(pol, deterministic=not a.stochastic, **kw)
        learned_avg.print_summary("Learned Controller")
    if a.compare_pid or a.pid_only:
        print("\nEvaluating PID Baseline Controller")
        print(f"Episodes: {a.n_episodes}, Difficulty: {a.difficulty}, Command: {a.command_type}")
        _, pid_avg = evaluate_pid_controller(pid_dt=0.02 if a.pid_true_dt else None, **kw)
        pid_avg.print_summary("PID Controller")
        if learned_avg is not None:
            compare_metrics(learned_avg, pid_avg, name_a="Learned", name_b="PID")
    print("\nEvaluation complete!")


if __name__ == "__main__":
    main()
