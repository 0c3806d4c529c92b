"""`python recover_training.py [--checkpoint-step S] [--output PATH]` -- learned_controllers/recover_training.py:18-152:
pick the checkpoint whose evaluation scored best (`evaluations.npz`, written by `EvalCallback`), copy it to `--output`,
verify that it loads, and print how to continue from it (`train_rate.py --resume`).

Differences from the reference: the directories are arguments (`--checkpoint-dir`, `--eval-dir`; the reference hard-codes
`learned_controllers/models/...`), checkpoints are this trainer's `rate_controller_<steps>_steps.pt` files (`.zip`
archives in SB3's layout are accepted too), and verification reads the weights with `weights_only=True` instead of
unpickling a model.  Without an evaluation file the earliest checkpoint is used (the reference falls back to its 10k one).
"""
import argparse
import os
import re
import shutil
import sys

from .training_utils import find_best_checkpoint


def list_checkpoints(checkpoint_dir: str, prefix: str = "rate_controller"):
    """-> [(steps, path)] sorted by steps."""
    out = []
    if os.path.isdir(checkpoint_dir):
        for f in os.listdir(checkpoint_dir):
            m = re.fullmatch(rf"{re.escape(prefix)}_(\d+)_steps\.(pt|zip)", f)
            if m:
                out.append((int(m.group(1)), os.path.join(checkpoint_dir, f)))
    return sorted(out)


def recover_checkpoint(checkpoint_step: int, output_path: str, checkpoint_dir: str, prefix: str = "rate_controller") -> bool:
    """recover_training.py:56-95: copy `<prefix>_<step>_steps.*` to `output_path` and check that it loads."""
    found = [p for s, p in list_checkpoints(checkpoint_dir, prefix) if s == checkpoint_step]
    if not found:
        print(f"ERROR: Checkpoint not found: {os.path.join(checkpoint_dir, f'{prefix}_{checkpoint_step}_steps.pt')}")
        print("\nAvailable checkpoints:")
        for _s, p in list_checkpoints(checkpoint_dir, prefix):
            print(f"  - {os.path.basename(p)}")
        return False
    src = found[0]
    if os.path.dirname(output_path):
        os.makedirs(os.path.dirname(output_path), exist_ok=True)
    print(f"\nCopying checkpoint from {src}\n             to {output_path}")
    shutil.copy2(src, output_path)
    print("\nVerifying checkpoint...")
    try:
        import torch
        from .sb3_zip import is_sb3_zip, policy_kwargs_from_state_dict, read_sb3_zip
        if is_sb3_zip(output_path):
            sd, _ = read_sb3_zip(output_path)
        else:
            ck = torch.load(output_path, map_location="cpu", weights_only=True)
            sd = ck["policy"] if "policy" in ck else ck
        kw = policy_kwargs_from_state_dict(sd)
        print("Checkpoint loaded successfully!")
        print(f"  Policy: {'RecurrentPPO / MlpLstmPolicy' if kw['use_lstm'] else 'PPO / MlpPolicy'}, "
              f"{sum(v.numel() for v in sd.values())} parameters")
        return True
    except Exception as e:                                   # :93-95
        print(f"Error loading checkpoint: {e}")
        return False


def main(argv=None):
    ap = argparse.ArgumentParser(description="Recover training from best checkpoint")
    ap.add_argument("--checkpoint-step", type=int, default=None, help="Specific checkpoint step to recover (default: auto-select best)")
    ap.add_argument("--output", type=str, default="learned_controllers/models/recovered_model.pt", help="Output path for recovered model")
    ap.add_argument("--checkpoint-dir", type=str, default="learned_controllers/models/checkpoints")
    ap.add_argument("--eval-dir", type=str, default="learned_controllers/models/best_rate_controller")
    a = ap.parse_args(argv)
    print("\n" + "=" * 60 + "\nTRAINING RECOVERY SCRIPT\n" + "=" * 60)
    if a.checkpoint_step is None:
        best = find_best_checkpoint(a.eval_dir)
        ckpts = list_checkpoints(a.checkpoint_dir)
        if best is None:
            print(f"No evaluation file found at {os.path.join(a.eval_dir, 'evaluations.npz')}")
            if not ckpts:
                print("no checkpoints either\n" + "=" * 60 + "\nRECOVERY FAILED\n" + "=" * 60)
                sys.exit(1)
            step = ckpts[0][0]
            print(f"Using earliest checkpoint ({step} steps)...")
        else:
            print("=" * 60 + f"\nBest evaluation: step {best[0]} (reward: {best[1]:.2f})\n" + "=" * 60)
            # checkpoints and evaluations run on their own periods: take the checkpoint closest to the best evaluation
            step = min((s for s, _ in ckpts), key=lambda s: abs(s - best[0]), default=best[0])
    else:
        step = a.checkpoint_step
        print(f"\nUsing specified checkpoint: {step} steps")
    if recover_checkpoint(step, a.output, a.checkpoint_dir):
        print("\n" + "=" * 60 + "\nRECOVERY COMPLETE!\n" + "=" * 60)
        print(f"\nRecovered model saved to: {a.output}\n\nContinue training with:\n   python train_rate.py --config <config.yaml> --resume {a.output}")
        print("=" * 60)
        return step
    print("\n" + "=" * 60 + "\nRECOVERY FAILED\n" + "=" * 60)
    sys.exit(1)


if __name__ == "__main__":
    main()
