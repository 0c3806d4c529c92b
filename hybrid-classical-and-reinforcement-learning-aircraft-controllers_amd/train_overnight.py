"""`python train_overnight.py --config <overnight yaml> [--resume CKPT] [--skip-demos] [--skip-bc]` --
learned_controllers/train_overnight.py:86-228 over the HIP path: PID demonstrations (saved to `demonstrations.save_path`,
reused with --skip-demos) -> behaviour cloning (`bc_pretrained`) -> the curriculum phases with evaluation / checkpoint /
progress callbacks -> `final_model` -> final evaluation on `hard`.

The work is `train_rate.main` (which reads this YAML schema through `normalize_config`); this entry point adds the
reference's banner, its flags, and turns the callbacks on as train_overnight.py:53-83 always does.  One device-resident
vec-env replaces the SubprocVecEnv workers; `parallel.n_envs` is the env count.
"""
import argparse
from datetime import datetime

from . import train_rate
from .training_utils import load_config, normalize_config


def main(argv=None):
    ap = argparse.ArgumentParser(description="Overnight training: imitation + curriculum")
    ap.add_argument("--config", type=str, required=True, help="Config file path")
    ap.add_argument("--resume", type=str, default=None, help="Resume from checkpoint")
    ap.add_argument("--skip-demos", action="store_true", help="Skip demo collection (use existing)")
    ap.add_argument("--skip-bc", action="store_true", help="Skip behavior cloning")
    ap.add_argument("--bf16", action="store_true", help="policy GEMMs in bf16 (fp32 accumulate)")
    ap.add_argument("--precision", default="mixed")
    a = ap.parse_args(argv)
    cfg = normalize_config(load_config(a.config))
    total = sum(p["timesteps"] for p in cfg["curriculum"]["phases"]) or cfg["training"]["total_timesteps"]
    print("=" * 60 + f"\nOVERNIGHT TRAINING\nStarted: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}\n" + "=" * 60)
    print(f"\nTotal steps: {total:,}")
    print(f"Envs: {cfg['training']['n_envs']} on one device (the reference estimates {total / 1100 / 3600:.1f} hours at its 1100 FPS)")
    args = ["--config", a.config, "--callbacks", "--precision", a.precision]
    for flag, on in (("--skip-demos", a.skip_demos), ("--skip-bc", a.skip_bc), ("--bf16", a.bf16)):
        if on:
            args.append(flag)
    if a.resume:
        args += ["--resume", a.resume]
    train_rate.main(args)
    print("=" * 60 + f"\nTRAINING COMPLETE!\nFinished: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}\n" + "=" * 60)
    print(f"  Final model: {cfg['paths']['model_save_dir']}/final_model.pt (+ final_model.zip)")
    print(f"  Best model: {cfg['paths']['best_model_path']}")
    print(f"  Scalars: {cfg['paths']['tensorboard_log']} (progress.jsonl + events.out.tfevents.*)")


if __name__ == "__main__":
    main()
