#!/usr/bin/env python3
"""Train the learned rate controller: same CLI and config schema as the reference's learned_controllers/train_rate.py
(:62-209) -- `--config`, `--no-lstm` -- with the curriculum loop of :150-180, on the device-resident env.

    python train_rate.py --config <yaml>                                (1 GPU; repo-root wrapper)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_rate.py ...   (8 GPUs)
"""
import argparse
import os

import numpy as np
import torch
import torch.distributed as dist

from .policy import RateLSTMPolicy
from .ppo import PPOConfig, RecurrentPPO
from .training_utils import (behavior_cloning_pretrain, collect_pid_demonstrations, create_callbacks, create_vec_env, load_config,
                             normalize_config,
                             run_final_evaluation)

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "training", "ppo_lstm.yaml")


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train learned rate controller")
    ap.add_argument("--config", type=str, default=DEFAULT_CONFIG)
    ap.add_argument("--no-lstm", action="store_true")
    ap.add_argument("--bc-pretrain", type=int, default=0, help="behaviour-cloning epochs on fused-PID demonstrations")
    ap.add_argument("--timesteps-scale", type=float, default=1.0)
    ap.add_argument("--precision", default="mixed")
    ap.add_argument("--set", action="append", default=[], metavar="SECTION.KEY=VALUE",
                    help="override a config entry, e.g. --set ppo.learning_rate=1e-3 --set training.n_envs=4096")
    ap.add_argument("--bf16", action="store_true", help="run the policy GEMMs in bf16 (fp32 accumulate)")
    ap.add_argument("--resume", type=str, default=None, help="checkpoint (RecurrentPPO.save) to continue from")
    ap.add_argument("--skip-demos", action="store_true",
                    help="train_overnight.py:91: reuse the demonstrations at `demonstrations.save_path` instead of flying new ones")
    ap.add_argument("--skip-bc", action="store_true", help="train_overnight.py:92: no behaviour cloning even if the config asks for it")
    ap.add_argument("--callbacks", action="store_true",
                    help="periodic evaluation (best model, evaluations.npz) and checkpoints per the config's eval_freq / save_freq")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend under torch.distributed.run (nccl = RCCL over xGMI; gloo for rehearsals)")
    ap.add_argument("--device-index", type=int, default=-1, help="force every rank onto this GPU (one-GPU rehearsal of N > 1)")
    ap.add_argument("--rank-report", type=str, default=None, metavar="DIR",
                    help="every rank writes DIR/rank<r>.json at the end: digests of its parameters and of its env shard's first "
                         "observations, its env seed, whether the rollout / update graphs ran (what a data-parallel run must "
                         "agree and differ on)")
    args = ap.parse_args(argv)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between ranks on this driver stack
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = args.device_index if args.device_index >= 0 else local
    torch.cuda.set_device(dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    rank = dist.get_rank() if world > 1 else 0

    config = normalize_config(load_config(args.config))
    for item in args.set:                                   # SECTION.KEY=VALUE overrides (YAML-typed values)
        import yaml as _yaml
        key, val = item.split("=", 1)
        sec, k = key.split(".", 1)
        v = _yaml.safe_load(val)
        if isinstance(v, str):                              # YAML 1.1 reads '1e-3' as a string
            try:
                v = float(v)
            except ValueError:
                pass
        config.setdefault(sec, {})[k] = v
    seed = config.get("seed", 42)
    np.random.seed(seed)
    n_envs = config["training"]["n_envs"]
    use_lstm = not (args.no_lstm or not config["lstm"]["enabled"])
    policy = RateLSTMPolicy(features_dim=config["lstm"]["features_dim"], lstm_hidden_size=config["lstm"]["lstm_hidden_size"],
                            n_lstm_layers=config["lstm"]["n_lstm_layers"], use_lstm=use_lstm,
                            mlp_net_arch=tuple(config["mlp"]["net_arch"]),
                            compute_dtype=torch.bfloat16 if args.bf16 else None)
    env = create_vec_env(config, n_envs=n_envs, seed=seed + rank * n_envs, precision=args.precision)
    model = RecurrentPPO(env, policy, PPOConfig.from_dict(config["ppo"]), seed=seed)
    first_obs = model.obs.detach().cpu().numpy().tobytes() if args.rank_report else None

    if args.resume:
        model.load(args.resume)
        if rank == 0:
            print(f"resumed from {args.resume} at {model.num_timesteps} timesteps")
    callback = create_callbacks(config) if (args.callbacks and rank == 0) else None

    imit = config.get("imitation")                          # train_overnight-style files carry their own imitation settings
    if (args.bc_pretrain or (imit and not args.resume and not args.skip_bc)) and rank == 0:
        from_cfg = bool(imit) and not args.bc_pretrain
        demo_path = imit.get("save_path") if from_cfg else None
        if demo_path and not demo_path.endswith(".npz"):     # the reference pickles to `*.pkl`; this package writes data-only .npz
            demo_path += ".npz"
        if from_cfg and args.skip_demos and demo_path and os.path.exists(demo_path):       # train_overnight.py:120-123
            from .training_utils import load_demonstrations
            print(f"Loading existing demos from {demo_path}")
            obs, acts = load_demonstrations(demo_path)
        else:
            obs, acts = collect_pid_demonstrations(n_episodes=imit["n_episodes"] if from_cfg else 2048,
                                                   difficulty=imit["difficulty"] if from_cfg else "medium", seed=seed,
                                                   save_path=demo_path)
        kw = dict(epochs=args.bc_pretrain) if args.bc_pretrain else dict(epochs=imit["epochs"], batch_size=imit["batch_size"],
                                                                         lr=imit["learning_rate"])
        print("BC losses:", behavior_cloning_pretrain(model, obs, acts, **kw))
        if from_cfg:                                        # train_overnight.py:181-183
            os.makedirs(config["paths"]["model_save_dir"], exist_ok=True)
            model.save(os.path.join(config["paths"]["model_save_dir"], "bc_pretrained.pt"))
    if world > 1:
        from .ppo import broadcast_parameters
        broadcast_parameters(model.policy)

    if config["curriculum"]["enabled"]:
        for phase in config["curriculum"]["phases"]:
            config["environment"]["difficulty"] = phase["difficulty"]
            config["environment"]["command_type"] = phase["command_type"]
            env = create_vec_env(config, n_envs=n_envs, seed=seed + rank * n_envs, precision=args.precision)
            model.set_env(env)
            if callback is not None:                        # train_rate.py:163-167: callbacks rebuilt on the phase's task
                callback = create_callbacks(config, previous=callback)
            if rank == 0:
                print(f"=== phase {phase['name']}: {phase['difficulty']}/{phase['command_type']} {phase['timesteps']} steps")
            model.learn(int(phase["timesteps"] * args.timesteps_scale), log_interval=config["training"]["log_interval"],
                        callback=callback)
    else:
        model.learn(int(config["training"]["total_timesteps"] * args.timesteps_scale),
                    log_interval=config["training"]["log_interval"], callback=callback)

    if rank == 0:
        os.makedirs(config["paths"]["model_save_dir"], exist_ok=True)
        model.save(os.path.join(config["paths"]["model_save_dir"], "final_model.pt"))
        model.save_sb3_zip(os.path.join(config["paths"]["model_save_dir"], "final_model"))   # train_rate.py:353-355 layout
        final = run_final_evaluation(model, difficulty=config["environment"]["difficulty"],
                                     command_type=config["environment"]["command_type"])
        print(f"final evaluation ({config['environment']['difficulty']}/{config['environment']['command_type']}):",
              {k: v for k, v in final.items() if k.startswith("mean")})
    if args.rank_report:
        import hashlib
        import json
        os.makedirs(args.rank_report, exist_ok=True)
        h = hashlib.sha256()
        for name, t in sorted(model.policy.state_dict().items()):
            h.update(name.encode())
            h.update(t.detach().cpu().contiguous().numpy().tobytes())
        rep = {"rank": rank, "world": world, "param_sha256": h.hexdigest(), "env_seed": seed + rank * n_envs,
               "first_obs_sha256": hashlib.sha256(first_obs).hexdigest(), "num_timesteps": int(model.num_timesteps),
               "rollout_graph": model._graph is not None, "update_graph": getattr(model, "_update_graph", None) is not None,
               "collective_backend": (args.backend if world > 1 else None), "device_index": dev_index}
        with open(os.path.join(args.rank_report, f"rank{rank}.json"), "w") as f:
            json.dump(rep, f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
