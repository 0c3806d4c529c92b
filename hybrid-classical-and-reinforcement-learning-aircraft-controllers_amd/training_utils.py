"""Trainer glue with the reference's names and signatures (learned_controllers/utils/training_utils.py:23-249,
learned_controllers/utils/pid_demonstrations.py:13-110), over the device-resident env.
"""
import os
import time
from typing import Optional, Tuple

import numpy as np
import torch
import yaml

from .rate_env import GpuRateVecEnv


def load_config(config_path: str) -> dict:
    """training_utils.py:144-155 -- plain YAML -> dict; the reference's config files load unchanged."""
    with open(config_path, "r") as f:
        return yaml.safe_load(f)


def normalize_config(config: dict) -> dict:
    """Bring either of the reference's two YAML schemas to the one `train_rate` reads.

    `train_rate.py`-style files (config/ppo_lstm.yaml, fast_mlp_training.yaml, quick_test.yaml ...) only get defaults for
    sections they omit (e.g. `lstm: {enabled: false}` without sizes).  `train_overnight.py`-style files
    (config/overnight_v2.yaml: `network`, `parallel`, `logging`, `checkpointing`, `approach`, `demonstrations`,
    `behavior_cloning`; train_overnight.py:53-193) are mapped onto the same keys, and their imitation settings are returned
    under `imitation` so the trainer can apply them."""
    c = dict(config)
    net = c.get("network", {})
    lstm = dict(c.get("lstm", {}))
    if "enabled" not in lstm:
        lstm["enabled"] = net.get("type", "lstm" if "lstm" in c else "mlp") == "lstm"
    lstm.setdefault("lstm_hidden_size", net.get("lstm", {}).get("hidden_size", 256))
    lstm.setdefault("n_lstm_layers", net.get("lstm", {}).get("n_layers", 2))
    lstm.setdefault("features_dim", 128)
    c["lstm"] = lstm
    mlp = dict(c.get("mlp", {}))
    mlp.setdefault("net_arch", net.get("mlp", {}).get("net_arch", [256, 128, 64]))
    c["mlp"] = mlp
    cur = dict(c.get("curriculum", {}))
    cur.setdefault("phases", [])
    cur.setdefault("enabled", bool(cur["phases"]) and c.get("approach", {}).get("use_curriculum", True))
    c["curriculum"] = cur
    env = dict(c.get("environment", {}))
    first = cur["phases"][0] if cur["phases"] else {}
    env.setdefault("difficulty", first.get("difficulty", "medium"))
    env.setdefault("command_type", first.get("command_type", "step"))
    env.setdefault("episode_length", 10.0)
    env.setdefault("dt", 0.02)
    c["environment"] = env
    tr = dict(c.get("training", {}))
    tr.setdefault("n_envs", c.get("parallel", {}).get("n_envs", 4))
    tr.setdefault("log_interval", c.get("logging", {}).get("log_interval", 10))
    tr.setdefault("eval_freq", c.get("evaluation", {}).get("eval_freq", 10000))
    tr.setdefault("save_freq", c.get("checkpointing", {}).get("save_freq", 50000))
    tr.setdefault("total_timesteps", sum(p["timesteps"] for p in cur["phases"]) or 1000000)
    c["training"] = tr
    paths = dict(c.get("paths", {}))
    paths.setdefault("model_save_dir", paths.get("model_dir", "runs/checkpoints"))
    paths.setdefault("tensorboard_log", "runs/tensorboard")
    paths.setdefault("best_model_path", paths.get("best_model", "runs/best_rate_controller"))
    c["paths"] = paths
    c.setdefault("seed", 42)
    if c.get("approach", {}).get("use_imitation"):
        demo, bc = c.get("demonstrations", {}), c.get("behavior_cloning", {})
        c["imitation"] = {"n_episodes": demo.get("n_episodes", 100), "difficulty": demo.get("difficulty", "medium"),
                          "save_path": demo.get("save_path"),
                          "epochs": bc.get("epochs", 10), "batch_size": bc.get("batch_size", 256),
                          "learning_rate": bc.get("learning_rate", 1e-3)}
    return c


def create_vec_env(config: dict, n_envs: int = 4, seed: Optional[int] = None, **kw) -> GpuRateVecEnv:
    """training_utils.py:49-69: same arguments; returns ONE device-resident vec-env instead of n_envs subprocesses.
    Env `i` is seeded `seed + i` like `make_env(config, rank=i, seed)` (:41)."""
    e = config["environment"]
    return GpuRateVecEnv(n_envs, e["difficulty"], e["episode_length"], e["dt"], e["command_type"], seed=seed, **kw)


def collect_pid_demonstrations(n_episodes: int = 100, difficulty: str = "medium", save_path: Optional[str] = None,
                               seed: int = 42, n_envs: Optional[int] = None, precision: str = "mixed",
                               sampling: str = "device") -> Tuple[np.ndarray, np.ndarray]:
    """pid_demonstrations.py:13-110: roll the rate PID (throttle 0.6, dt 0.02) in the env and record (obs_t, action_t).

    Here `n_envs` envs fly one episode each in parallel with the PID fused into the env-step kernel.  Unlike the
    reference -- which appends its re-used observation buffer un-copied and so stores one unique row --
    each pair holds the observation the action was computed from.
    """
    n = n_envs or n_episodes
    env = GpuRateVecEnv(n, difficulty, 10.0, 0.02, "step", seed=seed, precision=precision, sampling=sampling)
    obs = env.reset().clone()
    alive = torch.ones(n, dtype=torch.bool, device=env.device)
    obs_l, act_l = [], []
    for _ in range(int(10.0 / 0.02)):
        _, _, term, trunc = env.step_device(None, auto_reset=False)
        obs_l.append(obs[alive].clone())
        act_l.append(env.actions_taken[alive].clone())
        alive &= ~(term | trunc).bool()
        obs = env.obs.clone()
        if not bool(alive.any()):
            break
    observations = torch.cat(obs_l).cpu().numpy().astype(np.float32)
    actions = torch.cat(act_l).cpu().numpy().astype(np.float32)
    if save_path:
        os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
        np.savez_compressed(save_path, observations=observations, actions=actions)   # data-only file, no pickle
    return observations, actions


def behavior_cloning_pretrain(model, observations, actions, epochs=10, batch_size=256, lr=1e-3):
    """training_utils.py:158-213: MSE(mean_action, expert) with Adam on the actor trunk + action head.
    `model` is a RecurrentPPO / anything with `.policy`; recurrent policies are cloned from zero state."""
    policy = model.policy
    dev = next(policy.parameters()).device
    obs_t = torch.as_tensor(observations, dtype=torch.float32, device=dev)
    act_t = torch.as_tensor(actions, dtype=torch.float32, device=dev)
    params = [p for n, p in policy.named_parameters() if not n.startswith(("vf_net", "value_net", "lstm_critic", "log_std"))]
    opt = torch.optim.Adam(params, lr=lr)
    losses = []
    for _ in range(epochs):
        perm = torch.randperm(obs_t.shape[0], device=dev)
        total, nb = 0.0, 0
        for s in range(0, obs_t.shape[0], batch_size):
            idx = perm[s:s + batch_size]
            st = policy.initial_state(idx.numel(), dev)
            lat_pi, _, _ = policy._core(obs_t[idx], st)
            loss = torch.nn.functional.mse_loss(policy.action_net(lat_pi), act_t[idx])
            opt.zero_grad(set_to_none=False)
            loss.backward()
            opt.step()
            total += float(loss.detach()); nb += 1
        losses.append(total / max(nb, 1))
    return losses


@torch.no_grad()
def load_demonstrations(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """pid_demonstrations.py:113-127 for the data-only .npz this package writes (the reference pickles a dict)."""
    d = np.load(path)
    return d["observations"], d["actions"]


@torch.no_grad()
def run_final_evaluation(model, difficulty="hard", n_episodes=10, dt=0.02, seed=0, residual_scale=0.0, command_type="step"):
    """training_utils.py:216-249: deterministic policy, one episode per env, mean length and return.  Runs without autograd
    (the same fused inference path the rollouts use; with grad enabled the policy would fall back to the un-fused cells and
    build a 500-step graph through the recurrent state).  `command_type` as in the env: the periodic evaluation of a
    curriculum phase flies that phase's command type (random-walk increments are drawn in-kernel)."""
    env = GpuRateVecEnv(n_episodes, difficulty, 10.0, dt, command_type, seed=seed, precision="mixed", sampling="device",
                        residual_scale=residual_scale)
    obs = env.reset().clone()
    pol = model.policy
    st = pol.initial_state(n_episodes, env.device)
    start = torch.ones(n_episodes, device=env.device)
    alive = torch.ones(n_episodes, dtype=torch.bool, device=env.device)
    ret = torch.zeros(n_episodes, device=env.device); length = torch.zeros(n_episodes, device=env.device)
    for _ in range(int(10.0 / dt)):
        a, _, _, st = pol.step(obs, st, start, deterministic=True)
        start = torch.zeros_like(start)
        obs, rew, term, trunc = env.step_device(a, auto_reset=False)
        obs = obs.clone()
        ret += rew * alive; length += alive.float()
        alive &= ~(term | trunc).bool()
        if not bool(alive.any()):
            break
    return {"mean_length": float(length.mean()), "mean_reward": float(ret.mean()), "lengths": length.cpu().numpy(),
            "rewards": ret.cpu().numpy()}


# ---- callbacks: learned_controllers/utils/training_utils.py:72-156 (SB3's EvalCallback / CheckpointCallback there) -----------
class CheckpointCallback:
    """Saves `<save_path>/<name_prefix>_<num_timesteps>_steps.pt` every `save_freq` vec-env steps (SB3 counts callback calls,
    one per vec-env step: num_timesteps / n_envs)."""

    def __init__(self, save_freq: int, save_path: str, name_prefix: str = "rate_controller"):
        self.save_freq, self.save_path, self.name_prefix = max(int(save_freq), 1), save_path, name_prefix
        self._last = 0
        self.saved = []

    def __call__(self, model, stats):
        n_calls = model.num_timesteps // model.env.num_envs
        if n_calls // self.save_freq > self._last:
            self._last = n_calls // self.save_freq
            os.makedirs(self.save_path, exist_ok=True)
            path = os.path.join(self.save_path, f"{self.name_prefix}_{model.num_timesteps}_steps.pt")
            model.save(path)
            self.saved.append(path)


class EvalCallback:
    """Every `eval_freq` vec-env steps: `n_eval_episodes` deterministic episodes on a fresh env, results appended to
    `<log_path>/evaluations.npz` (arrays `timesteps`, `results`, `ep_lengths` -- the file recover_training.py:18-95 reads) and
    the best mean reward so far kept as `<best_model_save_path>/best_model.pt`."""

    def __init__(self, difficulty: str, best_model_save_path: str, log_path: str, eval_freq: int, n_eval_episodes: int = 10,
                 deterministic: bool = True, residual_scale: float = 0.0, writer=None, command_type: str = "step",
                 history=None):
        self.difficulty, self.command_type = difficulty, command_type
        self.best_dir, self.log_dir = best_model_save_path, log_path
        self.writer = writer                               # ProgressLogger: eval/mean_reward, eval/mean_ep_length
        self.residual_scale = residual_scale
        self.eval_freq, self.n_eval_episodes, self.deterministic = max(int(eval_freq), 1), n_eval_episodes, deterministic
        self._start, self._last, self.best_mean_reward = None, 0, -float("inf")
        # evaluation history: a curriculum rebuilds this callback per phase (fresh best-reward baseline, the phase's difficulty
        # and command type -- train_rate.py:150-170); the history list can be handed on so evaluations.npz keeps every phase
        self.timesteps, self.results, self.ep_lengths = history if history is not None else ([], [], [])

    @property
    def history(self):
        return self.timesteps, self.results, self.ep_lengths

    def __call__(self, model, stats):
        n_calls = model.num_timesteps // model.env.num_envs
        if self._start is None:                            # SB3 counts a callback's own calls: a phase starts at zero
            self._start = n_calls - model.cfg.n_steps
        if (n_calls - self._start) // self.eval_freq <= self._last:
            return
        self._last = (n_calls - self._start) // self.eval_freq
        ev = run_final_evaluation(model, difficulty=self.difficulty, n_episodes=self.n_eval_episodes,
                                  residual_scale=self.residual_scale, command_type=self.command_type)
        self.timesteps.append(model.num_timesteps); self.results.append(ev["rewards"]); self.ep_lengths.append(ev["lengths"])
        os.makedirs(self.log_dir, exist_ok=True)
        np.savez(os.path.join(self.log_dir, "evaluations.npz"), timesteps=np.array(self.timesteps),
                 results=np.array(self.results), ep_lengths=np.array(self.ep_lengths))
        if self.writer is not None:
            self.writer.scalars({"eval/mean_reward": float(ev["mean_reward"]), "eval/mean_ep_length": float(np.mean(ev["lengths"]))},
                                int(model.num_timesteps))
        if ev["mean_reward"] > self.best_mean_reward:
            self.best_mean_reward = ev["mean_reward"]
            os.makedirs(self.best_dir, exist_ok=True)
            model.save(os.path.join(self.best_dir, "best_model.pt"))


class ProgressLogger:
    """Per-iteration scalars, twice: JSON lines in `<tensorboard_log>/progress.jsonl`, and a TensorBoard event file
    `<tensorboard_log>/events.out.tfevents.*` under SB3's tag names -- the reference hands `tensorboard_log` to SB3
    (train_rate.py:144) and reads those files back in visualize/learning_curves.py:35-121.  TensorBoard is not in this
    image: tfevents.py writes the format directly."""
    TAGS = {"ep_rew_mean": "rollout/ep_rew_mean", "ep_len_mean": "rollout/ep_len_mean", "value_loss": "train/value_loss",
            "policy_loss": "train/policy_gradient_loss", "approx_kl": "train/approx_kl", "clip_frac": "train/clip_fraction",
            "grad_norm": "train/grad_norm", "mean_reward_per_step": "rollout/mean_reward_per_step"}

    def __init__(self, log_dir: str, tensorboard: bool = True):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "progress.jsonl")
        self._f = open(self.path, "a")
        self.events = None
        if tensorboard:
            from .tfevents import EventFileWriter
            self.events = EventFileWriter(log_dir)
        self._t0, self._steps0 = time.time(), None

    def scalars(self, values: dict, step: int):
        """Extra scalars under their full tag names (EvalCallback: eval/mean_reward, eval/mean_ep_length)."""
        if self.events is not None:
            self.events.add_scalars(values, step)
            self.events.flush()

    def __call__(self, model, stats):
        import json
        step = int(model.num_timesteps)
        self._f.write(json.dumps({"timesteps": step, **{k: float(v) for k, v in stats.items()}}) + "\n")
        self._f.flush()
        if self.events is None:
            return
        if self._steps0 is None:                           # timesteps at the start of the first logged iteration
            cfg, env = getattr(model, "cfg", None), getattr(model, "env", None)
            self._steps0 = step - cfg.n_steps * env.num_envs if cfg is not None and env is not None else 0
        tb = {tag: float(stats[k]) for k, tag in self.TAGS.items() if k in stats}
        policy, opt = getattr(model, "policy", None), getattr(model, "opt", None)
        if policy is not None:
            tb["train/entropy_loss"] = -float(policy.entropy().detach())
            tb["train/std"] = float(policy.log_std.detach().exp().mean())
        if opt is not None:
            tb["train/learning_rate"] = float(opt.param_groups[0]["lr"])
        tb["time/fps"] = (step - self._steps0) / max(time.time() - self._t0, 1e-9)
        self.scalars(tb, step)


class CallbackList:
    def __init__(self, callbacks):
        self.callbacks = list(callbacks)

    def __call__(self, model, stats):
        for cb in self.callbacks:
            cb(model, stats)


def create_callbacks(config: dict, eval_env=None, flight_logger=None, previous: Optional[CallbackList] = None) -> CallbackList:
    """training_utils.py:72-156: evaluation + checkpoint callbacks from the `paths` / `training` / `evaluation` sections.
    The evaluation flies `environment.difficulty` / `environment.command_type` AS THEY ARE WHEN THIS IS CALLED: the curriculum
    loop calls it once per phase after updating them, as the reference does (train_rate.py:150-170), so each phase is judged
    on its own task with a fresh best-reward baseline.  `previous` (the callbacks of the phase before) hands on the ONE progress
    logger / event file, the checkpoint schedule and the evaluation history.
    (`eval_env` is accepted for signature compatibility; evaluation builds its own device env.  The reference's optional
    TensorBoard flight-logging callback needs its tensorboard plugin, which is outside the hot path.)"""
    paths, tr, ev = config["paths"], config["training"], config.get("evaluation", {})
    env_cfg = config.get("environment", {})
    prev = {type(c).__name__: c for c in (previous.callbacks if previous is not None else [])}
    progress = prev.get("ProgressLogger") or ProgressLogger(paths["tensorboard_log"])
    checkpoints = prev.get("CheckpointCallback") or CheckpointCallback(tr["save_freq"], paths["model_save_dir"], "rate_controller")
    history = prev["EvalCallback"].history if "EvalCallback" in prev else None
    return CallbackList([
        EvalCallback(env_cfg.get("difficulty", "medium"), paths["best_model_path"], paths["best_model_path"], tr["eval_freq"],
                     ev.get("n_eval_episodes", 10), ev.get("deterministic", True), writer=progress,
                     command_type=env_cfg.get("command_type", "step"), history=history),
        checkpoints,
        progress])


def find_best_checkpoint(log_path: str):
    """recover_training.py:18-95: the evaluation with the highest mean reward in `<log_path>/evaluations.npz`
    -> (timesteps, mean_reward) or None."""
    f = os.path.join(log_path, "evaluations.npz")
    if not os.path.exists(f):
        return None
    d = np.load(f)
    means = d["results"].mean(axis=1)
    i = int(np.argmax(means))
    return int(d["timesteps"][i]), float(means[i])
