"""`RateControlEnv`: the single-env gym-style API of learned_controllers/envs/rate_env.py:17-470 over the HIP path.

Constructor arguments, `reset(seed, options) -> (obs, info)`, `step(action) -> (obs, reward, terminated, truncated,
info)`, the `observation_space` / `action_space` bounds (:110-138), `sim.get_state()`, `rate_command`, `dt` and the
`info` keys (:421-433) follow the reference.  `gymnasium` is not a dependency: `Box` below carries what callers use.
Episodes are sampled on the host with the reference's three MT19937 streams (samplers.EpisodeStreams), so
`RateControlEnv(..., rng_seed=s).reset(seed=s)` reproduces the reference's episode for the same `s`.
`info["reward_components"]` and the per-component `episode_rewards` (rate_env.py:141-149,276-279,421-433): the fused env step
materialises only the total, so each `step` scores its own transition a second time with `fdyn_rate_reward_seq_*` (T = n = 1:
the tracker state the step started from, the action it applied, the flight condition it ended in) -- one extra small launch
per step of a host-paced single env; `reward_components=False` drops it.
"""
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import layout as L
from .flight_types import AircraftState
from .rate_env import GpuRateVecEnv
from .rewards import COMPONENTS, params_block, score_sequences
from .samplers import EpisodeStreams

# rate_env.py:141-149 -- the keys the reference accumulates per episode ("crash_penalty" is listed there and never added to)
_EPISODE_REWARD_KEYS = COMPONENTS + ("settle_bonus", "crash_penalty")


class Box:
    def __init__(self, low, high, dtype=np.float32):
        self.low, self.high, self.dtype = np.asarray(low, dtype), np.asarray(high, dtype), dtype
        self.shape = self.low.shape

    def sample(self, rng=np.random):
        return rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class _SimView:
    """`env.sim.get_state()` as used by pid_demonstrations.py:57 and eval scripts."""

    def __init__(self, env):
        self._env = env

    def get_state(self) -> AircraftState:
        v = self._env._vec
        x = v.x[:, 0].to(torch.float64).cpu().numpy()
        u, vv, w = x[3:6]
        st = AircraftState.from_vector(x, time=float(v.e[L.FD_E_TIME, 0]))
        st.airspeed, st.altitude = float(np.sqrt(u * u + vv * vv + w * w)), float(-x[2])
        return st


class RateControlEnv:
    metadata = {"render_modes": ["human"], "render_fps": 50}

    def __init__(self, difficulty: str = "medium", episode_length: float = 10.0, dt: float = 0.02,
                 command_type: str = "step", render_mode: Optional[str] = None, rng_seed: Optional[int] = None,
                 precision: str = "f64", residual_scale: float = 0.0, reward_components: bool = True):
        self.difficulty, self.episode_length, self.dt, self.command_type = difficulty, episode_length, dt, command_type
        self._want_components = reward_components
        self.render_mode = render_mode
        self._streams = EpisodeStreams(difficulty, command_type, rng_seed)
        # residual_scale > 0: `step` takes residuals and the fused rate PID supplies the baseline (ResidualRateControlEnv below)
        self._vec = GpuRateVecEnv(1, difficulty, episode_length, dt, command_type, seed=rng_seed, precision=precision,
                                  sampling="device", residual_scale=residual_scale)
        self._vec.pool = torch.zeros((1, 1, L.FD_NR), dtype=torch.float64, device=self._vec.device)
        self._vec.pool_depth = 1
        self.sim = _SimView(self)
        self.max_steps = int(episode_length / dt)
        self.step_count, self.current_time = 0, 0.0
        lo = [-10.0] * 3 + [-10.0] * 3 + [-20.0] * 3 + [5.0, 0.0, -np.pi, -np.pi / 2, -np.pi] + [-1.0, -1.0, -1.0, 0.0]
        hi = [10.0] * 3 + [10.0] * 3 + [20.0] * 3 + [50.0, 500.0, np.pi, np.pi / 2, np.pi] + [1.0, 1.0, 1.0, 1.0]
        self.observation_space = Box(lo, hi, np.float32)
        self.action_space = Box([-1.0, -1.0, -1.0, 0.0], [1.0, 1.0, 1.0, 1.0], np.float32)
        self.episode_rewards = {k: 0.0 for k in _EPISODE_REWARD_KEYS}
        # the settle timer word holds seconds in fp64 env words and a settled-step count in fp32 ones (fdyn_core.hpp
        # env_reward): scored with dt = 1 against the count at which the reference's accumulated timer reaches 0.2 s there
        t, k = 0.0, 0
        while t < 0.2:
            t, k = t + dt, k + 1
        self._settle_steps = k
        self._reward_params = params_block() if self._vec.e.dtype == torch.float64 else params_block(min_settle_time=k - 0.5)

    @property
    def rate_command(self) -> np.ndarray:
        return self._vec.e[L.FD_E_CMD_P:L.FD_E_CMD_R + 1, 0].to(torch.float64).cpu().numpy()

    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None) -> Tuple[np.ndarray, Dict[str, Any]]:
        if seed is not None:
            self._streams.reseed_env_rng(seed)                       # rate_env.py:167-168: ONLY env.rng is re-seeded
        self._vec.pool.copy_(torch.as_tensor(self._streams.next_record()[None, None]))
        obs = self._vec.reset()
        self.step_count, self.current_time = 0, 0.0
        self.episode_rewards = {k: 0.0 for k in _EPISODE_REWARD_KEYS}
        return obs[0].cpu().numpy().copy(), self._get_info()

    def step(self, action) -> Tuple[np.ndarray, float, bool, bool, Dict[str, Any]]:
        a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, 4), device=self._vec.device)
        rw = None
        if self.command_type == "random":
            d = self._streams.random_walk_delta(self.dt)
            rw = torch.as_tensor(np.ascontiguousarray(d[:, None]), device=self._vec.device).to(self._vec.dtype)
        v = self._vec
        if self._want_components:                                    # the tracker state this step starts from
            rstate = v.e[L.FD_E_PERR_P:L.FD_E_IS_SETTLED + 1].clone()
            prev = v.e[L.FD_E_PREV_AIL:L.FD_E_PREV_THR + 1].clone()
        obs, _, term, trunc = v.step_device(a, auto_reset=False, rw_delta=rw)
        self.step_count += 1
        self.current_time += self.dt
        reward = float(v.rewards_full[0])
        terminated, truncated = bool(term[0]), bool(trunc[0])
        comps = self._score_step(rstate, prev, terminated, truncated) if self._want_components else None
        return obs[0].cpu().numpy().copy(), reward, terminated, truncated, self._get_info(comps)

    def _score_step(self, rstate, prev, terminated, truncated) -> Dict[str, float]:
        """rate_env.py:255-295: RateTrackingReward.compute's components, the settle bonus and the crash penalty of the step
        just taken, from the device (`fdyn_rate_reward_seq_*`, T = n = 1), accumulated the way the reference accumulates."""
        v, dt_ = self._vec, self._vec.e.dtype
        x = v.x[:, 0].to(dt_)
        cmd = v.e[L.FD_E_CMD_P:L.FD_E_CMD_R + 1]                                        # after _update_command (:342)
        errs = (cmd[:, 0] - x[9:12]).reshape(1, 3, 1)
        act = v.e[L.FD_E_PREV_AIL:L.FD_E_PREV_THR + 1].reshape(1, 4, 1)                 # prev_action = action.copy() (:282)
        flight = torch.stack([torch.linalg.vector_norm(x[3:6]), -x[2], x[6], x[7]]).reshape(1, 4, 1)
        fp64 = dt_ == torch.float64
        out = score_sequences(errs, act, prev, flight, cmd, self.dt if fp64 else 1.0, self._reward_params, rstate)
        c = out["components"][0, :, 0].to(torch.float64).cpu().numpy()
        comps = {name: float(c[k]) for k, name in enumerate(COMPONENTS)}
        comps["total"] = float(out["tracking"][0, 0])
        comps["tracking_error_mse"] = float((errs.to(torch.float64) ** 2).sum() / 3.0)   # rewards.py:76,131
        settle = float(out["settle"][0, 0]) * (1.0 if fp64 else self.dt)                 # bonus = multiplier * dt (rewards.py:212)
        for k in COMPONENTS:                                                             # :276-278 (only keys episode_rewards holds)
            self.episode_rewards[k] += comps[k]
        self.episode_rewards["settle_bonus"] += settle                                   # :279
        comps["settle_bonus"] = settle        # not in the reference's dict (it only accumulates it); kept for logging
        if terminated and not truncated:
            comps["crash_penalty"] = -100.0                                              # :289-294
        return comps

    def _get_info(self, reward_components: Optional[Dict[str, float]] = None) -> Dict[str, Any]:
        v = self._vec
        x = v.x[:, 0].to(torch.float64).cpu().numpy()
        cmd = self.rate_command
        info = {"time": self.current_time, "step": self.step_count, "position": x[0:3].copy(), "rate_command": cmd.copy(),
                "rate_error": cmd - x[9:12], "airspeed": float(np.linalg.norm(x[3:6])), "altitude": float(-x[2]),
                "is_settled": bool(v.e[L.FD_E_IS_SETTLED, 0] != 0)}
        if reward_components is not None:
            info["reward_components"] = reward_components                                # :430-431
        return info

    def render(self):
        pass

    def close(self):
        pass


class ResidualRateControlEnv:
    """learned_controllers/envs/residual_rate_env.py:17-182: the action is a correction ADDED to the rate PID's output
    (`clip(PID + residual_scale * action)`), with a small bonus for small corrections.

    One fused launch per step: the kernel evaluates the rate PID (throttle command 0.6, PID dt = env dt, as :115-125),
    adds the scaled residual in float32, clips, steps and adds the bonus (`rate_env_step_kernel`, residual mode).  The
    `info` keys of :147-149 come from that launch (`combined_action` = the action the kernel applied) and from a
    `RateAgent` drop-in fed the same command and state just before it (`pid_action`, what the reference's own
    `self.pid_agent` returns; its fp32 PID state evolves bit-identically to the fused one)."""
    metadata = {"render_modes": ["human"], "render_fps": 50}

    def __init__(self, difficulty: str = "medium", episode_length: float = 10.0, dt: float = 0.02, command_type: str = "step",
                 render_mode: Optional[str] = None, rng_seed: Optional[int] = None, residual_scale: float = 0.3,
                 precision: str = "f64"):
        from .agents import RateAgent
        from .flight_types import ControllerConfig
        if not residual_scale > 0.0:
            raise ValueError("residual_scale must be > 0 (use RateControlEnv for direct surface actions)")
        self.residual_scale = residual_scale
        self.base_env = RateControlEnv(difficulty=difficulty, episode_length=episode_length, dt=dt, command_type=command_type,
                                       render_mode=render_mode, rng_seed=rng_seed, precision=precision,
                                       residual_scale=residual_scale)
        self.pid_config = ControllerConfig()
        self.pid_agent = RateAgent(self.pid_config)
        self.observation_space = self.base_env.observation_space
        self.action_space = Box([-1.0, -1.0, -1.0, -1.0], [1.0, 1.0, 1.0, 1.0], np.float32)
        self.last_pid_action = np.zeros(4)

    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None) -> Tuple[np.ndarray, Dict[str, Any]]:
        obs, info = self.base_env.reset(seed=seed, options=options)
        self.pid_agent.reset()
        self.last_pid_action = np.zeros(4)
        info["pid_action"] = self.last_pid_action.copy()
        return obs, info

    def step(self, residual_action) -> Tuple[np.ndarray, float, bool, bool, Dict[str, Any]]:
        from .flight_types import ControlCommand, ControlMode
        residual_action = np.asarray(residual_action, dtype=np.float32)
        p_cmd, q_cmd, r_cmd = self.base_env.rate_command
        command = ControlCommand(mode=ControlMode.RATE, roll_rate=p_cmd, pitch_rate=q_cmd, yaw_rate=r_cmd, throttle=0.6)
        surfaces = self.pid_agent.compute_action(command, self.base_env.sim.get_state(), dt=self.base_env.dt)
        pid_action = np.array([surfaces.aileron, surfaces.elevator, surfaces.rudder, surfaces.throttle], dtype=np.float32)
        self.last_pid_action = pid_action.copy()
        obs, reward, terminated, truncated, info = self.base_env.step(residual_action)
        info["pid_action"] = pid_action
        info["residual_action"] = residual_action * np.float32(self.residual_scale)
        info["combined_action"] = self.base_env._vec.actions_taken[0].cpu().numpy().copy()
        return obs, reward, terminated, truncated, info

    @property
    def sim(self):
        return self.base_env.sim

    @property
    def rate_command(self):
        return self.base_env.rate_command

    def render(self):
        return self.base_env.render()

    def close(self):
        self.base_env.close()
