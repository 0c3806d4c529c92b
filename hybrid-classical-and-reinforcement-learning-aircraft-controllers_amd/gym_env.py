"""`RateControlEnv`: the single-env gym-style API of learned_controllers/envs/rate_env.py:17-470 over the HIP path.

Constructor arguments, `reset(seed, options) -> (obs, info)`, `step(action) -> (obs, reward, terminated, truncated,
info)`, the `observation_space` / `action_space` bounds (:110-138), `sim.get_state()`, `rate_command`, `dt` and the
`info` keys (:421-433) follow the reference.  `gymnasium` is not a dependency: `Box` below carries what callers use.
Episodes are sampled on the host with the reference's three MT19937 streams (samplers.EpisodeStreams), so
`RateControlEnv(..., rng_seed=s).reset(seed=s)` reproduces the reference's episode for the same `s`.
Per-step `reward_components` are not materialised by the fused kernel (only the total), unlike rate_env.py:276-279.
"""
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import layout as L
from .flight_types import AircraftState
from .rate_env import GpuRateVecEnv
from .samplers import EpisodeStreams


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
                 precision: str = "f64", residual_scale: float = 0.0):
        self.difficulty, self.episode_length, self.dt, self.command_type = difficulty, episode_length, dt, command_type
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
        self.episode_rewards = {"total": 0.0}

    @property
    def rate_command(self) -> np.ndarray:
        return self._vec.e[L.FD_E_CMD_P:L.FD_E_CMD_R + 1, 0].to(torch.float64).cpu().numpy()

    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None) -> Tuple[np.ndarray, Dict[str, Any]]:
        if seed is not None:
            self._streams.reseed_env_rng(seed)                       # rate_env.py:167-168: ONLY env.rng is re-seeded
        self._vec.pool.copy_(torch.as_tensor(self._streams.next_record()[None, None]))
        obs = self._vec.reset()
        self.step_count, self.current_time = 0, 0.0
        self.episode_rewards = {"total": 0.0}
        return obs[0].cpu().numpy().copy(), self._get_info()

    def step(self, action) -> Tuple[np.ndarray, float, bool, bool, Dict[str, Any]]:
        a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, 4), device=self._vec.device)
        rw = None
        if self.command_type == "random":
            d = self._streams.random_walk_delta(self.dt)
            rw = torch.as_tensor(np.ascontiguousarray(d[:, None]), device=self._vec.device).to(self._vec.dtype)
        obs, _, term, trunc = self._vec.step_device(a, auto_reset=False, rw_delta=rw)
        self.step_count += 1
        self.current_time += self.dt
        reward = float(self._vec.rewards_full[0])
        self.episode_rewards["total"] += reward
        return obs[0].cpu().numpy().copy(), reward, bool(term[0]), bool(trunc[0]), self._get_info()

    def _get_info(self) -> Dict[str, Any]:
        v = self._vec
        x = v.x[:, 0].to(torch.float64).cpu().numpy()
        cmd = self.rate_command
        return {"time": self.current_time, "step": self.step_count, "position": x[0:3].copy(), "rate_command": cmd.copy(),
                "rate_error": cmd - x[9:12], "airspeed": float(np.linalg.norm(x[3:6])), "altitude": float(-x[2]),
                "is_settled": bool(v.e[L.FD_E_IS_SETTLED, 0] != 0)}

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
