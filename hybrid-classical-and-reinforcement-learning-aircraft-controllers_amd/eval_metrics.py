"""Rate-control evaluation: metrics and the PID-vs-learned comparison, computed on the device.

Host mirror of learned_controllers/eval/metrics.py (RateControlMetrics :8-93, MetricsCalculator :95-362,
compare_metrics :365-424) and of the two evaluation loops + aggregation of learned_controllers/eval_rate.py:25-263.
The reference runs one Python episode at a time and computes the metrics in NumPy; here all episodes of an evaluation
run side by side in one `GpuRateVecEnv`, the trajectories are recorded in device memory ([T][word][N]) and
`fdyn_rate_metrics_*` (csrc/eval_kernels.hip) reduces every episode in one launch.
"""
from dataclasses import dataclass, fields
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib, layout as L
from .config import cascade_consts
from .flight_types import ControllerConfig

FIELD_ORDER = ("settling_time_roll", "settling_time_pitch", "settling_time_yaw",
               "overshoot_roll", "overshoot_pitch", "overshoot_yaw",
               "steady_state_error_roll", "steady_state_error_pitch", "steady_state_error_yaw",
               "rise_time_roll", "rise_time_pitch", "rise_time_yaw",
               "control_smoothness", "tracking_rmse", "success", "episode_length", "total_reward")
assert len(FIELD_ORDER) == L.FD_NM


@dataclass
class RateControlMetrics:
    """metrics.py:8-40; same fields, defaults and order (= the FD_M_* rows of the kernel's output)."""
    settling_time_roll: float = 0.0
    settling_time_pitch: float = 0.0
    settling_time_yaw: float = 0.0
    overshoot_roll: float = 0.0
    overshoot_pitch: float = 0.0
    overshoot_yaw: float = 0.0
    steady_state_error_roll: float = 0.0
    steady_state_error_pitch: float = 0.0
    steady_state_error_yaw: float = 0.0
    rise_time_roll: float = 0.0
    rise_time_pitch: float = 0.0
    rise_time_yaw: float = 0.0
    control_smoothness: float = 0.0
    tracking_rmse: float = 0.0
    success: Union[bool, float] = False
    episode_length: float = 0.0
    total_reward: float = 0.0

    def to_dict(self) -> Dict:
        return {f: getattr(self, f) for f in FIELD_ORDER}

    @classmethod
    def from_vector(cls, v: Sequence[float], success_as_rate: bool = False) -> "RateControlMetrics":
        m = cls()
        for name, val in zip(FIELD_ORDER, v):
            setattr(m, name, float(val))
        m.success = float(m.success) if success_as_rate else bool(m.success)
        return m

    def print_summary(self, name: str = "Controller"):
        bar = "=" * 60
        print(f"\n{bar}\n{name} Performance Metrics\n{bar}")
        print(f"Settling Time (s):  Roll={self.settling_time_roll:.3f}, Pitch={self.settling_time_pitch:.3f}, "
              f"Yaw={self.settling_time_yaw:.3f}")
        print(f"Overshoot (%):      Roll={self.overshoot_roll:.1f}, Pitch={self.overshoot_pitch:.1f}, "
              f"Yaw={self.overshoot_yaw:.1f}")
        print(f"Rise Time (s):      Roll={self.rise_time_roll:.3f}, Pitch={self.rise_time_pitch:.3f}, "
              f"Yaw={self.rise_time_yaw:.3f}")
        print(f"Steady-State Error: Roll={self.steady_state_error_roll:.4f}, Pitch={self.steady_state_error_pitch:.4f}, "
              f"Yaw={self.steady_state_error_yaw:.4f} rad/s")
        print(f"Tracking RMSE:      {self.tracking_rmse:.4f} rad/s")
        print(f"Control Smoothness: {self.control_smoothness:.4f}")
        print(f"Success:            {self.success}")
        print(f"Total Reward:       {self.total_reward:.2f}\n{bar}\n")


class MetricsCalculator:
    """metrics.py:95-116.  `compute_metrics` keeps the reference's one-episode NumPy signature; `compute_batch` is the
    device path it wraps (N episodes, one launch)."""

    def __init__(self, settling_threshold: float = 0.05, settling_duration: float = 0.2, dt: float = 0.02):
        self.settling_threshold = settling_threshold
        self.settling_duration = settling_duration
        self.dt = dt

    @property
    def settle_steps(self) -> int:
        return int(self.settling_duration / self.dt)                     # metrics.py:209

    def compute_batch(self, times: torch.Tensor, rates: torch.Tensor, commands: torch.Tensor, actions: torch.Tensor,
                      rewards: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """times [T] f64; rates, commands [T,3,N] (f64 or f32); actions [T,N,4] f32; rewards [T,N]; lengths [N] int32.
        Returns [FD_NM, N] float64 (rows in FIELD_ORDER)."""
        lib = _lib.load()
        dev = _lib.require_gpu()
        T, three, n = rates.shape
        assert three == 3 and commands.shape == (T, 3, n) and actions.shape == (T, n, 4) and rewards.shape == (T, n)
        assert lengths.shape == (n,) and times.shape == (T,)
        sdt = rates.dtype
        assert sdt in (torch.float64, torch.float32) and commands.dtype == sdt
        fn = lib.fdyn_rate_metrics_f64 if sdt == torch.float64 else lib.fdyn_rate_metrics_f32
        times = times.to(device=dev, dtype=torch.float64).contiguous()
        rates, commands = rates.to(dev).contiguous(), commands.to(dev).contiguous()
        actions = actions.to(device=dev, dtype=torch.float32).contiguous()
        rewards = rewards.to(device=dev, dtype=sdt).contiguous()
        lengths = lengths.to(device=dev, dtype=torch.int32).contiguous()
        out = torch.empty((L.FD_NM, n), dtype=torch.float64, device=dev)
        rc = fn(_lib.ptr(times), _lib.ptr(rates), _lib.ptr(commands), _lib.ptr(actions), _lib.ptr(rewards),
                _lib.ptr(lengths), float(self.settling_threshold), self.settle_steps, T, n, _lib.ptr(out),
                _lib.current_stream())
        _lib.check(rc, "MetricsCalculator.compute_batch")
        return out

    def compute_metrics(self, times: np.ndarray, rates: np.ndarray, commands: np.ndarray, actions: np.ndarray,
                        rewards: np.ndarray) -> RateControlMetrics:
        """metrics.py:118-180: one episode, NumPy in ([len], [len,3], [len,3], [len,4], [len])."""
        dev = _lib.require_gpu()
        n = len(times)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        out = self.compute_batch(t(times, torch.float64), t(rates, torch.float64).reshape(n, 3, 1),
                                 t(commands, torch.float64).reshape(n, 3, 1), t(actions, torch.float32).reshape(n, 1, 4),
                                 t(rewards, torch.float64).reshape(n, 1),
                                 torch.full((1,), n, dtype=torch.int32, device=dev))
        return RateControlMetrics.from_vector(out[:, 0].cpu().numpy())


def aggregate_metrics(metrics: Union[torch.Tensor, List[RateControlMetrics]]) -> RateControlMetrics:
    """eval_rate.py:238-263: mean of every field over the episodes; `success` becomes the success rate."""
    if isinstance(metrics, torch.Tensor):
        return RateControlMetrics.from_vector(metrics.mean(dim=1).cpu().numpy(), success_as_rate=True)
    n = len(metrics)
    avg = RateControlMetrics()
    for f in fields(avg):
        if f.name == "success":
            avg.success = sum(float(m.success) for m in metrics) / n
        else:
            setattr(avg, f.name, float(np.mean([getattr(m, f.name) for m in metrics])))
    return avg


def compare_metrics(metrics_a: RateControlMetrics, metrics_b: RateControlMetrics, name_a: str = "Controller A",
                    name_b: str = "Controller B"):
    """metrics.py:365-424: side-by-side print with percentage differences."""
    def pct(a, b):
        return 0.0 if b == 0 else ((a - b) / b) * 100.0

    bar = "=" * 60
    print(f"\n{bar}\nPerformance Comparison: {name_a} vs {name_b}\n{bar}")
    print("\nSettling Time (s):")
    for ax in ("roll", "pitch", "yaw"):
        a, b = getattr(metrics_a, f"settling_time_{ax}"), getattr(metrics_b, f"settling_time_{ax}")
        print(f"  {ax.capitalize() + ':':6s} {a:.3f} vs {b:.3f} ({pct(a, b):+.1f}%)")
    print("\nOvershoot (%):")
    for ax in ("roll", "pitch", "yaw"):
        print(f"  {ax.capitalize() + ':':6s} {getattr(metrics_a, f'overshoot_{ax}'):.1f} vs "
              f"{getattr(metrics_b, f'overshoot_{ax}'):.1f}")
    print("\nTracking RMSE (rad/s):")
    print(f"  {metrics_a.tracking_rmse:.4f} vs {metrics_b.tracking_rmse:.4f} "
          f"({pct(metrics_a.tracking_rmse, metrics_b.tracking_rmse):+.1f}%)")
    print("\nControl Smoothness:")
    print(f"  {metrics_a.control_smoothness:.4f} vs {metrics_b.control_smoothness:.4f} "
          f"({pct(metrics_a.control_smoothness, metrics_b.control_smoothness):+.1f}%)")
    print("\nSuccess:")
    print(f"  {metrics_a.success} vs {metrics_b.success}")
    print("\nTotal Reward:")
    print(f"  {metrics_a.total_reward:.2f} vs {metrics_b.total_reward:.2f}\n{bar}\n")


# ---- evaluation loops (eval_rate.py:25-235), all episodes in parallel -------------------------------------------------
class EpisodeRecorder:
    """Device-side [T][word][N] trajectory block of the FIRST episode of each env."""

    def __init__(self, env, T: int):
        n, dev = env.n, env.device
        self.env, self.T = env, T
        self.rates = torch.zeros((T, 3, n), dtype=env.dtype, device=dev)
        self.commands = torch.zeros((T, 3, n), dtype=env.dtype, device=dev)
        self.actions = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
        self.rewards = torch.zeros((T, n), dtype=env.dtype, device=dev)
        self.lengths = torch.zeros(n, dtype=torch.int32, device=dev)
        self.alive = torch.ones(n, dtype=torch.bool, device=dev)
        # RateControlEnv accumulates current_time += dt (rate_env.py:241); np.cumsum adds in the same order
        self.times = torch.as_tensor(np.cumsum(np.full(T, env.dt, dtype=np.float64)), device=dev)

    def record_state(self, t: int):
        env = self.env
        self.rates[t].copy_(env.x[L.FD_X_P:L.FD_X_R + 1])
        self.commands[t].copy_(env.e[L.FD_E_CMD_P:L.FD_E_CMD_R + 1])

    def record_step(self, t: int, actions: torch.Tensor):
        env = self.env
        self.actions[t].copy_(actions)
        self.rewards[t].copy_(env.rewards_full)
        done = (env.terminated | env.truncated).bool()
        ended = self.alive & done
        self.lengths.masked_fill_(ended, t + 1)
        self.alive &= ~done

    def finish(self):
        self.lengths.masked_fill_(self.alive, self.T)

    def metrics(self, calculator: Optional[MetricsCalculator] = None) -> torch.Tensor:
        calc = calculator or MetricsCalculator(dt=self.env.dt)
        return calc.compute_batch(self.times, self.rates, self.commands, self.actions, self.rewards, self.lengths)


def _make_env(n_episodes, difficulty, episode_length, command_type, seed, precision, sampling, dt):
    from .rate_env import GpuRateVecEnv
    return GpuRateVecEnv(n_episodes, difficulty, episode_length, dt, command_type, seed=seed, precision=precision,
                         sampling=sampling)


@torch.no_grad()
def evaluate_pid_controller(n_episodes: int = 10, difficulty: str = "medium", episode_length: float = 10.0,
                            command_type: str = "step", seed: Optional[int] = None, precision: str = "mixed",
                            sampling: str = "device", dt: float = 0.02, throttle: float = 0.5,
                            pid_dt: Optional[float] = None, return_recorder: bool = False):
    """eval_rate.py:129-235: the rate PID (default gains) flies every episode; rates and command are sampled BEFORE each
    step, time after it (:190-222); throttle 0.5 (:196); `compute_action` is called without dt (:200), so the PIDs see
    ControllerConfig.rate_loop_dt (rate_agent.py:103) -- pass pid_dt=dt for a PID told the true step.
    Returns (metrics [FD_NM, N] float64 on the device, aggregated RateControlMetrics)."""
    env = _make_env(n_episodes, difficulty, episode_length, command_type, seed, precision, sampling, dt)
    env.casc_consts = torch.as_tensor(
        cascade_consts(pid_throttle=throttle, pid_dt=ControllerConfig().rate_loop_dt if pid_dt is None else pid_dt),
        device=env.device)
    T = int(episode_length / dt)
    rec = EpisodeRecorder(env, T)
    env.reset()
    for t in range(T):
        rec.record_state(t)
        env.step(None, auto_reset=False)
        rec.record_step(t, env.actions_taken)
    rec.finish()
    m = rec.metrics()
    return (m, aggregate_metrics(m), rec) if return_recorder else (m, aggregate_metrics(m))


@torch.no_grad()
def evaluate_learned_controller(policy, n_episodes: int = 10, difficulty: str = "medium", episode_length: float = 10.0,
                                command_type: str = "step", deterministic: bool = True, seed: Optional[int] = None,
                                precision: str = "mixed", sampling: str = "device", dt: float = 0.02,
                                return_recorder: bool = False):
    """eval_rate.py:25-126: the policy flies every episode (recurrent state zeroed at the start, :84-86); rates and
    command are sampled AFTER each step (:101-106); the recorded action is the policy's output clipped to the action
    space, as SB3's `predict` returns it.  `policy`: a `RateLSTMPolicy` (already on the device)."""
    env = _make_env(n_episodes, difficulty, episode_length, command_type, seed, precision, sampling, dt)
    T = int(episode_length / dt)
    rec = EpisodeRecorder(env, T)
    obs = env.reset()
    states = policy.initial_state(env.n, env.device)
    start = torch.ones(env.n, device=env.device)
    lo = torch.tensor([-1.0, -1.0, -1.0, 0.0], device=env.device)
    hi = torch.ones(4, device=env.device)
    for t in range(T):
        act, _, _, states = policy.step(obs, states, start, deterministic=deterministic)
        start = torch.zeros_like(start)
        act = torch.minimum(torch.maximum(act.float(), lo), hi)
        obs, _, _, _ = env.step(act, auto_reset=False)
        rec.record_state(t)
        rec.record_step(t, act)
    rec.finish()
    m = rec.metrics()
    return (m, aggregate_metrics(m), rec) if return_recorder else (m, aggregate_metrics(m))
