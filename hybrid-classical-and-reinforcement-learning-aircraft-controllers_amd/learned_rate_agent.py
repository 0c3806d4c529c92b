"""Level 4 learned rate controller: host mirror of controllers/learned_rate_agent.py:26-284 over the device policy.

`LearnedRateAgent(model_path, config)` keeps the reference's surface -- `compute_action(command, state, dt=None) ->
ControlSurfaces`, `reset()`, `get_control_level()`, `using_fallback`, PID fallback -- so a policy trained here drops into
the reference's cascade / GUI worker (gui/simulation_worker_learned.py:51-93) where its SB3 agent sits.  `model_path` is a
checkpoint written by `RecurrentPPO.save` (train_rate.py) or an archive in the Stable-Baselines3 `.zip` layout (the
reference's own format, :87-118): its `policy.pth` is read with `weights_only=True` and mapped by name (sb3_zip.py);
the pickled parts of such an archive are never decoded.  stable-baselines3 is absent here, so that file format is
written and read from its published layout only (parity unpinned).
`BatchedLearnedRateAgent` is the same mapping for a fleet: rate commands [N,3] + state block [12][N] -> actions [N,4].
"""
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from . import layout as L
from .flight_types import AircraftState, ControlCommand, ControlMode, ControllerConfig, ControlSurfaces

_ACT_LOW = (-1.0, -1.0, -1.0, 0.0)


def _clip_action(a: torch.Tensor) -> torch.Tensor:
    lo = torch.tensor(_ACT_LOW, device=a.device, dtype=a.dtype)
    return torch.minimum(torch.maximum(a, lo), torch.ones(4, device=a.device, dtype=a.dtype))


class BatchedLearnedRateAgent:
    """Observation assembly (learned_rate_agent.py:158-178 = rate_env.py:374-408) + deterministic policy step for N
    aircraft on the device.  `x` [12][N] state block, `airspeed`/`altitude` [N] (or None: from x), `rate_cmd` [N,3]."""

    def __init__(self, policy, n: int, config: Optional[ControllerConfig] = None, device=None):
        self.policy, self.n = policy, int(n)
        self.device = device or next(policy.parameters()).device
        c = config or ControllerConfig()
        self.max_rates = torch.tensor(np.radians([c.max_roll_rate, c.max_pitch_rate, c.max_yaw_rate]), dtype=torch.float64,
                                      device=self.device)
        self.obs = torch.zeros((self.n, L.FD_OBS_DIM), dtype=torch.float32, device=self.device)
        self.reset()

    def reset(self, mask: Optional[torch.Tensor] = None):
        if mask is None:
            self.states = self.policy.initial_state(self.n, self.device)
            self.prev_action = torch.tensor([0.0, 0.0, 0.0, 0.5], device=self.device).repeat(self.n, 1)
            self.start = torch.ones(self.n, device=self.device)
        else:                                   # per-aircraft reset: the policy zeroes the state where start == 1
            m = mask.bool()
            self.prev_action[m] = torch.tensor([0.0, 0.0, 0.0, 0.5], device=self.device)
            self.start = torch.maximum(self.start, m.float())

    @torch.no_grad()
    def compute_actions(self, rate_cmd: torch.Tensor, x: torch.Tensor, airspeed: Optional[torch.Tensor] = None,
                        altitude: Optional[torch.Tensor] = None) -> torch.Tensor:
        o = self.obs
        mr = self.max_rates.to(x.dtype)                      # clip and difference in the state's precision, then narrow
        cmd = torch.minimum(torch.maximum(rate_cmd.to(x.dtype), -mr), mr)                                 # :152-155
        rates = x[L.FD_X_P:L.FD_X_R + 1].T
        if airspeed is None:
            airspeed = x[L.FD_X_U:L.FD_X_W + 1].to(torch.float64).square().sum(0).sqrt()
        if altitude is None:
            altitude = -x[L.FD_X_D]
        o[:, 0:3] = rates
        o[:, 3:6] = cmd
        o[:, 6:9] = cmd - rates
        o[:, 9], o[:, 10] = airspeed, altitude
        o[:, 11:14] = x[L.FD_X_ROLL:L.FD_X_YAW + 1].T
        o[:, 14:18] = self.prev_action
        act, _, _, self.states = self.policy.step(o, self.states, self.start, deterministic=True)
        self.start = torch.zeros_like(self.start)
        act = _clip_action(act.float())                  # SB3 `predict` clips to the action space
        self.prev_action = act
        return act


class LearnedRateAgent:
    """controllers/learned_rate_agent.py:26-284."""

    def __init__(self, model_path: Optional[str], config: ControllerConfig, fallback_to_pid: bool = True,
                 device: str = "auto", policy=None):
        self.config, self.fallback_to_pid, self.model_path = config, fallback_to_pid, model_path
        if policy is None:
            if not Path(model_path).exists():
                raise FileNotFoundError(f"Model not found: {model_path}")          # :98-99
            from .eval_rate import load_policy
            policy = load_policy(model_path, device="cuda" if device == "auto" else device)
        self.model = policy
        self.is_recurrent = bool(getattr(policy, "use_lstm", True))
        self._batched = BatchedLearnedRateAgent(policy, 1, config)
        self.prev_action = np.array([0.0, 0.0, 0.0, 0.5])
        self.obs = np.zeros(18, dtype=np.float32)
        self.max_roll_rate, self.max_pitch_rate = np.radians(config.max_roll_rate), np.radians(config.max_pitch_rate)
        self.max_yaw_rate = np.radians(config.max_yaw_rate)
        self._pid_fallback = None
        self.using_fallback = False

    def get_control_level(self) -> ControlMode:
        return ControlMode.RATE

    def compute_action(self, command: ControlCommand, state: AircraftState, dt: float = None) -> ControlSurfaces:
        assert command.mode == ControlMode.RATE, f"Learned rate agent expects RATE mode, got {command.mode}"
        dev = self._batched.device
        try:
            cmd = torch.tensor([[command.roll_rate, command.pitch_rate, command.yaw_rate]], dtype=torch.float64, device=dev)
            x = torch.as_tensor(state.to_vector(), device=dev).reshape(L.FD_NX, 1)
            action = self._predict(cmd, x, state)
            self.using_fallback = False
        except Exception as e:                                                      # :182-190
            if not self.fallback_to_pid:
                raise
            print(f"Model prediction failed, using PID fallback: {e}")
            action = self._pid_fallback_action(command, state, dt)
            self.using_fallback = True
        self.prev_action = action.copy()
        self.obs = self._batched.obs[0].cpu().numpy()
        return ControlSurfaces(aileron=float(np.clip(action[0], -1.0, 1.0)), elevator=float(np.clip(action[1], -1.0, 1.0)),
                               rudder=float(np.clip(action[2], -1.0, 1.0)), throttle=float(np.clip(action[3], 0.0, 1.0)))

    def _predict(self, cmd, x, state) -> np.ndarray:
        a = self._batched.compute_actions(cmd, x, torch.tensor([state.airspeed], device=x.device),
                                          torch.tensor([state.altitude], device=x.device))
        return a[0].cpu().numpy().astype(np.float64)

    def _pid_fallback_action(self, command, state, dt=None) -> np.ndarray:
        """controllers/rate_agent.py:65-124 on the HIP PID (lazy, :243-246)."""
        from .config import pid_table
        from . import _lib
        if self._pid_fallback is None:
            dev = self._batched.device
            self._pid_fallback = {"cfg": torch.as_tensor(pid_table(self.config)[:3].copy(), device=dev),
                                  "state": torch.zeros((3, L.FD_NPS, 1), dtype=torch.float32, device=dev)}
        fb, lib = self._pid_fallback, _lib.load()
        dev = fb["cfg"].device
        cmd = np.clip([command.roll_rate, command.pitch_rate, command.yaw_rate],
                      [-self.max_roll_rate, -self.max_pitch_rate, -self.max_yaw_rate],
                      [self.max_roll_rate, self.max_pitch_rate, self.max_yaw_rate])
        out = torch.zeros(1, dtype=torch.float32, device=dev)
        res = []
        for k, meas in enumerate((state.p, state.q, state.r)):
            sp = torch.tensor([cmd[k]], dtype=torch.float32, device=dev)
            ms = torch.tensor([meas], dtype=torch.float32, device=dev)
            rc = lib.fdyn_pid_compute_batch(fb["cfg"][k].data_ptr(), 0, fb["state"][k].data_ptr(), sp.data_ptr(), ms.data_ptr(),
                                            float(dt if dt is not None else self.config.rate_loop_dt), out.data_ptr(), 1,
                                            _lib.current_stream())
            _lib.check(rc, "rate PID")
            res.append(float(out.item()))
        thr = command.throttle if command.throttle is not None else 0.0
        return np.array([np.clip(res[0], -1, 1), np.clip(-res[1], -1, 1), np.clip(-res[2], -1, 1), np.clip(thr, 0, 1)])

    def reset(self):
        self._batched.reset()
        self.prev_action[:] = [0.0, 0.0, 0.0, 0.5]
        self.obs = np.zeros(18, dtype=np.float32)
        if self._pid_fallback is not None:
            self._pid_fallback["state"].zero_()
        self.using_fallback = False

    def __repr__(self) -> str:
        model_type = "RecurrentPPO" if self.is_recurrent else "PPO"
        return f"LearnedRateAgent(model={model_type}, path={self.model_path}, fallback={self.fallback_to_pid})"
