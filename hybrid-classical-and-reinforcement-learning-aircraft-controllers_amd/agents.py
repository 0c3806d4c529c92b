"""The four PID agents of the cascade commanded directly: host mirror of controllers/rate_agent.py:19-141,
attitude_agent.py:19-170, hsa_agent.py:22-250, waypoint_agent.py:20-275 over `fdyn_agent_step_*`.

`RateAgent / AttitudeAgent / HSAAgent / WaypointAgent` keep the reference's single-aircraft surface --
`compute_action(command, state, dt) -> ControlSurfaces`, `reset()`, `get_control_level()` (+ `reached_waypoint`) -- each call
one launch with N = 1 and no physics (a drop-in, not the fast path).  `AgentFleet` is the batched form: N aircraft, a
command row per aircraft, `compute_action` (surfaces only) or `run` (n_steps of agent -> set_controls -> RK4 in one launch),
which is also how the reference's closed-loop tests drive these agents (tests/test_control_integration.py:34-74).
"""
from abc import ABC, abstractmethod
from typing import Optional

import numpy as np
import torch

from . import _lib, layout as L
from .config import FlightControlConfig, GuidanceConfig, cascade_consts, pid_table
from .fleet import BatchedSixDOF
from .flight_types import AircraftState, ControlCommand, ControlMode, ControllerConfig, ControlSurfaces, Waypoint

LEVELS = {ControlMode.RATE: L.FD_LEVEL_RATE, ControlMode.ATTITUDE: L.FD_LEVEL_ATTITUDE, ControlMode.HSA: L.FD_LEVEL_HSA,
          ControlMode.WAYPOINT: L.FD_LEVEL_WAYPOINT}


def validate_command(command: ControlCommand, mode_name: str, fields):
    """controllers/utils/validation.py: the named command fields must be set."""
    missing = [f for f in fields if getattr(command, f) is None]
    if missing:
        raise ValueError(f"{mode_name} command missing required fields: {missing}")


def command_row(command: ControlCommand) -> np.ndarray:
    """[4] float64 command row of `fdyn_agent_step_*` for a ControlCommand (NaN = 'not given' where the agent allows it)."""
    nan = float("nan")
    if command.mode == ControlMode.RATE:
        validate_command(command, "RATE", ["roll_rate", "pitch_rate", "yaw_rate"])
        return np.array([command.roll_rate, command.pitch_rate, command.yaw_rate,
                         0.0 if command.throttle is None else command.throttle])
    if command.mode == ControlMode.ATTITUDE:
        validate_command(command, "ATTITUDE", ["roll_angle", "pitch_angle"])
        return np.array([command.roll_angle, command.pitch_angle, nan if command.yaw_angle is None else command.yaw_angle,
                         0.0 if command.throttle is None else command.throttle])
    if command.mode == ControlMode.HSA:
        validate_command(command, "HSA", ["heading", "altitude", "speed"])
        return np.array([command.heading, command.speed, command.altitude, 0.0])
    if command.mode == ControlMode.WAYPOINT:
        validate_command(command, "WAYPOINT", ["waypoint"])
        w = command.waypoint
        return np.array([w.north, w.east, w.altitude, nan if w.speed is None else w.speed])
    raise ValueError(f"no PID agent for mode {command.mode}")


# ---- observation / action spaces per control level: interfaces/agent.py:154-323 --------------------------------------
_INF = float("inf")
_OBS_SPACES = {          # :166-198
    ControlMode.WAYPOINT: ((12,), "[pos(3), vel(3), att(3), wp_error(3)]"),
    ControlMode.HSA: ((12,), "[pos(3), vel(3), att(3), target_HSA(3)]"),
    ControlMode.RATE: ((10,), "[vel(3), att(3), rates(3), airspeed]"),
    ControlMode.ATTITUDE: ((10,), "[vel(3), att(3), rates(3), airspeed]"),
    ControlMode.SURFACE: ((14,), "[vel(3), att(3), rates(3), airspeed, aoa, sideslip, load]"),
}
_ACT_SPACES = {          # :214-246
    ControlMode.WAYPOINT: ([-_INF, -_INF, -_INF, 0.0], [_INF, _INF, _INF, 100.0], "[N, E, D, speed]"),
    ControlMode.HSA: ([0.0, 0.0, 0.0], [2 * np.pi, 100.0, 1000.0], "[heading(rad), speed(m/s), altitude(m)]"),
    ControlMode.RATE: ([-1.0, -1.0, -1.0, 0.0], [1.0, 1.0, 1.0, 1.0], "[roll_rate, pitch_rate, yaw_rate, throttle]"),
    ControlMode.ATTITUDE: ([-1.0, -1.0, -1.0, 0.0], [1.0, 1.0, 1.0, 1.0], "[roll_angle, pitch_angle, yaw_angle, throttle]"),
    ControlMode.SURFACE: ([-1.0, -1.0, -1.0, 0.0], [1.0, 1.0, 1.0, 1.0], "[elevator, aileron, rudder, throttle]"),
}


def observation_space(mode: ControlMode) -> dict:
    """interfaces/agent.py:154-200: {'shape', 'low', 'high', 'description'} of an agent commanded at `mode`."""
    if mode not in _OBS_SPACES:
        return {"shape": (0,), "low": 0, "high": 0}
    shape, desc = _OBS_SPACES[mode]
    return {"shape": shape, "low": -np.inf, "high": np.inf, "description": desc}


def action_space(mode: ControlMode) -> dict:
    """interfaces/agent.py:202-248."""
    if mode not in _ACT_SPACES:
        return {"shape": (0,), "low": 0, "high": 0}
    lo, hi, desc = _ACT_SPACES[mode]
    return {"shape": (len(lo),), "low": np.array(lo), "high": np.array(hi), "description": desc}


def preprocess_observations(mode: ControlMode, x: torch.Tensor, airspeed: torch.Tensor, altitude: torch.Tensor) -> torch.Tensor:
    """interfaces/agent.py:250-323 for a fleet: state block x [12][N] (+ airspeed / altitude [N]) -> observations [N, dim]
    on the device of `x`.  The target-dependent columns of levels 1-2 are zeros and the aerodynamic-angle columns of level 5
    are the reference's placeholders (0, 0, load factor 1): that is what the reference returns."""
    n = x.shape[1]
    if mode in (ControlMode.WAYPOINT, ControlMode.HSA):
        return torch.cat([x[0:9].T, torch.zeros((n, 3), dtype=x.dtype, device=x.device)], 1)
    if mode in (ControlMode.RATE, ControlMode.ATTITUDE):
        return torch.cat([x[3:12].T, airspeed.to(x.dtype).reshape(n, 1)], 1)
    if mode == ControlMode.SURFACE:
        col = lambda v: torch.full((n, 1), v, dtype=x.dtype, device=x.device)      # noqa: E731
        return torch.cat([x[3:12].T, airspeed.to(x.dtype).reshape(n, 1), col(0.0), col(0.0), col(1.0),
                          altitude.to(x.dtype).reshape(n, 1)], 1)
    raise ValueError(f"Unknown control level: {mode}")


class RLAgentInterface(ABC):
    """interfaces/agent.py:21-327: the contract of a learning agent at any control level (observation -> ControlCommand).
    Abstract: `get_control_level`, `reset(initial_state)`, `get_action(observation)`; `update` / `save` / `load` default to
    no-ops, `switch_control_level` raises, and the space / observation helpers follow the level."""

    @abstractmethod
    def get_control_level(self) -> ControlMode: ...

    @abstractmethod
    def reset(self, initial_state: AircraftState) -> None: ...

    @abstractmethod
    def get_action(self, observation: np.ndarray) -> ControlCommand: ...

    def update(self, transition: dict) -> None:
        pass

    def save(self, path: str) -> None:
        pass

    def load(self, path: str) -> None:
        pass

    def switch_control_level(self, level: ControlMode) -> None:
        raise NotImplementedError(f"{self.__class__.__name__} does not support level switching")       # :150-152

    def get_observation_space(self) -> dict:
        return observation_space(self.get_control_level())

    def get_action_space(self) -> dict:
        return action_space(self.get_control_level())

    def preprocess_observation(self, state: AircraftState) -> np.ndarray:
        x = torch.as_tensor(state.to_vector(), dtype=torch.float64).reshape(L.FD_NX, 1)
        one = lambda v: torch.tensor([float(v)], dtype=torch.float64)                                   # noqa: E731
        return preprocess_observations(self.get_control_level(), x, one(state.airspeed), one(state.altitude))[0].numpy()

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(level={self.get_control_level().name})"


class SurfaceAgent:
    """controllers/surface_agent.py:8-103, Level 5: surface commands pass straight through, saturated at the configured
    limits.  No device work: the physics kernels apply the same clip when the surfaces are set (`Controls::set`)."""
    _DEFAULTS = (("elevator", -1.0, 1.0), ("aileron", -1.0, 1.0), ("rudder", -1.0, 1.0), ("throttle", 0.0, 1.0))

    def __init__(self, config: Optional[dict] = None):
        self.config = config or {}
        limits = self.config.get("surface_limits", {})
        for name, lo, hi in self._DEFAULTS:
            setattr(self, f"{name}_min", limits.get(f"{name}_min", lo))
            setattr(self, f"{name}_max", limits.get(f"{name}_max", hi))

    def get_control_level(self) -> ControlMode:
        return ControlMode.SURFACE

    def compute_action(self, command: ControlCommand, state: Optional[AircraftState] = None, dt: Optional[float] = None) -> ControlSurfaces:
        assert command.mode == ControlMode.SURFACE, f"Surface agent expects SURFACE mode, got {command.mode}"
        return ControlSurfaces(**{name: np.clip(getattr(command, name), getattr(self, f"{name}_min"), getattr(self, f"{name}_max"))
                                  for name, _, _ in self._DEFAULTS})

    def limits(self, device=None) -> torch.Tensor:
        """[2][4] (min row, max row) in physics control order [elevator, aileron, rudder, throttle], for fleets:
        `u.clamp(lim[0], lim[1])` on a `[N][4]` surface block is this agent for N aircraft."""
        order = ("elevator", "aileron", "rudder", "throttle")
        return torch.tensor([[getattr(self, f"{n}_min") for n in order], [getattr(self, f"{n}_max") for n in order]],
                            dtype=torch.float64, device=device)

    def reset(self):
        pass

    def __repr__(self) -> str:
        return (f"SurfaceAgent(elevator=[{self.elevator_min}, {self.elevator_max}], aileron=[{self.aileron_min}, {self.aileron_max}], "
                f"rudder=[{self.rudder_min}, {self.rudder_max}], throttle=[{self.throttle_min}, {self.throttle_max}])")


class AgentFleet(BatchedSixDOF):
    """N aircraft, each with its own PID states for all nine loops, commanded at any level."""

    def __init__(self, n: int, precision: str = "f64", config: Optional[ControllerConfig] = None,
                 flight_config: Optional[FlightControlConfig] = None, guidance_type: str = "LOS", **kw):
        super().__init__(n, precision, **kw)
        self.pid_cfg = torch.as_tensor(pid_table(config, flight_config), device=self.device)
        self.cfg_per_lane = 0
        self.consts = torch.as_tensor(cascade_consts(config, flight_config, guidance_type), device=self.device)
        self.pid_state = torch.zeros((L.FD_NPID * L.FD_NPS, self.n), dtype=torch.float32, device=self.device)
        self.surfaces = torch.zeros((L.FD_NU, self.n), dtype=self.dtype, device=self.device)
        self._agent_fn = getattr(self.lib, f"fdyn_agent_step_{precision}")

    def reset_agents(self):
        self.pid_state.zero_()

    def set_gain_tables(self, tables):
        """One PID table per aircraft ([N][FD_NPID][FD_NPC] float32, rows as config.pid_table): a gain sweep in one launch."""
        t = torch.as_tensor(np.asarray(tables, np.float32), device=self.device).contiguous()
        assert t.shape == (self.n, L.FD_NPID, L.FD_NPC)
        self.pid_cfg, self.cfg_per_lane = t, 1

    def _cmd(self, cmd):
        c = torch.as_tensor(cmd, dtype=self.dtype, device=self.device)
        if c.ndim == 1:
            c = c[:, None].expand(4, self.n)
        assert c.shape == (4, self.n)
        return c.contiguous()

    def _launch(self, level, cmd, dt, n_steps):
        c = self._cmd(cmd)
        rc = self._agent_fn(int(level), _lib.ptr(self.x), _lib.ptr(self.pid_state), _lib.ptr(self.type_index), _lib.ptr(self.params),
                            self.n_types, _lib.ptr(self.pid_cfg), self.cfg_per_lane, _lib.ptr(self.consts), _lib.ptr(c), self.n, float(dt),
                            int(n_steps), _lib.ptr(self.surfaces), _lib.current_stream())
        _lib.check(rc, "agent step")

    def compute_action(self, level: int, cmd, dt: float = 0.01) -> torch.Tensor:
        """Surfaces [4][N] (FD_U_* rows: elevator, aileron, rudder, throttle) for the current states; PID states advance."""
        self._launch(level, cmd, dt, 0)
        return self.surfaces

    def run(self, level: int, cmd, dt: float, n_steps: int):
        """n_steps x {compute_action -> set_controls -> one RK4 of dt}, state in registers, one launch."""
        self._launch(level, cmd, dt, n_steps)
        self.time += dt * n_steps


class _SingleAgent:
    LEVEL = None
    MODE = None

    def __init__(self, config: ControllerConfig, flight_config: Optional[FlightControlConfig] = None, guidance_type: str = "LOS",
                 precision: str = "f64"):
        self.config = config
        self._fleet = AgentFleet(1, precision, config, flight_config, guidance_type)

    def get_control_level(self) -> ControlMode:
        return self.MODE

    def _default_dt(self) -> float:
        return 0.01

    def compute_action(self, command: ControlCommand, state: AircraftState, dt: Optional[float] = None) -> ControlSurfaces:
        assert command.mode == self.MODE, f"{type(self).__name__} expects {self.MODE.name} mode, got {command.mode}"
        row = command_row(command)
        f = self._fleet
        f.x.copy_(torch.as_tensor(state.to_vector(), device=f.device).to(f.dtype).reshape(L.FD_NX, 1))
        s = f.compute_action(self.LEVEL, row, self._default_dt() if dt is None else dt)[:, 0].to(torch.float64).cpu().numpy()
        return ControlSurfaces(elevator=float(s[L.FD_U_ELEVATOR]), aileron=float(s[L.FD_U_AILERON]),
                               rudder=float(s[L.FD_U_RUDDER]), throttle=float(s[L.FD_U_THROTTLE]))

    def reset(self):
        self._fleet.reset_agents()

    def __repr__(self) -> str:
        return f"{type(self).__name__}(level={self.MODE.name})"


class RateAgent(_SingleAgent):
    """controllers/rate_agent.py:19-141.  dt=None -> ControllerConfig.rate_loop_dt (:103)."""
    LEVEL, MODE = L.FD_LEVEL_RATE, ControlMode.RATE

    def __init__(self, config: ControllerConfig, precision: str = "f64"):
        super().__init__(config, precision=precision)

    def _default_dt(self) -> float:
        return self.config.rate_loop_dt


class AttitudeAgent(_SingleAgent):
    """controllers/attitude_agent.py:19-170 (outer angle loop -> inner RateAgent, same dt)."""
    LEVEL, MODE = L.FD_LEVEL_ATTITUDE, ControlMode.ATTITUDE

    def __init__(self, config: ControllerConfig, precision: str = "f64"):
        super().__init__(config, precision=precision)


class HSAAgent(_SingleAgent):
    """controllers/hsa_agent.py:22-250 (heading -> bank, TECS energy / balance -> throttle / pitch -> AttitudeAgent)."""
    LEVEL, MODE = L.FD_LEVEL_HSA, ControlMode.HSA

    def __init__(self, config: ControllerConfig, flight_config: Optional[FlightControlConfig] = None, precision: str = "f64"):
        super().__init__(config, flight_config, precision=precision)


class WaypointAgent(_SingleAgent):
    """controllers/waypoint_agent.py:20-275 (LOS / pure-pursuit / default guidance -> HSAAgent)."""
    LEVEL, MODE = L.FD_LEVEL_WAYPOINT, ControlMode.WAYPOINT

    def __init__(self, config: ControllerConfig, guidance_type: str = "LOS", flight_config: Optional[FlightControlConfig] = None,
                 precision: str = "f64"):
        super().__init__(config, flight_config, guidance_type, precision)
        self.guidance_type = guidance_type
        g = flight_config.guidance if flight_config is not None else GuidanceConfig()
        self.acceptance_radius = g.acceptance_radius

    def reached_waypoint(self, state: AircraftState, waypoint: Waypoint) -> bool:
        """waypoint_agent.py:255-267: 3-D distance to the waypoint below the acceptance radius."""
        err = np.array([waypoint.north - state.north, waypoint.east - state.east, waypoint.down - state.down])
        return bool(np.linalg.norm(err) < self.acceptance_radius)
