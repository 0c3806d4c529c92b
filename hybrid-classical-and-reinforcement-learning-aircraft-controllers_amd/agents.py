"""The four PID agents of the cascade commanded directly: host mirror of controllers/rate_agent.py:19-141,
attitude_agent.py:19-170, hsa_agent.py:22-250, waypoint_agent.py:20-275 over `fdyn_agent_step_*`.

`RateAgent / AttitudeAgent / HSAAgent / WaypointAgent` keep the reference's single-aircraft surface --
`compute_action(command, state, dt) -> ControlSurfaces`, `reset()`, `get_control_level()` (+ `reached_waypoint`) -- each call
one launch with N = 1 and no physics (a drop-in, not the fast path).  `AgentFleet` is the batched form: N aircraft, a
command row per aircraft, `compute_action` (surfaces only) or `run` (n_steps of agent -> set_controls -> RK4 in one launch),
which is also how the reference's closed-loop tests drive these agents (tests/test_control_integration.py:34-74).
"""
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
