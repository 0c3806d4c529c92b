"""Device-resident fleets: batched `Simplified6DOF` and the batched 5-level cascade.

`BatchedSixDOF` is the N-aircraft counterpart of the reference's `Simplified6DOF` + `SimulationAircraftBackend`
(simulation/simplified_6dof.py:148-331, simulation/simulation_backend.py:13-135): same reset / set_controls / step
semantics, state held structure-of-arrays in HBM as `x[12][N]`, every step one launch of the HIP kernel.
`BatchedCascade` adds the mission planner + Waypoint/HSA/Attitude/Rate agents of controllers/*.py fused in front of
the integrator (examples/03_waypoint_square_demo.py:148-209 per aircraft).
"""
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib, layout as L
from .config import (FlightControlConfig, cascade_consts, pid_table, waypoint_table)
from .flight_types import ControllerConfig, Waypoint
from .params import AircraftParams, param_table


class BatchedSixDOF:
    def __init__(self, n: int, precision: str = "f64", types: Sequence = ("rc_plane",),
                 type_index: Optional[np.ndarray] = None, device=None):
        self.lib = _lib.load()
        self.device = device or _lib.require_gpu()
        self.n, self.precision = int(n), precision
        self.dtype = _lib.state_dtype(precision)
        self.params_host = param_table(types)
        self.n_types = len(self.params_host)
        self.params = torch.as_tensor(self.params_host, device=self.device).contiguous()
        self.type_index = None
        if type_index is not None:
            self.type_index = torch.as_tensor(np.asarray(type_index, np.uint8), device=self.device).contiguous()
        self.x = torch.zeros((L.FD_NX, self.n), dtype=self.dtype, device=self.device)
        self.u = torch.zeros((L.FD_NU, self.n), dtype=self.dtype, device=self.device)
        self.time = 0.0
        self._step_fn = getattr(self.lib, f"fdyn_sixdof_step_{precision}")
        self._derived_fn = getattr(self.lib, "fdyn_derived_f32" if precision == "f32" else "fdyn_derived_f64")
        self.reset()

    # Simplified6DOF.reset (simplified_6dof.py:189-212): default = level flight, 100 m, 20 m/s
    def reset(self, x0: Optional[np.ndarray] = None):
        if x0 is None:
            x0 = np.zeros((self.n, L.FD_NX))
            x0[:, L.FD_X_D] = -100.0
            x0[:, L.FD_X_U] = 20.0
        x0 = np.asarray(x0, dtype=np.float64).reshape(self.n, L.FD_NX)
        self.x.copy_(torch.as_tensor(np.ascontiguousarray(x0.T), device=self.device).to(self.dtype))
        self.time = 0.0

    # Simplified6DOF.set_controls (:214-226); rows = [elevator, aileron, rudder, throttle]; clip happens in-kernel
    def set_controls(self, u):
        if isinstance(u, torch.Tensor):
            self.u.copy_(u.to(self.dtype).reshape(L.FD_NU, self.n))
        else:
            u = np.asarray(u, dtype=np.float64).reshape(self.n, L.FD_NU)
            self.u.copy_(torch.as_tensor(np.ascontiguousarray(u.T), device=self.device).to(self.dtype))

    # SimulationAircraftBackend.step (simulation_backend.py:82-101) when dt_physics is given, else Simplified6DOF.step
    def step(self, dt: float, dt_physics: Optional[float] = None, derived_out: Optional[torch.Tensor] = None):
        n_sub = self.lib.fdyn_num_substeps(dt, dt_physics) if dt_physics else 1
        rc = self._step_fn(_lib.ptr(self.x), _lib.ptr(self.u), _lib.ptr(self.type_index), _lib.ptr(self.params),
                           self.n_types, self.n, float(dt), n_sub, _lib.ptr(derived_out), _lib.current_stream())
        _lib.check(rc, "Simplified6DOF.step")
        self.time += dt
        return n_sub

    def derived(self) -> torch.Tensor:
        """[4][N]: airspeed, altitude, ground_speed, heading (simplified_6dof.py:295-331)."""
        out = torch.empty((L.FD_ND, self.n), dtype=self.dtype, device=self.device)
        _lib.check(self._derived_fn(_lib.ptr(self.x), self.n, _lib.ptr(out), _lib.current_stream()), "get_state")
        return out

    def state_numpy(self) -> np.ndarray:
        """[N][12] float64 host copy."""
        return self.x.to(torch.float64).T.contiguous().cpu().numpy()


class BatchedCascade(BatchedSixDOF):
    """N aircraft each flying the same waypoint mission under the 5-level cascaded PID stack."""

    def __init__(self, n: int, waypoints: Sequence[Waypoint], precision: str = "f64",
                 config: Optional[ControllerConfig] = None, flight_config: Optional[FlightControlConfig] = None,
                 guidance_type: str = "PP", acceptance_radius: Optional[float] = None, on_complete: str = "freeze",
                 **kw):
        super().__init__(n, precision, **kw)
        self.n_wp = len(waypoints)
        self.wps = torch.as_tensor(waypoint_table(waypoints), device=self.device)
        self.pid_cfg = torch.as_tensor(pid_table(config, flight_config), device=self.device)
        self.consts = torch.as_tensor(cascade_consts(config, flight_config, guidance_type, acceptance_radius,
                                                     on_complete), device=self.device)
        self.pid_state = torch.zeros((L.FD_NPID * L.FD_NPS, self.n), dtype=torch.float32, device=self.device)
        self.wp_idx = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        self.reached_total = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        self.surfaces = torch.zeros((L.FD_NU, self.n), dtype=self.dtype, device=self.device)
        self._cascade_fn = getattr(self.lib, f"fdyn_cascade_step_{precision}")

    def reset(self, x0=None):
        super().reset(x0)
        if hasattr(self, "pid_state"):
            self.pid_state.zero_(); self.wp_idx.zero_(); self.reached_total.zero_()

    def run(self, dt: float, n_steps: int):
        """n_steps control steps in ONE launch (mission update -> agents -> RK4), all state in registers."""
        rc = self._cascade_fn(_lib.ptr(self.x), _lib.ptr(self.pid_state), _lib.ptr(self.wp_idx),
                              _lib.ptr(self.type_index), _lib.ptr(self.params), self.n_types, _lib.ptr(self.pid_cfg),
                              _lib.ptr(self.consts), _lib.ptr(self.wps), self.n_wp, self.n, float(dt), int(n_steps),
                              _lib.ptr(self.surfaces), _lib.ptr(self.reached_total), _lib.current_stream())
        _lib.check(rc, "cascade step")
        self.time += dt * n_steps

    def mission_complete(self) -> torch.Tensor:
        return self.wp_idx >= self.n_wp
