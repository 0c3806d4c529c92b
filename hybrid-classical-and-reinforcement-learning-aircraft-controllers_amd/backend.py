"""Single-aircraft views with the reference's class names and contracts, executed by the HIP path.

`AircraftInterface` (interfaces/aircraft.py:8-175), `Simplified6DOF` (simulation/simplified_6dof.py:148-331) and
`SimulationAircraftBackend` (simulation/simulation_backend.py:13-170) keep their method names, argument meaning, return
types and error behaviour; underneath each object is a 1-aircraft `BatchedSixDOF` in the fp64 variant, so existing
single-aircraft code (examples, tests, GUI loops) runs unchanged while fleets use `BatchedSixDOF` directly.
One step of one aircraft costs a launch plus a 128-byte read-back: use the batched classes for throughput.
"""
from abc import ABC, abstractmethod
from typing import Optional

import numpy as np
import torch

from . import layout as L
from .fleet import BatchedSixDOF
from .flight_types import AircraftState, ControlSurfaces
from .params import AircraftParams, aircraft_params_for


class AircraftInterface(ABC):
    @abstractmethod
    def step(self, dt: float) -> AircraftState: ...

    @abstractmethod
    def set_controls(self, surfaces: ControlSurfaces) -> None: ...

    @abstractmethod
    def reset(self, initial_state: Optional[AircraftState] = None) -> AircraftState: ...

    @abstractmethod
    def get_state(self) -> AircraftState: ...

    @abstractmethod
    def get_backend_type(self) -> str: ...

    def close(self) -> None:
        pass

    def get_dt_nominal(self) -> float:
        return 0.01

    def is_real_hardware(self) -> bool:
        return self.get_backend_type() in ["hardware", "hil"]

    def supports_reset(self) -> bool:
        return self.get_backend_type() == "simulation"

    def get_info(self) -> dict:
        return {"backend_type": self.get_backend_type(), "dt_nominal": self.get_dt_nominal()}

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(type={self.get_backend_type()})"


class Simplified6DOF:
    """One aircraft; `step(dt)` is ONE RK4 step like the reference class (no sub-stepping)."""

    def __init__(self, params: Optional[AircraftParams] = None, precision: str = "f64"):
        self.params = params or AircraftParams()
        self._fleet = BatchedSixDOF(1, precision, types=(self.params,))
        self._derived = torch.zeros((L.FD_ND, 1), dtype=self._fleet.dtype, device=self._fleet.device)
        self._controls = ControlSurfaces()
        self._time = 0.0
        self._refresh()

    def _refresh(self):
        d = self._fleet.derived()
        self._host = np.concatenate([self._fleet.x[:, 0].to(torch.float64).cpu().numpy(), d[:, 0].to(torch.float64).cpu().numpy()])

    @property
    def _state(self) -> np.ndarray:
        return self._host[:12]

    def reset(self, initial_state: Optional[AircraftState] = None) -> None:
        if initial_state is None:
            self._fleet.reset(None)
            self._time = 0.0
        else:
            self._fleet.reset(initial_state.to_vector()[None])
            self._time = initial_state.time
        self._refresh()

    def set_controls(self, controls: ControlSurfaces) -> None:
        self._controls = ControlSurfaces(elevator=float(np.clip(controls.elevator, -1.0, 1.0)),
                                         aileron=float(np.clip(controls.aileron, -1.0, 1.0)),
                                         rudder=float(np.clip(controls.rudder, -1.0, 1.0)),
                                         throttle=float(np.clip(controls.throttle, 0.0, 1.0)))
        self._fleet.set_controls(self._controls.to_array()[None])

    def step(self, dt: float, dt_physics: Optional[float] = None) -> AircraftState:
        self._fleet.step(dt, dt_physics, derived_out=self._derived)          # raises ValueError on a bad dt
        self._time += dt
        self._host = np.concatenate([self._fleet.x[:, 0].to(torch.float64).cpu().numpy(),
                                     self._derived[:, 0].to(torch.float64).cpu().numpy()])
        return self.get_state()

    def get_state(self) -> AircraftState:
        h = self._host
        return AircraftState.from_vector(h[:12].copy(), derived=h[12:16], time=self._time)


class SimulationAircraftBackend(AircraftInterface):
    def __init__(self, config: Optional[dict] = None):
        config = config or {}
        params = config["params"] if "params" in config else aircraft_params_for(config.get("aircraft_type", "rc_plane"))
        self._physics = Simplified6DOF(params, precision=config.get("precision", "f64"))
        self._dt_physics = config.get("dt_physics", 0.001)
        self._state = None

    def step(self, dt: float) -> AircraftState:
        self._state = self._physics.step(dt, self._dt_physics)              # n = max(1, int(dt/dt_physics)) sub-steps
        return self._state

    def set_controls(self, surfaces: ControlSurfaces) -> None:
        self._physics.set_controls(surfaces)

    def reset(self, initial_state: Optional[AircraftState] = None) -> AircraftState:
        self._physics.reset(initial_state)
        self._state = self._physics.get_state()
        return self._state

    def get_state(self) -> AircraftState:
        if self._state is None:
            self._state = self._physics.get_state()
        return self._state

    def get_backend_type(self) -> str:
        return "simulation"

    def get_dt_nominal(self) -> float:
        return 0.01

    def get_info(self) -> dict:
        info = super().get_info()
        info.update({"physics_engine": "simplified_6dof", "dt_physics": self._dt_physics,
                     "aircraft_mass": self._physics.params.mass, "max_thrust": self._physics.params.max_thrust})
        return info

    def __repr__(self) -> str:
        return f"SimulationAircraftBackend(physics=Simplified6DOF[HIP], dt={self._dt_physics})"
