"""Aircraft parameter blocks.

`AircraftParams` keeps the reference's field names, defaults and validation
(simulation/simplified_6dof.py:31-145); `to_block()` flattens it into the FD_P_* parameter block that the
kernels stage in LDS (one block per aircraft TYPE -- the reference has closed-form coefficients, no tables).
Degree limits are converted with np.radians on the host exactly where the reference does (:182-187).
"""
from dataclasses import dataclass

import numpy as np

from . import layout as L


@dataclass
class AircraftParams:
    mass: float = 8.0
    inertia_xx: float = 0.4
    inertia_yy: float = 0.6
    inertia_zz: float = 0.7
    wing_area: float = 0.5
    wing_span: float = 2.0
    chord: float = 0.25
    cl_0: float = 0.4
    cl_alpha: float = 5.0
    cd_0: float = 0.025
    cd_alpha2: float = 0.04
    cl_elevator: float = 0.6
    cm_elevator: float = -2.0
    cy_rudder: float = 0.5
    cn_rudder: float = -0.40
    cl_aileron: float = 0.50
    cm_alpha: float = -0.15
    cn_beta: float = 0.04
    cl_beta: float = -0.02
    damping_roll: float = -0.8
    damping_pitch: float = -2.0
    damping_yaw: float = -0.6
    max_thrust: float = 50.0
    air_density: float = 1.225
    gravity: float = 9.81
    min_airspeed_aero: float = 10.0
    min_u_velocity: float = 0.1
    max_elevator_deflection: float = 30.0
    max_aileron_deflection: float = 30.0
    max_rudder_deflection: float = 30.0
    thrust_zero_velocity: float = 50.0
    max_velocity: float = 100.0
    max_angular_rate: float = 360.0
    max_pitch_angle: float = 85.0
    max_alpha: float = 30.0
    max_acceleration: float = 50.0
    max_angular_acceleration: float = 1000.0
    max_timestep: float = 1.0
    min_timestep: float = 1e-6

    def __post_init__(self):
        if self.mass <= 0:
            raise ValueError(f"Mass must be positive, got {self.mass}")
        if min(self.inertia_xx, self.inertia_yy, self.inertia_zz) <= 0:
            raise ValueError("All inertia values must be positive")
        for name in ("wing_area", "wing_span", "chord"):
            if getattr(self, name) <= 0:
                raise ValueError(f"{name.replace('_', ' ').capitalize()} must be positive, got {getattr(self, name)}")
        if not (0 < self.air_density < 10):
            raise ValueError(f"Invalid air density: {self.air_density} kg/m³")
        if not (0 < self.gravity < 20):
            raise ValueError(f"Invalid gravity: {self.gravity} m/s²")
        if self.max_thrust < 0:
            raise ValueError(f"Max thrust cannot be negative, got {self.max_thrust}")

    def to_block(self) -> np.ndarray:
        b = np.zeros(L.FD_NP, dtype=np.float64)
        b[L.FD_P_MASS], b[L.FD_P_IXX], b[L.FD_P_IYY], b[L.FD_P_IZZ] = (
            self.mass, self.inertia_xx, self.inertia_yy, self.inertia_zz)
        b[L.FD_P_WING_AREA], b[L.FD_P_WING_SPAN], b[L.FD_P_CHORD] = self.wing_area, self.wing_span, self.chord
        b[L.FD_P_CL_0], b[L.FD_P_CL_ALPHA], b[L.FD_P_CD_0], b[L.FD_P_CD_ALPHA2] = (
            self.cl_0, self.cl_alpha, self.cd_0, self.cd_alpha2)
        b[L.FD_P_CL_ELEVATOR], b[L.FD_P_CM_ELEVATOR] = self.cl_elevator, self.cm_elevator
        b[L.FD_P_CY_RUDDER], b[L.FD_P_CN_RUDDER], b[L.FD_P_CL_AILERON] = (
            self.cy_rudder, self.cn_rudder, self.cl_aileron)
        b[L.FD_P_CM_ALPHA], b[L.FD_P_CN_BETA], b[L.FD_P_CL_BETA] = self.cm_alpha, self.cn_beta, self.cl_beta
        b[L.FD_P_DAMPING_ROLL], b[L.FD_P_DAMPING_PITCH], b[L.FD_P_DAMPING_YAW] = (
            self.damping_roll, self.damping_pitch, self.damping_yaw)
        b[L.FD_P_MAX_THRUST], b[L.FD_P_AIR_DENSITY], b[L.FD_P_GRAVITY] = (
            self.max_thrust, self.air_density, self.gravity)
        b[L.FD_P_MIN_AIRSPEED_AERO], b[L.FD_P_MIN_U_VELOCITY] = self.min_airspeed_aero, self.min_u_velocity
        b[L.FD_P_MAX_ELEVATOR_RAD] = np.radians(self.max_elevator_deflection)
        b[L.FD_P_MAX_AILERON_RAD] = np.radians(self.max_aileron_deflection)
        b[L.FD_P_MAX_RUDDER_RAD] = np.radians(self.max_rudder_deflection)
        b[L.FD_P_THRUST_ZERO_VELOCITY] = self.thrust_zero_velocity
        b[L.FD_P_MAX_VELOCITY] = self.max_velocity
        b[L.FD_P_MAX_RATE_RAD] = np.radians(self.max_angular_rate)
        b[L.FD_P_MAX_PITCH_RAD] = np.radians(self.max_pitch_angle)
        b[L.FD_P_MAX_ALPHA_RAD] = np.radians(self.max_alpha)
        b[L.FD_P_MAX_ACCELERATION] = self.max_acceleration
        b[L.FD_P_MAX_ANGULAR_ACCELERATION] = self.max_angular_acceleration
        b[L.FD_P_MAX_TIMESTEP], b[L.FD_P_MIN_TIMESTEP] = self.max_timestep, self.min_timestep
        return b


def aircraft_params_for(aircraft_type: str) -> AircraftParams:
    """simulation/simulation_backend.py:55-80: 'cessna' overrides, anything else is the rc_plane default."""
    if aircraft_type == "cessna":
        return AircraftParams(mass=15.0, inertia_xx=1.0, inertia_yy=2.0, inertia_zz=2.5,
                              wing_area=1.0, wing_span=3.0, max_thrust=80.0)
    return AircraftParams()


def param_table(types) -> np.ndarray:
    """Stack parameter blocks for a heterogeneous fleet: [n_types][FD_NP] float64."""
    return np.stack([(t if isinstance(t, AircraftParams) else aircraft_params_for(t)).to_block() for t in types])
