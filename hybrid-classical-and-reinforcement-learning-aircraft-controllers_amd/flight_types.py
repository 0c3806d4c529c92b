"""Value types that cross the drop-in boundaries (host side).

Field names, defaults and array conventions follow the reference's `controllers/types.py`
(ControlMode :9-24, AircraftState :27-96, Waypoint :99-132, ControlCommand :135-170,
ControlSurfaces :173-201, PIDGains :277-291, ControllerConfig :294-350) so code written against the
reference's dataclasses runs against these unchanged.  Only what the hot path touches is mirrored.
"""
from dataclasses import dataclass, field
from enum import Enum
from typing import Optional

import numpy as np


class ControlMode(Enum):
    WAYPOINT = 1
    HSA = 2
    ATTITUDE = 3
    RATE = 4
    SURFACE = 5


def _z3():
    return np.zeros(3)


@dataclass
class AircraftState:
    """time; position NED [m]; velocity body [m/s]; attitude [roll,pitch,yaw] rad; angular_rate [p,q,r] rad/s;
    plus the four derived scalars the physics computes (airspeed, altitude, ground_speed, heading)."""
    time: float = 0.0
    position: np.ndarray = field(default_factory=_z3)
    velocity: np.ndarray = field(default_factory=_z3)
    attitude: np.ndarray = field(default_factory=_z3)
    angular_rate: np.ndarray = field(default_factory=_z3)
    airspeed: float = 0.0
    altitude: float = 0.0
    ground_speed: float = 0.0
    heading: float = 0.0

    roll = property(lambda s: s.attitude[0])
    pitch = property(lambda s: s.attitude[1])
    yaw = property(lambda s: s.attitude[2])
    p = property(lambda s: s.angular_rate[0])
    q = property(lambda s: s.angular_rate[1])
    r = property(lambda s: s.angular_rate[2])
    north = property(lambda s: s.position[0])
    east = property(lambda s: s.position[1])
    down = property(lambda s: s.position[2])

    def to_vector(self) -> np.ndarray:
        """The 12-word state row (include/fdyn_layout.h FD_X_*)."""
        return np.concatenate([self.position, self.velocity, self.attitude, self.angular_rate]).astype(np.float64)

    @classmethod
    def from_vector(cls, x, derived=None, time=0.0):
        x = np.asarray(x, dtype=np.float64)
        s = cls(time=time, position=x[0:3], velocity=x[3:6], attitude=x[6:9], angular_rate=x[9:12])
        if derived is not None:
            s.airspeed, s.altitude, s.ground_speed, s.heading = (float(v) for v in derived)
        return s


@dataclass
class Waypoint:
    north: float
    east: float
    down: float
    speed: Optional[float] = None
    heading: Optional[float] = None

    @property
    def altitude(self) -> float:
        return -self.down

    @classmethod
    def from_ned(cls, north, east, down, **kw):
        return cls(north=north, east=east, down=down, **kw)

    @classmethod
    def from_altitude(cls, north, east, altitude, **kw):
        return cls(north=north, east=east, down=-altitude, **kw)


@dataclass
class ControlCommand:
    mode: ControlMode
    timestamp: float = 0.0
    waypoint: Optional[Waypoint] = None
    heading: Optional[float] = None
    speed: Optional[float] = None
    altitude: Optional[float] = None
    roll_angle: Optional[float] = None
    pitch_angle: Optional[float] = None
    yaw_angle: Optional[float] = None
    roll_rate: Optional[float] = None
    pitch_rate: Optional[float] = None
    yaw_rate: Optional[float] = None
    throttle: Optional[float] = None
    elevator: Optional[float] = None
    aileron: Optional[float] = None
    rudder: Optional[float] = None


@dataclass
class ControlSurfaces:
    """Normalised deflections; array order is [elevator, aileron, rudder, throttle] (FD_U_*)."""
    elevator: float = 0.0
    aileron: float = 0.0
    rudder: float = 0.0
    throttle: float = 0.0

    def to_array(self) -> np.ndarray:
        return np.array([self.elevator, self.aileron, self.rudder, self.throttle])

    @classmethod
    def from_array(cls, arr):
        return cls(elevator=arr[0], aileron=arr[1], rudder=arr[2], throttle=arr[3])


@dataclass
class PIDGains:
    kp: float = 0.0
    ki: float = 0.0
    kd: float = 0.0
    i_limit: float = 25.0


def _g(kp, ki, kd):
    return field(default_factory=lambda: PIDGains(kp=kp, ki=ki, kd=kd))


@dataclass
class ControllerConfig:
    """Legacy all-in-one gains/limits block (reference controllers/types.py:294-350 defaults)."""
    roll_angle_gains: PIDGains = _g(8.0, 2.0, 0.3)
    pitch_angle_gains: PIDGains = _g(6.0, 1.5, 0.2)
    roll_rate_gains: PIDGains = _g(1.3, 0.4, 0.012)
    pitch_rate_gains: PIDGains = _g(0.6, 0.2, 0.008)
    yaw_gains: PIDGains = _g(1.6, 0.15, 0.01)
    heading_gains: PIDGains = _g(1.0, 0.05, 0.2)
    speed_gains: PIDGains = _g(0.1, 0.01, 0.0)
    altitude_gains: PIDGains = _g(0.2, 0.01, 0.1)
    max_roll: float = 30.0
    max_pitch: float = 30.0
    max_roll_rate: float = 180.0
    max_pitch_rate: float = 180.0
    max_yaw_rate: float = 160.0
    dt: float = 0.01
    rate_loop_dt: float = 0.001
    max_bank_angle_hsa: float = 25.0
    baseline_throttle: float = 0.1
    waypoint_acceptance_radius: float = 40.0
    proportional_navigation_gain: float = 3.0
