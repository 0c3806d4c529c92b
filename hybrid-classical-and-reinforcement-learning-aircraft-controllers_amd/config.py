"""Controller / mission configuration and its flattening into the C-ABI tables.

Dataclasses and YAML schema follow the reference's controllers/config_loader.py (GuidanceConfig :40-62,
TECSConfig :65-72, HSAConfig :75-80, FlightControlConfig :83-109, load_controller_config :112-207,
MissionConfig :210-221, load_mission_config :224-260) so its YAML files load unchanged.

`pid_table` / `cascade_consts` reproduce which config feeds which PID in the cascade (the aliasing noted in
SURVEY §7): attitude and rate loops take gains AND limits from the legacy ControllerConfig
(attitude_agent.py:43-76, rate_agent.py:41-55); heading/TECS/guidance come from FlightControlConfig
(hsa_agent.py:48-109, waypoint_agent.py:50-65); the heading integral limit is ControllerConfig.heading_gains.i_limit
(hsa_agent.py:63-64) and the HSA pitch clamp is a hard-coded 10 deg (hsa_agent.py:109).
Gains are narrowed to float32 exactly where the reference's pybind11 module narrows Python doubles.
"""
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional, Sequence

import numpy as np
import yaml

from . import layout as L
from .flight_types import ControllerConfig, Waypoint

CONFIG_DIR = Path(__file__).parent / "configs"


@dataclass
class Gains3:
    kp: float = 0.0
    ki: float = 0.0
    kd: float = 0.0

    @classmethod
    def from_dict(cls, d):
        return cls(kp=d.get("kp", 0.0), ki=d.get("ki", 0.0), kd=d.get("kd", 0.0))


@dataclass
class GuidanceConfig:
    acceptance_radius: float = 40.0
    lookahead_time: float = 1.2
    lookahead_min: float = 12.0
    lookahead_max: float = 30.0
    proximity_scale_distance: float = 2.0
    los_max_bank: float = 20.0
    los_lead_angle: float = 30.0
    turn_threshold_angle: float = 60.0
    turn_threshold_distance: float = 60.0
    max_speed_reduction: float = 0.15
    min_speed: float = 14.0
    pn_gain: float = 3.0


@dataclass
class TECSConfig:
    energy_gains: Gains3 = field(default_factory=lambda: Gains3(0.12, 0.03, 0.03))
    balance_gains: Gains3 = field(default_factory=lambda: Gains3(0.06, 0.005, 0.03))
    max_pitch_command: float = 15.0
    baseline_throttle: float = 0.1
    load_factor_gain: float = 0.05


@dataclass
class HSAConfig:
    heading_gains: Gains3 = field(default_factory=lambda: Gains3(1.0, 0.05, 0.2))
    max_bank_angle: float = 25.0
    tecs: TECSConfig = field(default_factory=TECSConfig)


@dataclass
class FlightControlConfig:
    outer_loop_dt: float = 0.01
    inner_loop_dt: float = 0.001
    roll_rate_gains: Gains3 = field(default_factory=lambda: Gains3(1.3, 0.4, 0.012))
    pitch_rate_gains: Gains3 = field(default_factory=lambda: Gains3(0.6, 0.2, 0.008))
    yaw_rate_gains: Gains3 = field(default_factory=lambda: Gains3(1.6, 0.15, 0.01))
    max_roll_rate: float = 200.0
    max_pitch_rate: float = 100.0
    max_yaw_rate: float = 60.0
    roll_angle_gains: Gains3 = field(default_factory=lambda: Gains3(8.0, 2.0, 0.3))
    pitch_angle_gains: Gains3 = field(default_factory=lambda: Gains3(6.0, 1.5, 0.2))
    max_roll: float = 30.0
    max_pitch: float = 20.0
    hsa: HSAConfig = field(default_factory=HSAConfig)
    guidance: GuidanceConfig = field(default_factory=GuidanceConfig)


def _resolve(name, sub):
    p = Path(name)
    return p if p.is_absolute() or p.exists() else CONFIG_DIR / sub / name


def load_controller_config(config_file: str = "cascaded_pid.yaml") -> FlightControlConfig:
    path = _resolve(config_file, "controllers")
    cfg = FlightControlConfig()
    if not path.exists():
        print(f"Warning: Config file {path} not found, using defaults")
        return cfg
    data = yaml.safe_load(open(path)) or {}
    t = data.get("timing", {})
    cfg.outer_loop_dt = t.get("outer_loop_dt", 0.01)
    cfg.inner_loop_dt = t.get("inner_loop_dt", 0.001)
    rc = data.get("rate_control", {})
    for axis, attr in (("roll", "roll_rate_gains"), ("pitch", "pitch_rate_gains"), ("yaw", "yaw_rate_gains")):
        if axis in rc:
            setattr(cfg, attr, Gains3.from_dict(rc[axis]))
    if "limits" in rc:
        cfg.max_roll_rate = rc["limits"].get("max_roll_rate", 200.0)
        cfg.max_pitch_rate = rc["limits"].get("max_pitch_rate", 100.0)
        cfg.max_yaw_rate = rc["limits"].get("max_yaw_rate", 60.0)
    ac = data.get("attitude_control", {})
    if "roll" in ac:
        cfg.roll_angle_gains = Gains3.from_dict(ac["roll"])
    if "pitch" in ac:
        cfg.pitch_angle_gains = Gains3.from_dict(ac["pitch"])
    if "limits" in ac:
        cfg.max_roll = ac["limits"].get("max_roll", 30.0)
        cfg.max_pitch = ac["limits"].get("max_pitch", 20.0)
    hc = data.get("hsa_control", {})
    if "heading" in hc:
        cfg.hsa.heading_gains = Gains3.from_dict(hc["heading"])
    if "limits" in hc:
        cfg.hsa.max_bank_angle = hc["limits"].get("max_bank_angle", 25.0)
        cfg.hsa.tecs.baseline_throttle = hc["limits"].get("baseline_throttle", 0.1)
        cfg.hsa.tecs.max_pitch_command = hc["limits"].get("max_pitch_command", 15.0)
    tecs = hc.get("tecs", {})
    if "energy" in tecs:
        cfg.hsa.tecs.energy_gains = Gains3.from_dict(tecs["energy"])
    if "balance" in tecs:
        cfg.hsa.tecs.balance_gains = Gains3.from_dict(tecs["balance"])
    if "turn_compensation" in hc:
        cfg.hsa.tecs.load_factor_gain = hc["turn_compensation"].get("load_factor_gain", 0.05)
    wn = data.get("waypoint_navigation")
    if wn is not None:
        g = cfg.guidance
        g.acceptance_radius = wn.get("acceptance_radius", 40.0)
        pp = wn.get("pure_pursuit", {})
        if pp:
            g.lookahead_time = pp.get("lookahead_time", 1.2)
            g.lookahead_min = pp.get("lookahead_min", 12.0)
            g.lookahead_max = pp.get("lookahead_max", 30.0)
            g.proximity_scale_distance = pp.get("proximity_scale_distance", 2.0)
        los = wn.get("los", {})
        if los:
            g.los_max_bank = los.get("max_bank_for_turn_calc", 20.0)
            g.los_lead_angle = los.get("lead_angle", 30.0)
        sc = wn.get("speed_control", {})
        if sc:
            g.turn_threshold_angle = sc.get("turn_threshold_angle", 60.0)
            g.turn_threshold_distance = sc.get("turn_threshold_distance", 60.0)
            g.max_speed_reduction = sc.get("max_speed_reduction", 0.15)
            g.min_speed = sc.get("min_speed", 14.0)
        if "proportional_nav" in wn:
            g.pn_gain = wn["proportional_nav"].get("gain", 3.0)
    return cfg


@dataclass
class MissionConfig:
    name: str = "Unnamed Mission"
    pattern_type: str = "square"
    pattern_size: float = 300.0
    altitude: float = 100.0
    speed: float = 15.0
    guidance: str = "PP"
    max_duration: float = 180.0
    dt: float = 0.01
    output_dir: str = "final_figures"


def load_mission_config(config_file: str = "square_pattern.yaml") -> MissionConfig:
    path = _resolve(config_file, "missions")
    cfg = MissionConfig()
    if not path.exists():
        print(f"Warning: Config file {path} not found, using defaults")
        return cfg
    data = yaml.safe_load(open(path)) or {}
    m = data.get("mission", {})
    if m:
        cfg.name = m.get("name", "Unnamed Mission")
        cfg.pattern_type = m.get("pattern", {}).get("type", "square") if "pattern" in m else cfg.pattern_type
        cfg.pattern_size = m.get("pattern", {}).get("size", 300.0) if "pattern" in m else cfg.pattern_size
        fl = m.get("flight")
        if fl:
            cfg.altitude = fl.get("altitude", 100.0)
            cfg.speed = fl.get("speed", 15.0)
            cfg.guidance = fl.get("guidance", "PP")
    s = data.get("simulation")
    if s:
        cfg.max_duration = s.get("max_duration", 180.0)
        cfg.dt = s.get("dt", 0.01)
    if "output" in data:
        cfg.output_dir = data["output"].get("directory", "final_figures")
    return cfg


def square_mission(size: float, altitude: float, speed: float):
    """The 5-waypoint counter-clockwise square of examples/03_waypoint_square_demo.py:60-70."""
    pts = [(0, 0), (size, 0), (size, size), (0, size), (0, 0)]
    return [Waypoint.from_altitude(n, e, altitude, speed=speed) for n, e in pts]


def waypoint_table(waypoints: Sequence[Waypoint]) -> np.ndarray:
    """[n_wp][FD_NWP] float64 rows (north, east, altitude, speed; NaN speed = 'keep current airspeed')."""
    if not waypoints:
        raise ValueError("Mission must have at least one waypoint")
    if len(waypoints) > L.FD_MAX_WAYPOINTS:
        raise ValueError(f"at most {L.FD_MAX_WAYPOINTS} waypoints per mission")
    t = np.zeros((len(waypoints), L.FD_NWP), dtype=np.float64)
    for i, w in enumerate(waypoints):
        t[i] = [w.north, w.east, w.altitude, np.nan if w.speed is None else w.speed]
    return t


def _pid_row(kp, ki, kd, i_limit_lo, i_limit_hi, out_lo, out_hi, alpha=0.1):
    return np.array([kp, ki, kd, out_lo, out_hi, i_limit_lo, i_limit_hi, alpha], dtype=np.float64).astype(np.float32)


def pid_table(config: Optional[ControllerConfig] = None,
              flight_config: Optional[FlightControlConfig] = None) -> np.ndarray:
    """[FD_NPID][FD_NPC] float32 PID configs of the cascade."""
    c = config or ControllerConfig()
    hsa = flight_config.hsa if flight_config is not None else HSAConfig()
    tecs = hsa.tecs
    t = np.zeros((L.FD_NPID, L.FD_NPC), dtype=np.float32)

    def legacy(g, lo=-1.0, hi=1.0):           # controllers/utils/pid_utils.py:16-46
        return _pid_row(g.kp, g.ki, g.kd, -g.i_limit, g.i_limit, lo, hi)

    t[L.FD_PID_RATE_ROLL] = legacy(c.roll_rate_gains)
    t[L.FD_PID_RATE_PITCH] = legacy(c.pitch_rate_gains)
    t[L.FD_PID_RATE_YAW] = legacy(c.yaw_gains)
    rr, pr, yr = np.radians(c.max_roll_rate), np.radians(c.max_pitch_rate), np.radians(c.max_yaw_rate)
    t[L.FD_PID_ATT_ROLL] = legacy(c.roll_angle_gains, -rr, rr)
    t[L.FD_PID_ATT_PITCH] = legacy(c.pitch_angle_gains, -pr, pr)
    t[L.FD_PID_ATT_YAW] = legacy(c.yaw_gains, -yr, yr)
    bank = np.radians(hsa.max_bank_angle)
    hg = hsa.heading_gains
    t[L.FD_PID_HEADING] = _pid_row(hg.kp, hg.ki, hg.kd, -c.heading_gains.i_limit, c.heading_gains.i_limit, -bank, bank)
    eg, bg = tecs.energy_gains, tecs.balance_gains
    t[L.FD_PID_ENERGY] = _pid_row(eg.kp, eg.ki, eg.kd, -10.0, 10.0, -0.5, 0.5)
    mp = np.radians(tecs.max_pitch_command)
    t[L.FD_PID_BALANCE] = _pid_row(bg.kp, bg.ki, bg.kd, -5.0, 5.0, -mp, mp)
    return t


_GUIDANCE = {"LOS": L.FD_GUIDANCE_LOS, "PP": L.FD_GUIDANCE_PP, "PURE_PURSUIT": L.FD_GUIDANCE_PP}


def cascade_consts(config: Optional[ControllerConfig] = None,
                   flight_config: Optional[FlightControlConfig] = None,
                   guidance_type: str = "LOS", acceptance_radius: Optional[float] = None,
                   on_complete: str = "freeze", pid_throttle: float = 0.6, pid_dt: float = 0.0) -> np.ndarray:
    """[FD_NC] float64 glue constants of the cascade.  pid_throttle / pid_dt: what the env kernels' fused rate-PID
    driver holds / hands its PIDs (0.6 and the env dt for demonstrations, 0.5 and rate_loop_dt for eval_rate.py)."""
    c = config or ControllerConfig()
    hsa = flight_config.hsa if flight_config is not None else HSAConfig()
    g = flight_config.guidance if flight_config is not None else GuidanceConfig()
    C = np.zeros(L.FD_NC, dtype=np.float64)
    C[L.FD_C_MAX_ROLL_RATE] = np.radians(c.max_roll_rate)
    C[L.FD_C_MAX_PITCH_RATE] = np.radians(c.max_pitch_rate)
    C[L.FD_C_MAX_YAW_RATE] = np.radians(c.max_yaw_rate)
    C[L.FD_C_MAX_ROLL] = np.radians(c.max_roll)
    C[L.FD_C_MAX_PITCH] = np.radians(c.max_pitch)
    C[L.FD_C_MAX_BANK_RAD] = np.radians(hsa.max_bank_angle)
    C[L.FD_C_BASELINE_THROTTLE] = hsa.tecs.baseline_throttle
    C[L.FD_C_LOAD_FACTOR_GAIN] = hsa.tecs.load_factor_gain
    C[L.FD_C_MAX_PITCH_CMD_RAD] = np.radians(10.0)
    C[L.FD_C_GUIDANCE_TYPE] = _GUIDANCE.get(guidance_type, L.FD_GUIDANCE_DEFAULT)
    C[L.FD_C_WP_MAX_BANK_RAD] = np.radians(hsa.max_bank_angle if flight_config is not None else 25.0)
    C[L.FD_C_LOS_MAX_BANK_RAD] = np.radians(g.los_max_bank)
    C[L.FD_C_LOS_LEAD_ANGLE_RAD] = np.radians(g.los_lead_angle)
    C[L.FD_C_LOOKAHEAD_TIME] = g.lookahead_time
    C[L.FD_C_LOOKAHEAD_MIN] = g.lookahead_min
    C[L.FD_C_LOOKAHEAD_MAX] = g.lookahead_max
    C[L.FD_C_PROXIMITY_SCALE] = g.proximity_scale_distance
    C[L.FD_C_TURN_THRESHOLD_DIST] = g.turn_threshold_distance
    C[L.FD_C_TURN_THRESHOLD_ANGLE_RAD] = np.radians(g.turn_threshold_angle)
    C[L.FD_C_MAX_SPEED_REDUCTION] = g.max_speed_reduction
    C[L.FD_C_MIN_SPEED] = g.min_speed
    C[L.FD_C_ACCEPTANCE_RADIUS] = g.acceptance_radius if acceptance_radius is None else acceptance_radius
    C[L.FD_C_ON_COMPLETE] = {"freeze": 0, "restart": 1}[on_complete]
    C[L.FD_C_PID_THROTTLE] = pid_throttle
    C[L.FD_C_PID_DT] = pid_dt
    return C
