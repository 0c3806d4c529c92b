#!/usr/bin/env python3
"""BASELINE config 1: one aircraft, rate-PID loop (the harness of the reference's examples/01_hello_controls.py:55-121)
over the drop-in objects -- `SimulationAircraftBackend` (HIP physics, 10 sub-steps of 1 ms per 10 ms control step) and the
`aircraft_controls_bindings`-compatible PID module (compat/).  Level flight at 100 m / 20 m/s, roll-rate command 30 deg/s,
throttle 0.7, 500 steps.  Prints the checkpoints SURVEY.md 8a recorded from the reference.

    python examples/01_hello_controls.py            # needs an MI355X: there is no CPU fallback
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "compat")]
import aircraft_controls_bindings as acb  # noqa: E402
import hcrl_amd  # noqa: E402,F401
from hcrl_amd.backend import SimulationAircraftBackend  # noqa: E402
from hcrl_amd.flight_types import AircraftState, ControllerConfig, ControlSurfaces  # noqa: E402


def pid_config(g):
    c = acb.PIDConfig()
    c.gains = acb.PIDGains(g.kp, g.ki, g.kd)
    c.integral_min, c.integral_max = -g.i_limit, g.i_limit
    return c


def main(steps=500, dt=0.01):
    cfg = ControllerConfig()
    rate = acb.MultiAxisPIDController(pid_config(cfg.roll_rate_gains), pid_config(cfg.pitch_rate_gains), pid_config(cfg.yaw_gains))
    backend = SimulationAircraftBackend({"aircraft_type": "rc_plane"})
    state = backend.reset(AircraftState(position=np.array([0.0, 0.0, -100.0]), velocity=np.array([20.0, 0.0, 0.0]),
                                        airspeed=20.0, altitude=100.0))
    cmd = np.array([np.radians(30.0), 0.0, 0.0])
    lim = np.radians([cfg.max_roll_rate, cfg.max_pitch_rate, cfg.max_yaw_rate])
    for i in range(steps):
        sp = np.clip(cmd, -lim, lim)
        o = rate.compute(acb.Vector3(*sp), acb.Vector3(state.p, state.q, state.r), dt)       # rate_agent.py:92-122
        surf = ControlSurfaces(aileron=float(np.clip(o.roll, -1, 1)), elevator=float(np.clip(-o.pitch, -1, 1)),
                               rudder=float(np.clip(-o.yaw, -1, 1)), throttle=0.7)
        backend.set_controls(surf)
        state = backend.step(dt)
        if i in (0, 100, steps - 1):
            print(f"step {i:3d}: p = {np.degrees(state.p):+.6f} deg/s  r = {np.degrees(state.r):+.6f} deg/s  "
                  f"roll = {np.degrees(state.roll):+.6f} deg  aileron = {surf.aileron:+.6f}  rudder = {surf.rudder:+.6f}")
    print("reference (SURVEY 8a): step 0 p=60.821838 aileron=0.745605; step 100 p=65.718980 roll=18.493619; "
          "step 499 p=-129.185359 r=-48.746263 roll=10.117667 rudder=1.0")


if __name__ == "__main__":
    main()
