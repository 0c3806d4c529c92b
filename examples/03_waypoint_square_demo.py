#!/usr/bin/env python3
"""BASELINE config 3: the 5-level cascade flying the 300 m square (the harness of the reference's
examples/03_waypoint_square_demo.py:60-215: pure-pursuit guidance, 15 m/s, 100 m, acceptance radius 40 m, dt 0.01 with a
single RK4 per control step) -- for ONE aircraft, or a fleet of N flying it side by side in one fused launch per 10 steps.

    python examples/03_waypoint_square_demo.py [--aircraft 65536] [--precision f64|mixed|f32]
Prints the waypoint-arrival times of aircraft 0 (reference, SURVEY 8a: 0.00, 17.21, 31.35, 41.06, 50.79 s) and the fleet's
throughput.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import hcrl_amd  # noqa: E402,F401
from hcrl_amd import config as cfgmod  # noqa: E402
from hcrl_amd.fleet import BatchedCascade  # noqa: E402
from hcrl_amd.flight_types import ControllerConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--aircraft", type=int, default=1)
    ap.add_argument("--precision", default="f64", choices=["f64", "mixed", "f32"])
    ap.add_argument("--max-steps", type=int, default=6000)
    a = ap.parse_args()
    fc = cfgmod.load_controller_config("cascaded_pid.yaml")
    mc = cfgmod.load_mission_config("square_pattern.yaml")
    wps = cfgmod.square_mission(mc.pattern_size, mc.altitude, mc.speed)
    fleet = BatchedCascade(a.aircraft, wps, a.precision, ControllerConfig(), fc, guidance_type=mc.guidance)
    n = a.aircraft
    x0 = np.zeros((n, 12))
    x0[:, 2], x0[:, 3] = -mc.altitude, mc.speed
    if n > 1:                                            # SURVEY 8d cfg 3: per-aircraft offsets around the start
        rs = np.random.RandomState(0)
        x0[1:, 0:2] = rs.uniform(-20, 20, (n - 1, 2))
        x0[1:, 8] = rs.uniform(-0.1745, 0.1745, n - 1)
    fleet.reset(x0)
    dt, chunk = 0.01, 10
    reached, events = 0, []
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(0, a.max_steps, chunk):
        fleet.run(dt, chunk)
        r0 = int(fleet.reached_total[0])
        if r0 > reached:
            events.append((fleet.time, r0))
            reached = r0
        if bool(fleet.mission_complete().all()):
            break
    torch.cuda.synchronize()
    wall = time.time() - t0
    steps = k + chunk
    print(f"aircraft 0 reached waypoints at (10-step resolution): " + ", ".join(f"#{i} @ {t:.2f} s" for t, i in events))
    x = fleet.state_numpy()[0]
    print(f"aircraft 0 final N/E/alt/V = {x[0]:.3f} / {x[1]:.3f} / {-x[2]:.3f} / {np.linalg.norm(x[3:6]):.3f}   "
          f"missions complete: {int(fleet.mission_complete().sum())}/{n}")
    print(f"{steps} control steps x {n} aircraft in {wall:.3f} s = {steps * n / wall:,.0f} aircraft-steps/s ({a.precision})")


if __name__ == "__main__":
    main()
