#!/usr/bin/env python3
"""cfg-2 AT SCALE, from the reference itself: 1024 distinct aircraft x 1000 steps of Simplified6DOF.step at dt = 1 ms and
dt = 10 ms (SURVEY §8d recipe, the inputs of tests/test_gpu_parity_scale.py::_cfg2_inputs, first 1024 rows).

Runs only in the build container (needs /root/reference and `make -C oracle ref`); ~2 minutes on 8 cores.
Writes tests/golden/cfg2_scale_1024.npz: x0, ctrl, and the reference states after 250 / 500 / 750 / 1000 steps for both
step sizes.  DATA only (inputs + expected outputs).
"""
import multiprocessing as mp
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
N_REF, SEED, CHECKPOINTS = 1024, 20261004, (250, 500, 750, 1000)


def inputs(n, seed):
    """Identical to tests/test_gpu_parity_scale.py::_cfg2_inputs (kept in step by tests/test_oracle_scale.py)."""
    rs = np.random.RandomState(seed)
    x0 = np.zeros((n, 12))
    x0[:, 3] = rs.uniform(15.0, 30.0, n)
    x0[:, 2] = -rs.uniform(50.0, 200.0, n)
    x0[:, 6] = rs.uniform(-np.radians(15), np.radians(15), n)
    x0[:, 7] = rs.uniform(-np.radians(15), np.radians(15), n)
    x0[:, 8] = rs.uniform(0.0, 2 * np.pi, n)
    x0[:, 9:12] = rs.uniform(-0.1, 0.1, (n, 3))
    u = np.concatenate([rs.uniform(-0.3, 0.3, (n, 3)), rs.uniform(0.3, 0.9, (n, 1))], 1)
    return x0, u


def fly(args):
    lo, hi, x0, u = args
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))
    sys.path.insert(0, REF)
    from controllers.types import AircraftState, ControlSurfaces
    from simulation.simplified_6dof import AircraftParams, Simplified6DOF
    out = np.zeros((2, hi - lo, len(CHECKPOINTS), 12))
    for d, dt in enumerate((0.001, 0.01)):
        for i in range(lo, hi):
            x = x0[i]
            sim = Simplified6DOF(AircraftParams())
            sim.reset(AircraftState(time=0.0, position=np.array(x[0:3]), velocity=np.array(x[3:6]), attitude=np.array(x[6:9]),
                                    angular_rate=np.array(x[9:12]), airspeed=float(np.linalg.norm(x[3:6])), altitude=float(-x[2])))
            sim.set_controls(ControlSurfaces(elevator=u[i, 0], aileron=u[i, 1], rudder=u[i, 2], throttle=u[i, 3]))
            c = 0
            for k in range(1, CHECKPOINTS[-1] + 1):
                sim.step(dt)
                if k == CHECKPOINTS[c]:
                    out[d, i - lo, c] = np.array(sim._state, dtype=np.float64)
                    c += 1
    return lo, out


if __name__ == "__main__":
    x0, u = inputs(4096, SEED)
    x0, u = x0[:N_REF], u[:N_REF]
    procs = int(os.environ.get("PROCS", "8"))
    step = N_REF // (procs * 4)
    jobs = [(lo, min(lo + step, N_REF), x0, u) for lo in range(0, N_REF, step)]
    states = np.zeros((2, N_REF, len(CHECKPOINTS), 12))
    with mp.Pool(procs) as pool:
        for lo, out in pool.imap_unordered(fly, jobs):
            states[:, lo:lo + out.shape[1]] = out
    path = os.path.join(REPO, "tests", "golden", "cfg2_scale_1024.npz")
    np.savez_compressed(path, x0=x0, ctrl=u, checkpoints=np.array(CHECKPOINTS), dts=np.array([0.001, 0.01]),
                        states_dt0p001=states[0], states_dt0p01=states[1], seed=SEED)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")
