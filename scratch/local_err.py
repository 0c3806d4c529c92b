"""GPU box: per-step LOCAL error of the mixed variant (restart from the oracle state every step) for the outlier aircraft."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from oracle import oracle as orc
from conftest import rel_err, STATE_ANGLE_COLS
from test_gpu_parity_scale import _cfg2_inputs, N
from hcrl_amd.params import AircraftParams
from hcrl_amd.fleet import BatchedSixDOF
np.set_printoptions(linewidth=200, precision=6, suppress=False)
P = AircraftParams().to_block()
x0, u = _cfg2_inputs(N, 20261004)
for dt, ids in ((0.001, [2926, 710, 1996, 5]), (0.01, [3083, 859, 2030, 3314, 5])):
    n = len(ids)
    fl = BatchedSixDOF(n, "mixed"); fl.set_controls(u[ids])
    f64 = BatchedSixDOF(n, "f64"); f64.set_controls(u[ids])
    xs = [x0[i].copy() for i in ids]
    cs = [orc.clip_controls(u[i]) for i in ids]
    loc = np.zeros((1000, n)); loc64 = np.zeros((1000, n)); states = np.zeros((1001, n, 12))
    states[0] = np.array(xs)
    for k in range(1000):
        fl.reset(np.array(xs)); fl.step(dt); g = fl.state_numpy()
        f64.reset(np.array(xs)); f64.step(dt); g64 = f64.state_numpy()
        for j in range(n):
            orc.rk4_step(P, xs[j], cs[j], dt)
        o = np.array(xs); states[k + 1] = o
        loc[k] = rel_err(g, o, STATE_ANGLE_COLS).max(1); loc64[k] = rel_err(g64, o, STATE_ANGLE_COLS).max(1)
    for j, i in enumerate(ids):
        k = int(np.argmax(loc[:, j]))
        print(f"dt {dt} ac {i}: local err median {np.median(loc[:, j]):.2e} p99 {np.percentile(loc[:, j], 99):.2e} max {loc[k, j]:.2e} at step {k}; f64 local max {loc64[:, j].max():.2e}; steps with local > 1e-6: {(loc[:, j] > 1e-6).sum()}")
        print("    state before worst step:", states[k, j])
        print("    oracle after          :", states[k + 1, j])
        fl.reset(states[k][None, j].repeat(n, 0)); fl.step(dt); print("    mixed after           :", fl.state_numpy()[0])
