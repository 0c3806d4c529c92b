"""GPU-box exploration: per-aircraft forward error of each precision variant vs the oracle, next to the oracle's own
amplification of a 1e-12 initial-state perturbation (3 random directions).  Dumps gpurun_out/drift_detail.npz."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from oracle import oracle as orc
from conftest import rel_err, STATE_ANGLE_COLS
from test_gpu_parity_scale import _cfg2_inputs, N
from hcrl_amd.params import AircraftParams
from hcrl_amd.fleet import BatchedSixDOF
P = AircraftParams().to_block()
x0, u = _cfg2_inputs(N, 20261004)
us = np.ascontiguousarray(u.T)
T = min(16, int(orc.lib.orc_max_threads()))
out = {}
for dt in (0.001, 0.01):
    rs = np.random.RandomState(1)
    base = np.ascontiguousarray(x0.T)
    pert = [np.ascontiguousarray((x0 * (1 + 1e-12 * rs.choice([-1.0, 1.0], x0.shape))).T) for _ in range(3)]
    fls = {}
    for prec in ("f64", "mixed", "f32"):
        fl = BatchedSixDOF(N, prec); fl.reset(x0); fl.set_controls(u); fls[prec] = fl
    amp = np.zeros(N); err = {p: np.zeros(N) for p in fls}
    errt = {p: [] for p in fls}
    for k in range(20):
        orc.lib.orc_sixdof_step_batch(orc.dp(P), orc.dp(base), orc.dp(us), N, dt * 50, 50, T)
        for b in pert:
            orc.lib.orc_sixdof_step_batch(orc.dp(P), orc.dp(b), orc.dp(us), N, dt * 50, 50, T)
            amp = np.maximum(amp, rel_err(b.T, base.T, STATE_ANGLE_COLS).max(1) / 1e-12)
        for p, fl in fls.items():
            for _ in range(50):
                fl.step(dt)
            e = rel_err(fl.state_numpy(), base.T, STATE_ANGLE_COLS).max(1)
            err[p] = np.maximum(err[p], e); errt[p].append(e)
    out[f"amp_{dt}"] = amp
    for p in fls:
        out[f"err_{p}_{dt}"] = err[p]; out[f"errt_{p}_{dt}"] = np.array(errt[p])
        e = err[p]; eta = e / np.maximum(amp, 1.0)
        print(f"dt={dt} {p:5s} forward: p50 {np.percentile(e,50):.2e} p90 {np.percentile(e,90):.2e} p99 {np.percentile(e,99):.2e} max {e.max():.2e}"
              f" | backward eta: p50 {np.percentile(eta,50):.2e} p99 {np.percentile(eta,99):.2e} max {eta.max():.2e}"
              f" | A<=100: n {(amp<=100).sum()} max fwd {e[amp<=100].max():.2e} | A<=1000: n {(amp<=1000).sum()} max fwd {e[amp<=1000].max():.2e}")
np.savez_compressed(os.path.join(R, "gpurun_out", "drift_detail.npz"), **out)
