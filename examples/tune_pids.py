#!/usr/bin/env python3
"""PID gain sweep for altitude hold (the study of the reference's examples/tune_pids.py:59-230: hold 100 m for 30 s from level
flight at 20 m/s, HSA level, score = final altitude error / altitude spread / attitude excursions) -- but instead of five
hand-picked gain sets flown one after the other, a full grid flies at once: one aircraft per gain set, one launch.

    python examples/tune_pids.py [--grid 16] [--duration 30] [--precision mixed]
Sweeps the TECS energy / balance loop gains and the rate-loop kp (grid^3 sets) and prints the best stable ones.
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
from hcrl_amd import config as cfgmod, layout as L  # noqa: E402
from hcrl_amd.agents import AgentFleet  # noqa: E402
from hcrl_amd.flight_types import ControllerConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=16, help="points per swept gain (grid^3 aircraft)")
    ap.add_argument("--duration", type=float, default=30.0)
    ap.add_argument("--precision", default="mixed", choices=["f64", "mixed", "f32"])
    a = ap.parse_args()
    g = a.grid
    n = g ** 3
    balance_kp = np.geomspace(0.01, 0.3, g)              # pitch-from-energy-balance loop (hsa_agent.py:202-212)
    energy_kp = np.geomspace(0.02, 0.6, g)               # throttle-from-total-energy loop (:191-196)
    rate_kp = np.linspace(0.15, 1.5, g)                  # roll / pitch rate loops (rate_agent.py:92-122)
    B, E, R = (m.reshape(-1) for m in np.meshgrid(balance_kp, energy_kp, rate_kp, indexing="ij"))
    base = cfgmod.pid_table(ControllerConfig())
    tables = np.repeat(base[None], n, 0)
    tables[:, L.FD_PID_BALANCE, L.FD_PC_KP] = B
    tables[:, L.FD_PID_ENERGY, L.FD_PC_KP] = E
    tables[:, L.FD_PID_RATE_ROLL, L.FD_PC_KP] = R
    tables[:, L.FD_PID_RATE_PITCH, L.FD_PC_KP] = R

    fleet = AgentFleet(n, a.precision)
    fleet.set_gain_tables(tables)
    x0 = np.zeros((n, 12)); x0[:, 2], x0[:, 3] = -100.0, 20.0
    fleet.reset(x0)
    cmd = np.array([0.0, 20.0, 100.0, 0.0])              # heading 0, 20 m/s, 100 m
    dt, chunk = 0.01, 100
    steps = int(a.duration / dt)
    alts, max_roll, max_pitch = [], torch.zeros(n, device=fleet.device), torch.zeros(n, device=fleet.device)
    torch.cuda.synchronize(); t0 = time.time()
    for k in range(0, steps, chunk):
        fleet.run(L.FD_LEVEL_HSA, cmd, dt, chunk)
        alts.append(-fleet.x[L.FD_X_D].float().clone())
        max_roll = torch.maximum(max_roll, fleet.x[L.FD_X_ROLL].float().abs())
        max_pitch = torch.maximum(max_pitch, fleet.x[L.FD_X_PITCH].float().abs())
    torch.cuda.synchronize(); wall = time.time() - t0
    alts = torch.stack(alts)                             # [samples, n], one per second
    tail = alts[alts.shape[0] // 2:]                     # second half of the run (tune_pids.py:120-135 scores the settled part)
    err = (tail.mean(0) - 100.0).abs()
    std = tail.std(0)
    diverged = (alts.min(0).values < 20.0) | (alts.max(0).values > 300.0) | ~torch.isfinite(alts).all(0)
    stable = ~diverged & (err < 5.0) & (std < 3.0) & (max_roll < np.radians(45)) & (max_pitch < np.radians(30))
    print(f"{n} gain sets x {steps} control steps in {wall:.2f} s ({n * steps / wall:,.0f} aircraft-steps/s, {a.precision})")
    print(f"stable: {int(stable.sum())}   marginal: {int((~stable & ~diverged).sum())}   diverged: {int(diverged.sum())}")
    score = torch.where(stable, err + std, torch.full_like(err, float("inf")))
    best = torch.argsort(score)[:5].cpu().numpy()
    print("best stable sets (balance kp, energy kp, rate kp -> |altitude error|, altitude std):")
    for i in best:
        if np.isfinite(float(score[i])):
            print(f"  {B[i]:.4f}  {E[i]:.4f}  {R[i]:.3f}  ->  {float(err[i]):.3f} m  {float(std[i]):.3f} m")
    d = ControllerConfig()
    print(f"(defaults: balance 0.06, energy 0.12, rate kp roll {d.roll_rate_gains.kp} / pitch {d.pitch_rate_gains.kp})")


if __name__ == "__main__":
    main()
