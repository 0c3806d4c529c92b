"""Upper bound of the "two lanes per aircraft" restructuring of the env step (round-2 review item 4), measured with the kernel as it is.

Splitting one aircraft's `_dynamics` over a translational and a rotational lane turns 65 536 envs into 2048 waves (two per SIMD),
each carrying HALF an aircraft's arithmetic.  The most such a split can give is what the existing kernel does when every wave
really has half the work and two waves share a SIMD: 131 072 envs x 10 RK4 sub-steps (env dt = 0.01) against 65 536 envs x 20
(env dt = 0.02) -- the same number of aircraft-sub-steps per launch, no exchange between lanes, nothing computed twice."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcrl_amd.rate_env import GpuRateVecEnv
def timed(n, dt, steps=300):
    env = GpuRateVecEnv(n, "medium", 10.0, dt, "step", seed=0, precision="mixed", sampling="device")
    env.reset()
    g = torch.Generator(device=env.device).manual_seed(1)
    acts = [torch.cat([(torch.rand((n, 3), device=env.device, generator=g) - 0.5) * 0.8, 0.3 + 0.5 * torch.rand((n, 1), device=env.device, generator=g)], 1).contiguous() for _ in range(8)]
    for k in range(60): env.step_device(acts[k % 8])
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for k in range(steps): env.step_device(acts[k % 8])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
rows = []
for n, dt in ((65536, 0.02), (131072, 0.01), (65536, 0.01), (131072, 0.02)):
    us = min(timed(n, dt) for _ in range(3))
    rows.append((n, dt, us))
    print(f"{n:7d} envs x {int(round(dt / 0.001)):2d} RK4 sub-steps per launch: {us:7.2f} us   {n * dt / 0.001 / us / 1e3:7.2f} G aircraft-sub-steps/s", flush=True)
base, split = rows[0][2], rows[1][2]
print(f"perfect two-lane split (no exchange, no redundant work, register-capped build): {base:.2f} -> {split:.2f} us = {base / split:.2f} x")
