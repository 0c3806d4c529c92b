"""PCIe-inclusive rate of the SB3-style vec-env surface: numpy actions in, numpy obs / rewards / dones out, every step."""
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
import hcrl_amd  # noqa
from hcrl_amd.rate_env import GpuRateVecEnv

for n in (16384, 65536, 262144):
    env = GpuRateVecEnv(n, "medium", 10.0, 0.02, "step", seed=0, precision="mixed", sampling="device", numpy_io=True)
    env.reset()
    acts = np.concatenate([(np.random.rand(n, 3).astype(np.float32) - 0.5) * 0.6, 0.4 + 0.4 * np.random.rand(n, 1).astype(np.float32)], 1)
    for _ in range(20):
        out = env.step(acts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        out = env.step(acts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    assert isinstance(out[0], np.ndarray) and out[0].shape == (n, 18)
    print(f"{n} envs: {dt * 1e6:.0f} us per step with host arrays both ways = {n / dt:.3e} env-steps/s "
          f"({(acts.nbytes + out[0].nbytes + out[1].nbytes + out[2].nbytes) / 1e6:.1f} MB over PCIe per step)")
