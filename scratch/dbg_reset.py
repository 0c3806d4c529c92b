import sys, torch, numpy as np
sys.path.insert(0, '.')
import hcrl_amd
from hcrl_amd.rate_env import GpuRateVecEnv
from hcrl_amd import layout as L
torch.set_printoptions(precision=6, linewidth=200)
n = 256
env = GpuRateVecEnv(n, "medium", 10.0, 0.02, "step", seed=1, precision="mixed", sampling="device")
twin = GpuRateVecEnv(n, "medium", 10.0, 0.02, "step", seed=1, precision="mixed", sampling="device")
o1 = env.reset().clone(); o2 = twin.reset().clone()
torch.cuda.synchronize()
for i in (0, 3, 4, 5):
    print("env", i)
    print("  x1", env.x[:, i].tolist())
    print("  x2", twin.x[:, i].tolist())
    print("  e1", env.e[:8, i].tolist())
    print("  e2", twin.e[:8, i].tolist())
print("EC", env.env_consts.tolist(), twin.env_consts.tolist())
rows = (env.x != twin.x).nonzero()
print("which x rows differ:", rows[:, 0].unique().tolist())
rows = (env.e != twin.e).nonzero()
print("which e rows differ:", rows[:, 0].unique().tolist())
