import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcrl_amd.rate_env import GpuRateVecEnv
torch.manual_seed(0)
for n in (512, 65536, 131072):
    for diff, cmd in (("easy", "step"), ("medium", "step"), ("hard", "random"), ("medium", "ramp"), ("medium", "sine")):
        for pid in (False, True):
            env = GpuRateVecEnv(n, diff, 10.0, 0.02, cmd, seed=3, precision="mixed", sampling="device")
            env.reset()
            bad = None
            for k in range(300):
                a = None if pid else torch.randn(n, 4, device=env.device)
                obs, rew, term, trunc = env.step_device(a)
                if not (torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(env.x).all() and torch.isfinite(env.e).all()):
                    bad = k; break
            print(n, diff, cmd, "pid" if pid else "policy", "first non-finite step:", bad)
