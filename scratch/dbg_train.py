import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcrl_amd
from hcrl_amd.policy import RateLSTMPolicy
from hcrl_amd.ppo import PPOConfig, RecurrentPPO
from hcrl_amd.rate_env import GpuRateVecEnv
n = int(sys.argv[1]); mbs = int(sys.argv[2])
env = GpuRateVecEnv(n, "easy", 10.0, 0.02, "step", seed=0, precision="mixed", sampling="device")
m = RecurrentPPO(env, RateLSTMPolicy(compute_dtype=torch.bfloat16), PPOConfig(n_steps=16, n_epochs=1, n_minibatches=mbs), seed=1)
m.cfg.n_epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pol = m.policy
_pi = pol.prepare_inference
def dbg_prepare():
    fe = pol.features_extractor
    for name, p in pol.named_parameters():
        print("  touch", name, tuple(p.shape), p.dtype, p.is_contiguous(), flush=True)
        s_ = float(p.detach().float().abs().sum()); torch.cuda.synchronize()
    print("  params ok", flush=True)
    _pi(); torch.cuda.synchronize(); print("  prepare ok", flush=True)
pol.prepare_inference = dbg_prepare
for it in range(int(sys.argv[4]) if len(sys.argv) > 4 else 1):
    print("rollout", it, flush=True); m.collect_rollout(); torch.cuda.synchronize(); print("rollout ok", flush=True)
    print("update", it, flush=True); m.update(); torch.cuda.synchronize(); print("update ok", m.last_stats["value_loss"], flush=True)
