"""Which framework ops launch the fill / copy kernels of a PPO iteration (shapes included)."""
import sys
sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
import hcrl_amd  # noqa
from hcrl_amd.policy import RateLSTMPolicy
from hcrl_amd.ppo import PPOConfig, RecurrentPPO
from hcrl_amd.rate_env import GpuRateVecEnv

N, T, E, MB = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (65536, 16, 2, 2)))      # envs, steps, epochs, slices
env = GpuRateVecEnv(N, "easy", 10.0, 0.02, "step", seed=0, precision="mixed", sampling="device")
ppo = RecurrentPPO(env, RateLSTMPolicy(compute_dtype=torch.bfloat16), PPOConfig(n_steps=T, n_epochs=E, n_minibatches=MB), seed=0)
ppo.use_update_graph = False           # eager: the profiler attributes kernels to ops
for _ in range(2):
    ppo.collect_rollout(); ppo.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ppo.collect_rollout(); ppo.update()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if any(k in e.key for k in ("fill_", "zero_", "zeros", "copy_", "ones", "clone", "contiguous"))]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.key:28s} calls {e.count:4d} dev_us {e.device_time_total:10.1f}  shapes {str(e.input_shapes)[:110]}")

print("---- top ops by device time (self) ----")
allrows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in allrows)
print("total device us", tot)
for e in allrows[:44]:
    if e.key.startswith("void ") or "Cijk" in e.key or "kernel" in e.key:
        continue
    print(f"{e.key[:34]:34s} calls {e.count:4d} self_dev_us {e.self_device_time_total:10.1f}  shapes {str(e.input_shapes)[:100]}")
