"""Which parameter gradients differ between the captured PPO update (hipGraph replay) and the eager pass on the same inputs?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcrl_amd.rate_env import GpuRateVecEnv
from hcrl_amd.policy import RateLSTMPolicy
from hcrl_amd.ppo import PPOConfig, RecurrentPPO
torch.manual_seed(0)
N, T, MB = int(os.environ.get("N", 512)), 8, 4
env = GpuRateVecEnv(N, "easy", 10.0, 0.02, "step", seed=3, precision="mixed", sampling="device")
pol = RateLSTMPolicy(compute_dtype=torch.bfloat16)
ppo = RecurrentPPO(env, pol, PPOConfig(n_steps=T, n_epochs=1, n_minibatches=MB), seed=1)
ppo.collect_rollout()
mb = N // MB
ug = ppo._build_update_graph(mb)
names = [n for n, p in pol.named_parameters() if p.requires_grad]
for rep in range(5):
    idx = torch.randperm(N, device=env.device)[:mb]
    for k, src in (("obs", ppo.buf_obs), ("act", ppo.buf_act), ("starts", ppo.buf_start), ("adv", ppo.adv), ("ret", ppo.ret),
                   ("old_logp", ppo.buf_logp), ("old_v", ppo.buf_val)):
        torch.index_select(src, 1, idx, out=ug[k])
    for dst, src in zip(ug["states"], ppo.rollout_states):
        torch.index_select(src, 0, idx, out=dst)
    ug["graph"].replay()
    torch.cuda.synchronize()
    gg = {n: p.grad.detach().clone() for n, p in pol.named_parameters() if p.grad is not None}
    st_g = ug["stats"].clone()
    ppo.flat.zero()
    loss, st = ppo._minibatch_loss(ug["obs"], ug["act"], ug["starts"], ug["adv"], ug["ret"], ug["old_logp"], ug["old_v"], ug["states"])
    loss.backward()
    torch.cuda.synchronize()
    bad = []
    for n, p in pol.named_parameters():
        if p.grad is None: continue
        a, b = gg[n].float(), p.grad.float()
        err = float((a - b).norm() / (b.norm() + 1e-12)) if torch.isfinite(a).all() else float("nan")
        if not (err < 2e-2): bad.append((n, err))
    print("replay", rep, "stats graph", [round(float(v), 4) for v in st_g[:4]], "eager", [round(float(v), 4) for v in st[:4]], "bad:", bad[:8])
