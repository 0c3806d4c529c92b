import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcrl_amd
from hcrl_amd.rate_env import GpuRateVecEnv
from hcrl_amd.policy import RateLSTMPolicy, RNNStates
from hcrl_amd.ppo import RecurrentPPO, PPOConfig
N, T = 16384, 64
env = GpuRateVecEnv(N, "easy", 10.0, 0.02, "step", seed=42, precision="mixed", sampling="device")
m = RecurrentPPO(env, RateLSTMPolicy(compute_dtype=torch.bfloat16), PPOConfig(n_steps=T, n_epochs=4, n_minibatches=8, learning_rate=1e-3, reward_scale=0.02), seed=42, use_graph=False, use_update_graph=True)
m.collect_rollout()
mb = N // 8
ug = m._build_update_graph(mb)
names = [n for n, p in m.policy.named_parameters()]
for trial in range(6):
    idx = torch.randperm(N, device="cuda")[:mb]
    for k, src in (("obs", m.buf_obs), ("act", m.buf_act), ("starts", m.buf_start), ("adv", m.adv), ("ret", m.ret), ("old_logp", m.buf_logp), ("old_v", m.buf_val)):
        torch.index_select(src, 1, idx, out=ug[k])
    for dst, src in zip(ug["states"], m.rollout_states):
        torch.index_select(src, 0, idx, out=dst)
    ug["graph"].replay()
    torch.cuda.synchronize()
    g_flat, g_st = m.flat.buf.clone(), ug["stats"].clone()
    m.flat.zero()
    loss, st = m._minibatch_loss(ug["obs"], ug["act"], ug["starts"], ug["adv"], ug["ret"], ug["old_logp"], ug["old_v"], ug["states"])
    loss.backward()
    torch.cuda.synchronize()
    e_flat = m.flat.buf.clone()
    print(f"trial {trial}: stats graph {g_st.tolist()} eager {st.tolist()}  |g| graph {float(g_flat.norm()):.4g} eager {float(e_flat.norm()):.4g}", flush=True)
    off = 0
    for n, p in zip(names, m.flat.params):
        a, b = g_flat[off:off + p.numel()], e_flat[off:off + p.numel()]
        off += p.numel()
        err = float((a - b).norm() / (b.norm() + 1e-12))
        if err > 0.05 or not bool(torch.isfinite(a).all()):
            print(f"    {n}: rel err {err:.3g} |graph| {float(a.norm()):.3g} |eager| {float(b.norm()):.3g}")
