import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcrl_amd.rate_env import GpuRateVecEnv
from hcrl_amd.policy import RateLSTMPolicy
from hcrl_amd.ppo import PPOConfig, RecurrentPPO
torch.manual_seed(0)
for diff, cmd in (("easy", "step"),):
    env = GpuRateVecEnv(512, diff, 10.0, 0.02, cmd, seed=3, precision="mixed", sampling="device")
    pol = RateLSTMPolicy(compute_dtype=torch.bfloat16)
    ppo = RecurrentPPO(env, pol, PPOConfig(n_steps=8, n_epochs=1, n_minibatches=4), seed=1, use_update_graph=(os.environ.get('UG', '1') == '1'))
    for it in range(5):
        ppo.collect_rollout()
        fin = {k: bool(torch.isfinite(getattr(ppo, k)).all()) for k in ("buf_obs", "buf_act", "buf_rew", "buf_val", "buf_logp", "adv", "ret")}
        st = ppo.update()
        bad = [n for n, p in pol.named_parameters() if not torch.isfinite(p).all()]
        print(os.environ.get("TAG"), "iter", it, all(fin.values()), "bad params:", len(bad), round(st["value_loss"], 3))
