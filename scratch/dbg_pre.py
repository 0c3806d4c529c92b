import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcrl_amd.policy import RateLSTMPolicy
torch.manual_seed(0)
for dtype in (torch.bfloat16, None):
    for T, B in ((8, 128), (8, 512), (1, 128), (6, 8192)):
        pol = RateLSTMPolicy(compute_dtype=dtype).cuda()
        obs = torch.randn(T, B, 18, device="cuda"); act = torch.randn(T, B, 4, device="cuda").clamp(-1, 1)
        starts = (torch.rand(T, B, device="cuda") < 0.05).float()
        st = pol.initial_state(B, "cuda"); st = type(st)(*[torch.randn_like(s) * 0.3 for s in st])
        v, lp, ent = pol.evaluate_sequence(obs, act, starts, st)
        (lp.mean() + 0.5 * (v ** 2).mean() - 0.01 * ent).backward()
        bad = [n for n, p in pol.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print(dtype, T, B, "values finite", bool(torch.isfinite(v).all()), "bad grads:", bad[:6])
