import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcrl_amd import _lib
from hcrl_amd.policy import pack_rc_weights, rc_pack_c, rc_pack_h, rc_pack_x, rc_unpack_c, rc_unpack_h
lib = _lib.load()
B, dev, bf = 256, "cuda", torch.bfloat16
torch.manual_seed(0)
x = (torch.randn(B, 128, device=dev) * 0.7).to(bf)
keep = (torch.rand(B, device=dev) > 0.05).float()
cells = [((torch.randn(1024, 128, device=dev) * 0.08).to(bf), (torch.randn(1024, 256, device=dev) * 0.08).to(bf), torch.randn(1024, device=dev) * 0.3,
          (torch.randn(B, 256, device=dev) * 0.5).to(bf), torch.randn(B, 256, device=dev)) for _ in range(2)]
img = pack_rc_weights([(c[0], c[1]) for c in cells]); bias = torch.stack([c[2] for c in cells]).contiguous()
xi = rc_pack_x(x); hi = [rc_pack_h(c[3]) for c in cells]; ci = [rc_pack_c(c[4]) for c in cells]
for trial in range(3):
    ho = [torch.zeros_like(t) for t in hi]; co = [torch.zeros_like(t) for t in ci]
    lib.fdyn_policy_recurrent(xi.data_ptr(), keep.data_ptr(), img.data_ptr(), bias.data_ptr(), hi[0].data_ptr(), ci[0].data_ptr(), ho[0].data_ptr(), co[0].data_ptr(),
                              hi[1].data_ptr(), ci[1].data_ptr(), ho[1].data_ptr(), co[1].data_ptr(), B, _lib.current_stream())
    torch.cuda.synchronize()
    for k in range(2):
        gates = torch.cat([x.float(), cells[k][3].float() * keep[:, None]], 1) @ torch.cat([cells[k][0], cells[k][1]], 1).float().t() + cells[k][2]
        i, f, gg, o = gates.chunk(4, 1)
        c2 = torch.sigmoid(f) * (cells[k][4] * keep[:, None]) + torch.sigmoid(i) * torch.tanh(gg)
        h2 = torch.sigmoid(o) * torch.tanh(c2)
        hg, cg = rc_unpack_h(ho[k]).float(), rc_unpack_c(co[k])
        badc = ~torch.isfinite(cg) | ((cg - c2).abs() > 1e-3); badh = ~torch.isfinite(hg) | ((hg - h2).abs() > 2e-2)
        print(f"trial {trial} cell {k}: bad c {int(badc.sum())} bad h {int(badh.sum())} of {B*256}; nonfinite c {int((~torch.isfinite(cg)).sum())} h {int((~torch.isfinite(hg)).sum())}")
        if badc.any():
            rows = badc.any(1).nonzero().flatten().tolist(); cols = badc.any(0).nonzero().flatten().tolist()
            print("   c bad rows", rows[:40], "n", len(rows)); print("   c bad units: slices", sorted(set(c // 32 for c in cols)), "n units", len(cols), "first", cols[:40])
        if badh.any():
            rows = badh.any(1).nonzero().flatten().tolist(); cols = badh.any(0).nonzero().flatten().tolist()
            print("   h bad rows", rows[:40], "n", len(rows)); print("   h bad units: slices", sorted(set(c // 32 for c in cols)), "n units", len(cols), "first", cols[:40])
