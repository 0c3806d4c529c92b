import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcrl_amd
from hcrl_amd import _lib
lib = _lib.load()
H, B, T = 256, 32768, 16          # T distinct buffer sets (~4.7 GB): nothing stays in the 256 MB infinity cache
dt = torch.bfloat16
gates = torch.randn(T, B, 4 * H, device="cuda").to(dt); c_all = torch.randn(T + 1, B, H, device="cuda"); keep = torch.ones(B, device="cuda")
h = torch.empty(T, B, H, device="cuda", dtype=dt); act = torch.empty_like(gates)
xall = torch.empty(T + 1, B, 384, device="cuda", dtype=dt)
dh = torch.randn(T, B, H, device="cuda").to(dt); dcat = torch.randn(T + 1, B, 384, device="cuda").to(dt); dc = [torch.randn(B, H, device="cuda") for _ in range(2)]
st = _lib.current_stream()
def fwd():
    for t in range(T):
        lib.fdyn_lstm_seq_fwd(gates[t].data_ptr(), 1, c_all[t].data_ptr(), keep.data_ptr(), h[t].data_ptr(), c_all[t + 1].data_ptr(), act[t].data_ptr(),
                              xall[t + 1].data_ptr() + 128 * 2, 384, keep.data_ptr(), None, 1, B, H, st)
def bwd():
    for t in range(T - 1, -1, -1):
        lib.fdyn_lstm_seq_bwd(act[t].data_ptr(), 1, c_all[t].data_ptr(), keep.data_ptr(), c_all[t + 1].data_ptr(), dh[t].data_ptr(), dcat[t + 1].data_ptr() + 128 * 2, 384,
                              keep.data_ptr(), dc[(t + 1) & 1].data_ptr(), act[t].data_ptr(), dc[t & 1].data_ptr(), B, H, st)
src = torch.empty(T, B, 4 * H, device="cuda", dtype=dt); dst = torch.empty_like(src)
def copy():
    for t in range(T):
        dst[t].copy_(src[t])
fb = B * H * (4 * 2 + 4 + 2 + 4 + 8 + 2); bb = B * H * (8 + 4 + 4 + 2 + 2 + 4 + 8 + 4); cb = B * 4 * H * 2 * 2
for name, fn, nbytes in (("fwd", fwd, fb), ("bwd", bwd, bb), ("copy 67MB", copy, cb)):
    fn(); fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (5 * T) * 1e3
    print(f"{name}: {us:7.1f} us per step  {nbytes/us/1e6:5.2f} TB/s")
