import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcrl_amd
from hcrl_amd import _lib
from hcrl_amd.fused import lstm_cell
lib = _lib.load()
B, H, kx, kh = 65536, 256, 128, 256
x = (torch.randn(B, kx, device="cuda") * 0.7).bfloat16(); h = (torch.randn(B, kh, device="cuda") * 0.5).bfloat16()
c = torch.randn(B, H, device="cuda"); keep = (torch.rand(B, device="cuda") > 0.01).float()
W = (torch.randn(4 * H, kx + kh, device="cuda") * 0.08).bfloat16(); bias = torch.randn(4 * H, device="cuda") * 0.3
h_out = torch.empty((B, H), dtype=torch.bfloat16, device="cuda"); c_out = torch.empty((B, H), device="cuda")
def mfma():
    lib.fdyn_lstm_cell_mfma(x.data_ptr(), kx, h.data_ptr(), kh, c.data_ptr(), keep.data_ptr(), W.data_ptr(), bias.data_ptr(), h_out.data_ptr(), c_out.data_ptr(), None, B, H, _lib.current_stream())
def unfused():
    g = torch.nn.functional.linear(torch.cat([x, h * keep[:, None].bfloat16()], 1), W, bias.bfloat16())
    return lstm_cell(g, c * keep[:, None])
def gemm_only():
    return torch.nn.functional.linear(torch.cat([x, h], 1), W, bias.bfloat16())
for name, fn in (("mfma fused cell", mfma), ("hipBLASLt GEMM + fused pointwise", unfused), ("hipBLASLt GEMM only (with cat)", gemm_only)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{name:40s} {ms*1e3:8.1f} us   {2*B*(kx+kh)*4*H/ms/1e9:8.1f} TFLOP/s")
