# time fdyn_lstm_cell_mfma64_try from a given shared object (ablation builds of csrc/lstm_mfma64.hip; timing only)
import sys, ctypes, torch
so = sys.argv[1]
lib = ctypes.CDLL(so)
f = lib.fdyn_lstm_cell_mfma64_try
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
B, H, kx, kh = 65536, 256, 128, 256
x = (torch.randn(B, kx, device="cuda") * 0.7).bfloat16(); h = (torch.randn(B, kh, device="cuda") * 0.5).bfloat16()
c = torch.randn(B, H, device="cuda"); keep = (torch.rand(B, device="cuda") > 0.01).float()
W = (torch.randn(4 * H, kx + kh, device="cuda") * 0.08).bfloat16(); bias = torch.randn(4 * H, device="cuda") * 0.3
h_out = torch.empty((B, H), dtype=torch.bfloat16, device="cuda"); c_out = torch.empty((B, H), device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run():
    rc = f(x.data_ptr(), kx, h.data_ptr(), kh, c.data_ptr(), keep.data_ptr(), W.data_ptr(), bias.data_ptr(), h_out.data_ptr(), c_out.data_ptr(), B, H, st)
    assert rc == 1, rc
for _ in range(5): run()
torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print(f"{so.split('/')[-1]:32s} {ms*1e3:8.1f} us   {2*B*(kx+kh)*4*H/ms/1e9:8.1f} TFLOP/s")
