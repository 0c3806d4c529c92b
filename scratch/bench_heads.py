# time fdyn_policy_heads from a given shared object and print a checksum of its outputs (A/B of csrc/policy_kernels.hip builds)
import sys, ctypes, torch
so = sys.argv[1]
lib = ctypes.CDLL(so)
f = lib.fdyn_policy_heads
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p]
torch.manual_seed(0)
for B in (65536, 16384, 1000):
    pi = torch.randn(B, 64, device="cuda").bfloat16(); vf = torch.randn(B, 64, device="cuda").bfloat16()
    Wa = (torch.randn(4, 64, device="cuda") * 0.1).bfloat16(); ba = torch.randn(4, device="cuda").bfloat16()
    wv = (torch.randn(64, device="cuda") * 0.1).bfloat16(); bv = torch.randn(1, device="cuda").bfloat16()
    ls = torch.zeros(4, device="cuda") - 0.5
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    a = torch.empty((B, 4), device="cuda"); lp = torch.empty(B, device="cuda"); v = torch.empty(B, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def run():
        assert f(pi.data_ptr(), vf.data_ptr(), Wa.data_ptr(), ba.data_ptr(), wv.data_ptr(), bv.data_ptr(), ls.data_ptr(), 7, step.data_ptr(), 0,
                 a.data_ptr(), lp.data_ptr(), v.data_ptr(), B, st) == 0
    for _ in range(5): run()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{so.split('/')[-1]:28s} B={B:6d} {e0.elapsed_time(e1) / 100 * 1e3:7.2f} us  checksum {a.double().sum().item():.10e} {lp.double().sum().item():.10e} {v.double().sum().item():.10e}")
