"""Where policy_fe64 waits: per-wave totals of the time in front of the per-unit waits (experiment build policy_fe64_stamps_experiment)."""
import sys, ctypes, numpy as np, torch
lib = ctypes.CDLL(sys.argv[1])
f = lib.fdyn_policy_features
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
B = 65536
obs = torch.randn(B, 18, device="cuda")
img = (torch.randn(lib.fdyn_policy_features_image_bytes() // 2, device="cuda") * 0.05).bfloat16()
bias = torch.randn(2304, device="cuda") * 0.1
feats = torch.empty((B, 128), dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(5): assert f(obs.data_ptr(), img.data_ptr(), bias.data_ptr(), feats.data_ptr(), B, st) == 0
torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
e0.record()
for _ in range(50): f(obs.data_ptr(), img.data_ptr(), bias.data_ptr(), feats.data_ptr(), B, st)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch")
buf = np.zeros(1024 * 8, np.uint64)
lib.fdyn_fe_read_stamps(buf.ctypes.data_as(ctypes.c_void_p))
s = buf.reshape(1024, 8).astype(np.float64)
names = ["whole kernel", "in s_waitcnt vmcnt(0) (32 unit ends)", "in the barriers behind them", "prologue + embedding", "layer 1", "layer 2", "projection + store"]
for k, n in enumerate(names): print(f"  {n:40s} median {np.median(s[:, k]):8.0f}   p10 {np.percentile(s[:, k], 10):8.0f}   p90 {np.percentile(s[:, k], 90):8.0f} cycles")
