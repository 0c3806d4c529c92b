# time fdyn_policy_features from a given shared object (ablation builds of csrc/policy_fe64.hip; timing only)
import sys, ctypes, torch
so = sys.argv[1]
lib = ctypes.CDLL(so)
f = lib.fdyn_policy_features
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
B = 65536
obs = torch.randn(B, 18, device="cuda")
img = (torch.randn(lib.fdyn_policy_features_image_bytes() // 2, device="cuda") * 0.05).bfloat16()
bias = torch.randn(2304, device="cuda") * 0.1
feats = torch.empty((B, 128), dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run():
    assert f(obs.data_ptr(), img.data_ptr(), bias.data_ptr(), feats.data_ptr(), B, st) == 0
for _ in range(5): run()
torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"{so.split('/')[-1]:32s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us")
