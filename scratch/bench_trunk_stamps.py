"""Phase stamps of the trunks + heads kernel (experiment build scratch/ubench/policy_trunk_stamps_experiment.hip.txt)."""
import ctypes, os, sys, numpy as np, torch
lib = ctypes.CDLL(sys.argv[1])
f = lib.fdyn_policy_trunks_heads
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] * 11 + [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
B, dev, bf = 65536, "cuda", torch.bfloat16
h = [(torch.randn(B, 256, device=dev) * 0.5).to(bf) for _ in range(2)]
w1 = (torch.randn(2, 128, 256, device=dev) * 0.08).to(bf); b1 = torch.randn(2, 128, device=dev) * 0.1
w2 = (torch.randn(2, 64, 128, device=dev) * 0.1).to(bf); b2 = torch.randn(2, 64, device=dev) * 0.1
wa = (torch.randn(4, 64, device=dev) * 0.1).to(bf); ba = torch.zeros(4, device=dev).to(bf); wv = (torch.randn(64, device=dev) * 0.1).to(bf); bv = torch.zeros(1, device=dev).to(bf)
ls = torch.zeros(4, device=dev); step = torch.zeros(1, dtype=torch.int32, device=dev)
act = torch.empty(B, 4, device=dev); lp = torch.empty(B, device=dev); val = torch.empty(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    assert f(h[0].data_ptr(), h[1].data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), wa.data_ptr(), ba.data_ptr(), wv.data_ptr(), bv.data_ptr(),
             ls.data_ptr(), 1234, step.data_ptr(), 0, act.data_ptr(), lp.data_ptr(), val.data_ptr(), B, st) == 0
for _ in range(5): run()
torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"trunks + heads: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
buf = np.zeros(256 * 8 * 16, np.uint64)
lib.fdyn_trunk_read_stamps(buf.ctypes.data_as(ctypes.c_void_p))
s = buf.reshape(2, 128, 8, 16).astype(np.int64)
names = ["start", "staging loads + row requests issued", "staging data arrived, LDS writes issued", "LDS writes done", "barrier passed"] + [f"tile {i}: {n}" for i in range(2) for n in ("rows in fragments", "layer 1 pass 0 done", "layer 1 pass 1 done", "layer 2 done", "heads done")]
for trunk in range(2):
    for grp, ws in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        x = s[trunk, :, ws, :].reshape(-1, 16)
        rel = np.median(x - x[:, 0:1], axis=0)
        print(f"trunk {trunk} ({'pi' if trunk == 0 else 'vf'}), {grp}: " + "  ".join(f"{rel[k]:.0f}" for k in (0, 7, 13, 14, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12)) + f"  end {rel[15]:.0f}")
print("columns: " + " | ".join(names))
t0 = s[:, :, :, 0].min(); print(f"latest end over the launch: {(s[:, :, :, 15].max() - t0)} cycles; workgroup start spread p90: {np.percentile(s[:, :, 0, 0] - t0, 90):.0f}")
