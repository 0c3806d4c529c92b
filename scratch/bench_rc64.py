"""Stand-alone timing of csrc/policy_rc64.hip (both recurrent cells, one launch, in place) against two launches of the per-cell
kernel (lstm_mfma64, out of place as the rollout used to run it)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcrl_amd import _lib
from hcrl_amd.policy import pack_rc_weights, rc_pack_c, rc_pack_h, rc_pack_x
lib = _lib.load()
import ctypes
rclib = ctypes.CDLL(os.environ["RC64_LIB"]) if os.environ.get("RC64_LIB") else lib      # experiment builds of policy_rc64.hip alone
if rclib is not lib:
    rclib.fdyn_policy_recurrent.restype, rclib.fdyn_policy_recurrent.argtypes = ctypes.c_int, [ctypes.c_void_p] * 12 + [ctypes.c_int64, ctypes.c_void_p]
B, dev, bf = 65536, "cuda", torch.bfloat16
x = (torch.randn(B, 128, device=dev) * 0.7).to(bf)
keep = (torch.rand(B, device=dev) > 0.01).float()
cells = [((torch.randn(1024, 128, device=dev) * 0.08).to(bf), (torch.randn(1024, 256, device=dev) * 0.08).to(bf), torch.randn(1024, device=dev) * 0.3,
          (torch.randn(B, 256, device=dev) * 0.5).to(bf), torch.randn(B, 256, device=dev)) for _ in range(2)]
img = pack_rc_weights([(c[0], c[1]) for c in cells]); bias = torch.stack([c[2] for c in cells]).contiguous()
xi = rc_pack_x(x); hi = [rc_pack_h(c[3]) for c in cells]; ci = [rc_pack_c(c[4]) for c in cells]
W = [torch.cat([c[0], c[1]], 1).contiguous() for c in cells]
ho = [torch.empty_like(c[3]) for c in cells]; co = [torch.empty_like(c[4]) for c in cells]
st = _lib.current_stream()
def rc():
    rclib.fdyn_policy_recurrent(xi.data_ptr(), keep.data_ptr(), img.data_ptr(), bias.data_ptr(), hi[0].data_ptr(), ci[0].data_ptr(), hi[0].data_ptr(), ci[0].data_ptr(),
                              hi[1].data_ptr(), ci[1].data_ptr(), hi[1].data_ptr(), ci[1].data_ptr(), B, st)
def old():
    for k in range(2):
        lib.fdyn_lstm_cell_mfma(x.data_ptr(), 128, cells[k][3].data_ptr(), 256, cells[k][4].data_ptr(), keep.data_ptr(), W[k].data_ptr(), cells[k][2].data_ptr(),
                                ho[k].data_ptr(), co[k].data_ptr(), None, B, 256, st)
tag = os.environ.get("RC64_TAG", "policy_rc64: both cells, one launch, in place")
for name, fn in ((tag, rc),) + ((("lstm_mfma64 x 2, out of place", old),) if rclib is lib else ()):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{name:50s} {us:8.1f} us   {2 * 2 * B * 384 * 1024 / us / 1e6:8.1f} TFLOP/s   {2 * 218.1 / us * 1e3 / 1e3:6.2f} TB/s of state + activation traffic")

if hasattr(rclib, "fdyn_rc64_read_stamps"):
    import numpy as np
    buf = np.zeros(256 * 64 * 4, np.uint64)
    rc(); torch.cuda.synchronize()
    rclib.fdyn_rc64_read_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    st = buf.reshape(256, 64, 4).astype(np.int64)
    mf, wt, br = st[:, :, 1] - st[:, :, 0], st[:, :, 2] - st[:, :, 1], st[:, :, 3] - st[:, :, 2]
    gap = st[:, 1:, 0] - st[:, :-1, 3]
    print("shader cycles per unit (median over 256 workgroups; wave 0): unit body | end wait (vmcnt) | barrier | to next unit")
    for q, name in enumerate(("i + P_O", "g + P_I", "f + P_G", "o + P_F")):
        sel = np.arange(q, 64, 4)[1:]
        print(f"  {name:8s} {np.median(mf[:, sel]):7.0f} | {np.median(wt[:, sel]):6.0f} | {np.median(br[:, sel]):6.0f} | {np.median(gap[:, sel[:-1]]):5.0f}    (p90 body {np.percentile(mf[:, sel], 90):.0f}, wait {np.percentile(wt[:, sel], 90):.0f}, barrier {np.percentile(br[:, sel], 90):.0f})")
    tot = st[:, -1, 3] - st[:, 0, 0]
    print(f"  whole loop: median {np.median(tot):.0f} cycles, sum of bodies {np.median(mf.sum(1)):.0f}, waits {np.median(wt.sum(1)):.0f}, barriers {np.median(br.sum(1)):.0f}")

if hasattr(rclib, "fdyn_rc64_read_gap_stamps"):
    import numpy as np
    buf = np.zeros(256 * 98, np.uint64)
    rc(); torch.cuda.synchronize()
    rclib.fdyn_rc64_read_gap_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    st = buf.reshape(256, 2, 49).astype(np.int64)
    d = np.median(st[:, :, 1:] - st[:, :, :-1], axis=0)
    for w, name in enumerate(("unit i + P_O", "unit o + P_F")):
        print(name, "cycles per MFMA gap (median over workgroups), gaps 0..47:")
        print("   " + " ".join(f"{int(x):4d}" for x in d[w][:24])); print("   " + " ".join(f"{int(x):4d}" for x in d[w][24:]))
