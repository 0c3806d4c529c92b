"""Where does a wave of the fused env step spend its time?  Needs scratch/libfdyn_stamps.so (the library built with
-DFD_PHASE_STAMPS) loaded through FDYN_LIB; prints per-phase shader-clock cycles over the 256 workgroups' first waves."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcrl_amd import _lib                      # noqa: E402
from hcrl_amd.rate_env import GpuRateVecEnv    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = GpuRateVecEnv(n, "medium", 10.0, 0.02, "step", seed=0, precision="mixed", sampling="device")
env.reset()
dev = env.device
g = torch.Generator(device=dev).manual_seed(1)
mode = sys.argv[3] if len(sys.argv) > 3 else "bench"
if mode == "normal":          # what an untrained Gaussian policy emits: N(0, 1) per channel (the env clips)
    acts = [torch.randn((n, 4), device=dev, generator=g).contiguous() for _ in range(4)]
else:
    acts = [torch.cat([(torch.rand((n, 3), device=dev, generator=g) - 0.5) * 0.6, 0.4 + 0.4 * torch.rand((n, 1), device=dev, generator=g)], 1).contiguous()
            for _ in range(4)]
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for k in range(warm):
    env.step_device(acts[k % 4])
torch.cuda.synchronize()
lib = _lib.load()
fn = C.CDLL(_lib.LIB_PATH).fdyn_debug_read_stamps
fn.argtypes, fn.restype = [C.c_void_p, C.c_int], C.c_int
nb = min(4096, (n + 255) // 256)
fc = C.CDLL(_lib.LIB_PATH).fdyn_debug_read_counts
fc.argtypes, fc.restype = [C.c_void_p, C.c_int, C.c_int], C.c_int
cbuf = np.zeros(nb * 8, dtype=np.uint32)
fc(cbuf.ctypes.data, nb * 8, 1)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
rows = []
for rep in range(5):
    ev0.record(); env.step_device(acts[rep % 4]); ev1.record()
    torch.cuda.synchronize()
    buf = np.zeros(nb * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, nb * 8) == 0
    s = buf.reshape(nb, 8).astype(np.int64)
    d = np.diff(s[:, :6], axis=1)                    # phases 0->1 ... 4->5 (shader clock)
    tot = s[:, 5] - s[:, 0]
    real = (s[:, 6] - s[:, 7]) * 10.0                # ns per wave (100 MHz counter)
    span_ns = (s[:, 6].max() - s[:, 7].min()) * 10.0
    print(f"wave totals: cycles p50 {int(np.percentile(tot, 50))} max {int(tot.max())}; ns p50 {np.percentile(real, 50):.0f} max {real.max():.0f}; "
          f"shader clock {np.median(tot / np.maximum(real, 1)):.3f} GHz; first start -> last end {span_ns:.0f} ns; "
          f"start spread {(s[:, 7].max() - s[:, 7].min()) * 10} ns")
    d = np.concatenate([d, np.zeros((d.shape[0], 1), dtype=d.dtype)], 1)
    span = 0
    assert fc(cbuf.ctypes.data, nb * 8, 1) == 0
    cnt = cbuf.reshape(nb, 8).astype(np.int64)
    rk = d[:, 1]
    order = np.argsort(rk)
    print("rk4 cycles percentiles (min, p10, p50, p90, max):", [int(np.percentile(rk, q)) for q in (0, 10, 50, 90, 100)])
    print("  per wave of 80 dynamics calls / 20 steps: rare-block wave entries, lane entries, fix wave entries, lane entries, trig rebuild lanes, lanes over max_vel, over max_rate, below ground")
    for q in (0, 10, 50, 90, 100):
        w = order[min(len(order) - 1, int(q / 100 * (len(order) - 1)))]
        print(f"    p{q:<3d} wave: rk4 {rk[w]:7d} cycles  counts {cnt[w, :8].tolist()}")
    print("  mean counts:", cnt[:, :8].mean(0).round(1).tolist(), " corr(rk4, rare entries) =", round(float(np.corrcoef(rk, cnt[:, 0])[0, 1]), 3),
          " corr(rk4, fix entries) =", round(float(np.corrcoef(rk, cnt[:, 2])[0, 1]), 3))
    rows.append((ev0.elapsed_time(ev1) * 1e3, span, d.mean(0), d.max(0), 0, 0))
names = ["load+stage->ready", "rk4 x20", "cmd/reward/obs + stores issued", "compaction/auto-reset/state stores", "obs tile", "store drain"]
for us, span, mean, mx, dstart, dend in rows:
    print(f"launch {us:7.1f} us (event)  first-start..last-end {span} cycles; wave start spread {dstart}, end spread {dend}")
    for nm, a, b in zip(names, mean, mx):
        print(f"    {nm:42s} mean {a:9.0f}  max {b:9.0f} cycles")
