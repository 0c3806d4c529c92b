import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcrl_amd
from hcrl_amd import fused
for M, N in ((524288, 1024), (524288, 128), (524288, 256), (131072, 1024), (524288, 4), (524288, 64)):
    x = torch.randn(M, N, device="cuda").bfloat16()
    for name, fn in (("colsum", lambda: fused.colsum(x)), ("torch", lambda: x.sum(0, dtype=torch.float32))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"M={M} N={N} {name:7s} {us:8.1f} us  {M*N*2/us/1e6:6.2f} TB/s")
