"""Python view of include/fdyn_layout.h (the single source of truth for every flat-array slot).

The header is parsed at import so the host mirror, the C-ABI and the oracle can never drift apart.
"""
import os
import re

_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fdyn_layout.h")


def _parse(path):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", src, flags=re.S):
        nxt = 0
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = [t.strip() for t in item.split("=")]
                val = eval(val, {}, out)  # only earlier enumerators / integer literals
            else:
                name, val = item, nxt
            out[name] = int(val)
            nxt = int(val) + 1
    for name, val in re.findall(r"#define\s+(FD_\w+)\s+\(?([^\n]+?)\)?\s*$", src, flags=re.M):
        out[name] = int(eval(val, {}, out))
    return out


globals().update(_parse(_HDR))
__all__ = [k for k in globals() if k.startswith("FD_")]
