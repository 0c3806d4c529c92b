"""Instruction mix per basic block of a kernel in a hipcc -S listing.  usage: isa_mix.py file.s kernel-name-substring [min-block-size]"""
import re, sys, collections
L = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 200
start = next(i for i, l in enumerate(L) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
blocks, cur = [], ("entry", [])
for l in L[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = (m.group(1), [])
    else:
        t = l.strip()
        if t and not t.startswith(";") and not t.startswith("."):
            cur[1].append(t.split()[0])
blocks.append(cur)
def cat(op):
    if op.startswith("v_mfma"): return "mfma"
    if op in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32"): return "trans"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"
for name, ops in blocks:
    if len(ops) < minsz: continue
    c = collections.Counter(cat(o) for o in ops)
    print(name, len(ops), dict(c))
    v = collections.Counter(o for o in ops if cat(o) in ("valu", "acc_mov"))
    print("   ", v.most_common(16))
