"""VALU instructions of the env step's RK4 sub-step loop (main path), from the ISA of csrc/fdyn_kernels.hip -- at one wave per
SIMD a launch's time is proportional to this count.  usage: python scratch/count_hot.py [kernel-substring]"""
import collections, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd", "csrc", "fdyn_kernels.hip")
asm = "/tmp/isa/fdyn_kernels.s"
os.makedirs("/tmp/isa", exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", asm],
                      stderr=subprocess.DEVNULL)
want = sys.argv[1] if len(sys.argv) > 1 else "rate_env_step_kernelIdfLb0"
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
# the sub-step loop: the loop header whose fall-through run up to its s_cbranch_scc is the longest
best = None
for i, l in enumerate(body):
    if "Loop Header" in l:
        j = i + 1
        while j < len(body) and not re.match(r"\s*s_cbranch_scc", body[j]):
            j += 1
        if best is None or j - i > best[1] - best[0]:
            best = (i, j)
i, j = best
ops = [l.split()[0] for l in body[i:j] if l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")]
c = collections.Counter(ops)
valu = sum(n for k, n in c.items() if k.startswith("v_"))
print(f"{want}: loop lines {i}..{j}, VALU {valu}, scalar {sum(n for k, n in c.items() if k.startswith('s_'))}")
print("  pk:", {k: n for k, n in c.items() if k.startswith("v_pk")})
print("  top:", c.most_common(12))
# every instantiation's private segment and register count, straight from the kernel descriptors: a change made for the benched
# build must not leave scratch behind in another one (it did once: a dead 20-byte segment in the f64 env kernel)
name = None
for l in lines:
    m = re.match(r"\s*\.amdhsa_kernel (\S+)", l)
    if m:
        name = m.group(1)
    elif name and ".amdhsa_private_segment_fixed_size" in l:
        priv = int(l.split()[-1])
    elif name and ".amdhsa_next_free_vgpr" in l:
        short = re.sub(r"E+v?i?PT_.*", "", name)
        if any(k in short for k in ("rate_env_step", "sixdof_step", "cascade_step", "agent_step")):
            print(f"  {short:42s} private segment {priv:4d} B   vgpr {int(l.split()[-1])}")
        name = None
