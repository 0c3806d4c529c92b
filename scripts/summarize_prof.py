#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof/...) into the small, committed summaries under profiles/.

usage: summarize_prof.py <prof_dir> <tag>     ->  profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

prof, tag = sys.argv[1], sys.argv[2]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(repo, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:90]


# ---- kernel-trace --stats ---------------------------------------------------------------------------------
def newest_per_dir(paths):
    """gpurun merges every call's files into the same local directories: keep the most recent file of each directory."""
    best = {}
    for q in paths:
        d = os.path.dirname(q)
        if d not in best or os.path.getmtime(q) > os.path.getmtime(best[d]):
            best[d] = q
    return sorted(best.values())


stats = newest_per_dir(glob.glob(os.path.join(prof, "trace", "**", "*_kernel_stats.csv"), recursive=True))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:12]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"], r["StdDev"]])
    print("kernel stats ->", f"{tag}_kernel_stats.csv")

# ---- PMC passes: average per dispatch of the hot kernels ------------------------------------------------------
pmc = defaultdict(lambda: defaultdict(list))
meta = {}
for path in newest_per_dir(glob.glob(os.path.join(prof, "p*", "**", "*_counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not any(s in k for s in ("rate_env", "sixdof", "cascade", "lstm", "gate", "policy_fe", "policy_trunk")):
            continue
        pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "lds": int(r["LDS_Block_Size"]),
                   "scratch": int(r["Scratch_Size"]), "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]),
                   "sgpr": int(r["SGPR_Count"])}
summary = {}
for k, ctrs in pmc.items():
    s = {c: sum(v) / len(v) for c, v in ctrs.items()}
    s["dispatches_sampled"] = max(len(v) for v in ctrs.values())
    s.update(meta[k])
    if "FETCH_SIZE" in s or "WRITE_SIZE" in s:
        f, w = s.get("FETCH_SIZE", 0.0), s.get("WRITE_SIZE", 0.0)
        # rocprofv3 units are KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts 64 B per 128-B request for
        # wide coalesced streams (x2 correction); our loads are 4/8 B per lane = 256/512 B per wave-instruction, an
        # uncalibrated width, so both the raw and the x2-corrected figure are kept.
        s["hbm_bytes_raw"] = (f + w) * 1024.0
        s["hbm_bytes_fetch_x2"] = (2.0 * f + w) * 1024.0
    summary[k] = s
json.dump(summary, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
