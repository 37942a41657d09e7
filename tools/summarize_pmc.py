#!/usr/bin/env python3
"""Per-launch PMC figures of the dominant analysis kernel from the rocprofv3 --pmc passes of collect_profiles.sh."""
import csv, glob, json, sys
from collections import defaultdict
d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "letkf_cheb_kernel"
vals = defaultdict(list)
recorded = defaultdict(int)          # the kernel's name as rocprofv3 records it (template arguments included)
for f in glob.glob(d + "/pmc*/**/*counter_collection.csv", recursive=True):
    per = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            recorded[r["Kernel_Name"]] += 1
    for disp in per.values():
        for name, v in disp.items():
            vals[name].append(v)
# (median over the launches: the first launch of a kernel that needs scratch memory includes its allocation)
c = {k: sorted(v)[len(v) // 2] for k, v in vals.items()}
G = 100000
workload = sys.argv[3] if len(sys.argv) > 3 else "C2: 1e5 grid points, k=40, <=20 local obs, m=1 (tools/prof_kernel.py --reps 3)"
G = int(sys.argv[4]) if len(sys.argv) > 4 else G
# the kernel-trace pass of the same directory: average duration of that kernel (launches inside bench.py's loop and alone)
trace = {}
for f in glob.glob(d + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Name"]:
            trace = {"name": r["Name"], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                     "max_ns": float(r["MaxNs"])}
            break
out = {"kernel": kern, "kernel_name_recorded": (max(recorded, key=recorded.get) if recorded else trace.get("name")),
       "kernel_trace": trace or None, "grid_points": G, "workload": workload,
       "command": "rocprofv3 --pmc <set> --output-format csv -- python3 tools/prof_kernel.py (one pass per counter set; "
                  "FETCH_SIZE and WRITE_SIZE in their own passes)",
       "counters_per_launch": c, "derived": {}}
dv = out["derived"]
if "SQ_INSTS_VALU" in c:
    dv["valu_instr_per_analysis"] = c["SQ_INSTS_VALU"] / G
    dv["lds_instr_per_analysis"] = c.get("SQ_INSTS_LDS", 0) / G
    dv["salu_instr_per_analysis"] = c.get("SQ_INSTS_SALU", 0) / G
    dv["mfma_instr_per_analysis"] = c.get("SQ_INSTS_VALU_MFMA_F32", 0) / G
    if "SQ_INSTS_VALU_MFMA_F16" in c:
        dv["mfma_f16_instr_per_analysis"] = c["SQ_INSTS_VALU_MFMA_F16"] / G
if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0          # counter is summed over the 8 XCDs
    dv["kernel_cycles"] = cyc
    dv["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] / (cyc * 256)     # per SIMD-issue slot: 256 CUs (x4 SIMDs / 4 cycles)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        dv["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
if "SQ_WAVE_CYCLES" in c:
    for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if name in c:
            dv[name.lower() + "_over_wave_cycles"] = c[name] / c["SQ_WAVE_CYCLES"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    dv["hbm_bytes_raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    dv["hbm_bytes_fetch_doubled"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
print(json.dumps(out, indent=1))
