#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/prof_<tag> (tools/collect_profiles.sh) into profiles/ and refresh
profiles/latest_traffic.json (what bench.py reports as roofline.traffic)."""
import glob, json, os, shutil, sys
tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles", exist_ok=True)
stats = max(glob.glob(src + "/trace/*/*kernel_stats.csv"), key=os.path.getmtime)      # (the newest run of this tag)
shutil.copy(stats, "profiles/%s_kernel_stats.csv" % tag)
line = [l for l in open(src + "/bench.log") if l.startswith('{"metric"')][-1] if os.path.exists(src + "/bench.log") else open(src + "/bench.json").read()
open("profiles/%s_bench.json" % tag, "w").write(line)
pmc = json.load(open(src + "/pmc_summary.json"))
json.dump(pmc, open("profiles/%s_pmc.json" % tag, "w"), indent=1)
c = pmc["counters_per_launch"]
rec = {"grid_points": 100000, "k": 40, "kernel": pmc.get("kernel", "letkf_tile_kernel"),
       "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
       "hbm_bytes_raw": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
       "hbm_bytes_fetch_doubled": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
       "algorithmic_bytes": 41400000, "source": "profiles/%s_pmc.json" % tag,
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_tile.sh) on tools/prof_kernel.py; "
               "raw = (FETCH+WRITE)*1024; MI355X_MICROARCH.md: FETCH_SIZE counts 1/2 of wide coalesced streaming reads on "
               "gfx950, so the true read side lies between raw and doubled; WRITE_SIZE equals the 16 MB analysis ensemble"}
json.dump(rec, open("profiles/latest_traffic.json", "w"), indent=1)
print("published", tag)
