#!/usr/bin/env python3
"""Print the GPU timeline of the last step found in a rocprofv3 kernel-trace (+ memory-copy) CSV directory."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), "copy"))
rows.sort()
# last step = from the last pack_obs kernel on
starts = [i for i, r in enumerate(rows) if "pack_obs" in r[2]]
i0 = starts[-1] if starts else 0
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 20000 and "pack_obs" not in rows[i0 - 1][2]:
    i0 -= 1
t0 = rows[i0][0]
for s, e, name, q in rows[i0:]:
    print("%9.1f us  +%7.1f us  %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name))
