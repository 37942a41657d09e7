#!/usr/bin/env python3
"""GPU timeline of the pipelined bench loop from a rocprofv3 --kernel-trace CSV directory: for a window of steps in
the middle of the run, every kernel's start / duration / queue, plus per-kernel mean durations under overlap."""
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("mia::", "").replace("void ", "")[:44], r.get("Queue_Id", "?")))
rows.sort()
cheb = [i for i, r in enumerate(rows) if "letkf_cheb_kernel" in r[2] or "letkf_tile_kernel" in r[2] or "letkf_tile2" in r[2]]
mid = cheb[int(sys.argv[2]) if len(sys.argv) > 2 else 60]      # inside the timed (pipelined) loop of the default bench
t0 = rows[mid][0]
print("window of ~3 steps around the middle of the run (t in us relative to an analysis-kernel start)")
import os
for s, e, name, q in rows[mid - 8:mid + int(os.environ.get('TIMELINE_ROWS', '22'))]:
    print("%9.1f  +%7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name))
lo_i, hi_i = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (20, 100)
st = [rows[i][0] for i in cheb[lo_i:hi_i]]
gaps = [(b - a) / 1e3 for a, b in zip(st[:-1], st[1:])]
print("analysis-kernel start-to-start: mean %.1f us, min %.1f, max %.1f" % (sum(gaps) / len(gaps), min(gaps), max(gaps)))
acc = collections.defaultdict(list)
for s, e, name, q in rows[cheb[lo_i]:cheb[hi_i]]:
    acc[name].append((e - s) / 1e3)
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-46s n=%4d mean %8.1f us" % (name, len(v), sum(v) / len(v)))
