#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of the tile route's kernels against the number of grid points:
# flat = latency-bound (one round of resident waves), linear = throughput-bound.   gpurun -- 'bash tools/scale_kernels.sh'
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/scale_kernels; rm -rf $out; mkdir -p $out
for g in 12800 25600 51200 81920 100000 131072 200000 400000; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$g -- python3 tools/time_tile2.py c2 --grid $g --batches 3 > $out/log$g.txt 2>&1
  f=$(find $out/t$g -name '*kernel_stats.csv' | head -1)
  echo "G=$g"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "tile2_kernel" in n or "localize_tiles" in n or "index_" in n or "pack_split" in n:
        print("   %-60s calls %5s avg %8.1f us min %8.1f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  rm -rf $out/t$g
done
