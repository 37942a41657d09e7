#!/bin/bash
# A/B of build variants of the tile kernel: kernel time at C2 / C4 and the pipelined bench per variant
for flags in "$@"; do
  echo "== flags: [$flags]"
  MIA_BUILD_FLAGS="$flags" python tools/time_kernel.py c2 --methods matfun 2>&1 | tail -1
  bash tools/ab_bench.sh "$flags"
done
