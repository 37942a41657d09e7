#!/bin/bash
# A/B of build variants of the tile kernel: kernel time vs grid size per variant
for flags in "" "-DMIA_TILE_STAGGER=1" "-DMIA_TILE_STAGGER=2" "-DMIA_TILE_STAGGER=4"; do
  echo "== flags: $flags"; MIA_BUILD_FLAGS="$flags" python tools/time_latency.py 2>&1 | tail -3
done
