#!/bin/bash
# A/B of the tile2 kernel's occupancy target (experiment build): gpurun -- 'bash tools/ab_waves.sh'
export MIA_BUILD_FLAGS=-DMIA_EXPERIMENTS
python -c "import torch_assimilate_amd as m; m.build()" 2>&1 | tail -2
for w in 4 5 6 3; do echo "== waves $w"; MIA_TILE2_WAVES=$w python tools/time_tile2.py c2 2>/dev/null | grep "analysis_tiles"; done
