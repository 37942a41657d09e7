#!/bin/bash
# A/B of a start-up stagger of the tile2 kernel's waves (experiment build): gpurun -- 'bash tools/ab_waves.sh'
export MIA_BUILD_FLAGS=-DMIA_EXPERIMENTS
python -c "import torch_assimilate_amd as m; m.build()" 2>&1 | tail -1
for w in 0 4 8 16 32 64; do echo "== stagger $w x 64 cycles per wave slot"; MIA_TILE2_STAGGER=$w python tools/time_tile2.py c2 2>/dev/null | grep "analysis_tiles"; done
