#!/bin/bash
export MIA_BUILD_FLAGS=-DMIA_TILE_STAMPS
python tools/tile2_stamps.py "$@"
