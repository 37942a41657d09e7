timeout -k 10 300 python tools/lk_tile_check.py 100000 2>&1 | grep -E "oracle|tile route|retry|Error|error"
MIA_BUILD_FLAGS="-DMIA_LK_STAMPS" timeout -k 10 300 python tools/lk_stamps.py 100000
