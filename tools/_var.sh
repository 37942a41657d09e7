for f in "-DLK_NV=18 -DLK_ROT=1" "-DLK_NV=16 -DLK_ROT=1"; do echo "=== $f"; MIA_BUILD_FLAGS="$f" timeout -k 10 300 python tools/lk_tile_check.py 100000 2>&1 | grep -E "oracle|tile route|retry"; done
MIA_BUILD_FLAGS="-DLK_NV=18 -DMIA_LK_STAMPS" timeout -k 10 300 python tools/lk_stamps.py 100000
