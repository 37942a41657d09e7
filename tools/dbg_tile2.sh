for f in "-DT2_DBG_B"; do
echo "=== $f"
MIA_BUILD_FLAGS="$f" python tools/dbg_tile2.py 2>&1 | grep -v amdgpu.ids | tail -3
done
