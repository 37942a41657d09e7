for nw in 2 3; do echo "== NW $nw"; MIA_T2P_NW=$nw MIA_BUILD_FLAGS="-DMIA_EXPERIMENTS" timeout -k 10 400 python tools/pair_ab.py 2>&1 | grep -v amdgpu | head -2; done
