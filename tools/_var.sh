MIA_BUILD_FLAGS="-DMIA_T2P_STAMPS" timeout -k 10 300 python tools/t2p_stamps.py 100000 2>&1 | tail -19
