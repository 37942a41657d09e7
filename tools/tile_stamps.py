#!/usr/bin/env python3
"""Phase breakdown of letkf_tile_kernel from in-kernel s_memtime stamps (diagnostic build: MIA_BUILD_FLAGS=-DMIA_TILE_STAMPS).
    MIA_BUILD_FLAGS=-DMIA_TILE_STAMPS python tools/tile_stamps.py [G ...]"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
lib = C.CDLL(mia.LIB_PATH)
names = ["lists req", "lists+X req -> union", "gather", "convert + D scatter + dreg", "X wait + Gram + Z", "Gershgorin + table + coef", "recurrence", "zu + output", "flags"]
from torch_assimilate_amd import _cabi
SPLIT = int(os.environ.get("TILE_SPLIT", "1"))
_cabi.set_option("tile_split", SPLIT)
fetch = lib.mia_debug_tile_split_stamps if SPLIT else lib.mia_debug_tile_stamps
for G in [int(a) for a in sys.argv[1:]] or [16, 100000]:
    X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev)
    nb = eng.localize(gx, ox, [10.0])
    rec = eng.pack_obs(Yb, d, torch.float32)
    for _ in range(3):
        eng.analysis(X, None, None, nb, 1.1, rec=rec, method="matfun", defer_retry=True)
    torch.cuda.synchronize()
    nt = min((G + 15) // 16, 8192)
    buf = np.zeros((nt, 12), dtype=np.int64)
    assert fetch(buf.ctypes.data_as(C.c_void_p), nt) == 0
    dt = np.diff(buf[:, :9], axis=1).astype(np.float64)
    os.makedirs('gpurun_out', exist_ok=True)
    np.save('gpurun_out/tile_stamps_%d.npy' % G, buf)
    print("G = %d: %d tiles; per-tile wave lifetime median %.0f cycles (s_memtime ticks = shader cycles? see guide)" % (G, nt, np.median(buf[:, 8] - buf[:, 0])))
    for i, n in enumerate(names[:8]):
        print("  %-28s median %8.0f   p90 %8.0f" % (n, np.median(dt[:, i]), np.percentile(dt[:, i], 90)))
    print("  kernel span (first start -> last end): %.0f ticks" % (buf[:, 8].max() - buf[:, 0].min()))
