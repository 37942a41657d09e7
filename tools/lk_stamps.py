#!/usr/bin/env python3
"""Phase breakdown of lketkf_tile_kernel from in-kernel stamps (diagnostic build):
    MIA_BUILD_FLAGS=-DMIA_LK_STAMPS python tools/lk_stamps.py [G ...]"""
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
names = ["header + slot table", "record image -> LDS", "pair statistic + exp", "row sums, bound, degree", "recurrence (row 0)", "output + flags"]
for G in [int(a) for a in sys.argv[1:]] or [16, 100000]:
    X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev, seed=43)
    tiles = eng.localize_tiles(gx, ox, [10.0], 20)
    for _ in range(3):
        eng.analysis_tiles_rbf(X, Yb, d, tiles, 1.1, 0.5)
    torch.cuda.synchronize()
    nt = min((G + 15) // 16, 8192)
    buf = np.zeros((nt, 12), dtype=np.int64)
    assert lib.mia_debug_lk_stamps(buf.ctypes.data_as(C.c_void_p), nt) == 0
    dt = np.diff(buf[:, :7], axis=1).astype(np.float64)
    print("G = %d: %d tiles; wave lifetime median %.0f cycles, p90 %.0f" % (G, nt, np.median(buf[:, 6] - buf[:, 0]), np.percentile(buf[:, 6] - buf[:, 0], 90)))
    for i, n in enumerate(names):
        print("  %-28s median %8.0f   p90 %8.0f" % (n, np.median(dt[:, i]), np.percentile(dt[:, i], 90)))
    t0, t1 = buf[:, 10], buf[:, 11]
    print("  kernel span: %.2f us; wave life in real time: median %.2f us" % ((t1.max() - t0.min()) / 100.0, np.median(t1 - t0) / 100.0))
