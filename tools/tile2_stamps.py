#!/usr/bin/env python3
"""Phase breakdown and occupancy timeline of letkf_tile2_kernel from in-kernel stamps (diagnostic build):
    MIA_BUILD_FLAGS=-DMIA_TILE_STAMPS python tools/tile2_stamps.py [G ...]"""
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
names = ["header + slot table + tails", "gather / x / D requested", "... landed", "x' split, Gram + Z", "Gershgorin + table header",
         "recurrence", "output products + stores", "flags"]
for G in [int(a) for a in sys.argv[1:]] or [16, 100000]:
    X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev)
    tiles = eng.localize_tiles(gx, ox, [10.0], 20)
    rec = eng.pack_split(Yb, d)
    for _ in range(3):
        eng.analysis_tiles(X, rec, Yb.shape[1], tiles, 1.1)
    torch.cuda.synchronize()
    nt = min((G + 15) // 16, 8192)
    buf = np.zeros((nt, 20), dtype=np.int64)
    assert lib.mia_debug_tile2_stamps(buf.ctypes.data_as(C.c_void_p), nt) == 0
    dt = np.diff(buf[:, :9], axis=1).astype(np.float64)
    print("G = %d: %d tiles; wave lifetime median %.0f cycles, p90 %.0f" % (G, nt, np.median(buf[:, 8] - buf[:, 0]), np.percentile(buf[:, 8] - buf[:, 0], 90)))
    for i, n in enumerate(names):
        print("  %-30s median %8.0f   p90 %8.0f" % (n, np.median(dt[:, i]), np.percentile(dt[:, i], 90)))
    # occupancy timeline from the constant 100 MHz counter (10 ns ticks)
    t0, t1 = buf[:, 10], buf[:, 11]
    lo = t0.min()
    span = t1.max() - lo
    print("  kernel span: %.2f us (first wave start -> last wave end); wave life in real time: median %.2f us" % (span / 100.0, np.median(t1 - t0) / 100.0))
    edges = np.linspace(0, span, 13)
    for a, b in zip(edges[:-1], edges[1:]):
        mid = lo + 0.5 * (a + b)
        print("    t = %5.1f us: %5d waves resident, %5d started so far" % (0.5 * (a + b) / 100.0, int(((t0 <= mid) & (t1 > mid)).sum()), int((t0 <= mid).sum())))
    hw = buf[:, 9]
    simd = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xf) * 16 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 1) * 2048 + ((hw >> 4) & 3)
    u, c = np.unique(simd, return_counts=True)
    print("  distinct (xcc, se, sh, cu, simd) ids seen: %d; tiles per SIMD: min %d median %d max %d" % (len(u), c.min(), np.median(c), c.max()))
