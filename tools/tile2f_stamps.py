#!/usr/bin/env python3
"""Phase breakdown and occupancy timeline of the FUSED kernel (letkf_tile2f_kernel: every wavefront localises its tile, then analyses
it) from in-kernel stamps, through the step driver (diagnostic build):
    MIA_BUILD_FLAGS=-DMIA_TILE_STAMPS python tools/tile2f_stamps.py [G ...]"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
mia.build()
dev = torch.device("cuda:0")
lib = C.CDLL(mia.LIB_PATH)
names = ["localisation (cells, counts, candidates, tapers, ranks) + x requested", "gather of the records requested", "... landed", "x' split, Gram + Z",
         "Gershgorin + table header", "recurrence", "output products + stores", "flags"]
for G in [int(a) for a in sys.argv[1:]] or [16, 100000]:
    case = bench.make_case(G, 40, 2, dev)
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, fuse_tile_lists=True)
    for _ in range(4):
        r.assimilate(*case)
    torch.cuda.synchronize()
    nt = min((G + 15) // 16, 8192)
    buf = np.zeros((nt, 20), dtype=np.int64)
    assert lib.mia_debug_tile2f_stamps(buf.ctypes.data_as(C.c_void_p), nt) == 0
    dt = np.diff(buf[:, :9], axis=1).astype(np.float64)
    print("G = %d (%s): %d tiles; wave lifetime median %.0f cycles, p90 %.0f" % (G, r.dominant_kernel_name, nt, np.median(buf[:, 8] - buf[:, 0]), np.percentile(buf[:, 8] - buf[:, 0], 90)))
    for i, n in enumerate(names):
        print("  %-72s median %8.0f   p90 %8.0f" % (n, np.median(dt[:, i]), np.percentile(dt[:, i], 90)))
    lt = np.diff(buf[:, 12:18], axis=1).astype(np.float64)      # (rows indexed by tile, not by block: medians only)
    for i, n in enumerate(["grid coordinates -> cells (LDS)", "box, per-cell counts -> prefix", "candidates located, fetched -> LDS", "trips: tapers, union, sqrt(rho)", "ranks -> slots"]):
        print("      localisation: %-44s median %8.0f   p90 %8.0f" % (n, np.median(lt[:, i]), np.percentile(lt[:, i], 90)))
    t0, t1 = buf[:, 10], buf[:, 11]
    lo = t0.min()
    span = t1.max() - lo
    print("  kernel span: %.2f us (first wave start -> last wave end); wave life in real time: median %.2f us; shader clock %.2f GHz" %
          (span / 100.0, np.median(t1 - t0) / 100.0, np.median((buf[:, 8] - buf[:, 0]) / np.maximum(t1 - t0, 1)) / 10.0))
    edges = np.linspace(0, span, 13)
    for a, b in zip(edges[:-1], edges[1:]):
        mid = lo + 0.5 * (a + b)
        print("    t = %5.1f us: %5d waves resident, %5d started so far" % (0.5 * (a + b) / 100.0, int(((t0 <= mid) & (t1 > mid)).sum()), int((t0 <= mid).sum())))
    r.close()
