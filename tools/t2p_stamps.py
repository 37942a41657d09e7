"""Phase breakdown of letkf_tile2p_kernel (diagnostic build: MIA_BUILD_FLAGS=-DMIA_T2P_STAMPS)."""
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
names = ["header + slot table + x", "records -> LDS, tails", "Gram, Z, A fragments", "bound: exchange, degree", "recurrence", "output"]
for G in [int(a) for a in sys.argv[1:]] or [16, 100000]:
    X, gx, ox, Yb, d = bench.make_case(G, 80, 1, dev, seed=43)
    tiles = eng.localize_tiles(gx, ox, [16.5], 63)
    rec = eng.pack_split(Yb, d)
    for _ in range(3):
        eng.analysis_tiles(X, rec, Yb.shape[1], tiles, 1.1)
    torch.cuda.synchronize()
    nt = min((G + 15) // 16, 8192)
    buf = np.zeros((nt, 12), dtype=np.int64)
    assert lib.mia_debug_t2p_stamps(buf.ctypes.data_as(C.c_void_p), nt) == 0
    dt = np.diff(buf[:, :7], axis=1).astype(np.float64)
    print("G = %d: %d tiles; workgroup lifetime median %.0f cycles" % (G, nt, np.median(buf[:, 6] - buf[:, 0])))
    for i, n in enumerate(names):
        print("  %-28s median %8.0f   p90 %8.0f" % (n, np.median(dt[:, i]), np.percentile(dt[:, i], 90)))
    t0, t1 = buf[:, 10], buf[:, 11]
    lo = t0.min(); span = t1.max() - lo
    print("  kernel span %.2f us; workgroup life in real time: median %.2f us" % (span / 100.0, np.median(t1 - t0) / 100.0))
    for a, b in zip(np.linspace(0, span, 9)[:-1], np.linspace(0, span, 9)[1:]):
        mid = lo + 0.5 * (a + b)
        print("    t = %6.1f us: %5d workgroups resident" % (0.5 * (a + b) / 100.0, int(((t0 <= mid) & (t1 > mid)).sum())))
    hw = buf[:, 9]
    cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xf) * 16 + ((hw >> 13) & 0x7) * 256 + ((hw >> 12) & 1) * 2048
    u, c = np.unique(cu, return_counts=True)
    print("  distinct CUs seen: %d; workgroups per CU: min %d median %d max %d" % (len(u), c.min(), np.median(c), c.max()))
