#!/usr/bin/env python3
"""Dense local networks (p >> k, primal route): matfun (member Gram streamed from the records on the MFMA units) against
the eigensolver route, 1-D geometry with an observation at every grid point and wide Gaspari-Cohn radii."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
for (k, stride, c, G) in ((40, 1, 25.0, 100000), (40, 1, 100.0, 50000), (20, 1, 250.0, 20000), (80, 1, 50.0, 50000)):
    X, gx, ox, Yb, d = bench.make_case(G, k, stride, dev)
    Yb, d = Yb * 0.3, d * 0.3
    nb = eng.localize(gx, ox, [c])
    rec = eng.pack_obs(Yb, d, torch.float32)
    for method in ("matfun", "eig"):
        ts = []
        for b in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                out, fl, fin = eng.analysis(X, None, None, nb, 1.1, rec=rec, method=method, defer_retry=True, return_flags=True)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3 * 1e3)
        f = fl.cpu().numpy()
        print("k=%d local obs <= %d, G=%d, %-6s: %8.3f ms  %.2e analyses/s  (declined %d)" % (
            k, nb.p_max, G, method, min(ts[1:]), G / min(ts[1:]) * 1e3, int(((f & 8) != 0).sum())))
