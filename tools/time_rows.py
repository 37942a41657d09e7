#!/usr/bin/env python3
"""Both analysis routes against the number of state rows m per grid point (C2 geometry), and the weights-output route:
where should the eigensolver-free route hand over to the eigensolver kernel?"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
K, stride, c, gamma = {"c2": (40, 2, 10.0, None), "c4": (80, 1, 16.5, None), "c5": (40, 2, 10.0, 0.5)}[cfg]
G = 100000 if cfg == "c2" else 20000
X1, gx, ox, Yb, d = bench.make_case(G, K, stride, dev)
nb = eng.localize(gx, ox, [c])
print(cfg, "G =", G)
rec = eng.pack_obs(Yb, d, torch.float32)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    return min(ts)


for m in ((1, 2, 4, 6, 8, 12, 16, 32) if cfg == 'c2' else (1, 4, 16, 64)):
    X = torch.randn((m, K, G), device=dev)
    out = torch.empty_like(X)
    tm = timed(lambda: eng.analysis(X, None, None, nb, 1.1, rec=rec, method="matfun", defer_retry=True, out=out, rbf_gamma=gamma))
    te = timed(lambda: eng.analysis(X, None, None, nb, 1.1, rec=rec, method="eig", out=out, rbf_gamma=gamma), n=2) if "--eig" in sys.argv else float("nan")
    print("m = %2d   matfun %7.3f ms   eig %7.3f ms" % (m, tm, te), flush=True)
X = X1
tw = timed(lambda: eng.analysis(X, None, None, nb, 1.1, rec=rec, return_weights=True, rbf_gamma=gamma), n=2)
print("weights output (eig route, m = 1): %.3f ms" % tw)
