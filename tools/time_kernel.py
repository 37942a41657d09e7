#!/usr/bin/env python3
"""Robust kernel timing: min / median over batches (the shared GPU boxes show occasional multi-ms outliers)."""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", default=["c2"])
ap.add_argument("--methods", default="matfun,eig")
ap.add_argument("--batches", type=int, default=7)
ap.add_argument("--grid", type=int, default=100000)
ap.add_argument("--option", action="append", default=[], help="name=value for mia_set_option (e.g. tile_split=0)")
a = ap.parse_args()
mia.build()
from torch_assimilate_amd import _cabi
for o in a.option:
    _cabi.set_option(o.split("=")[0], int(o.split("=")[1]))
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
cfgs = {"c2": (40, 2, 10.0, None), "c4": (80, 1, 16.5, None), "c5": (40, 2, 10.0, 0.5)}
for name in a.configs:
    k, stride, c, gamma = cfgs[name]
    X, gx, ox, Yb, d = bench.make_case(a.grid, k, stride, dev)
    nb = eng.localize(gx, ox, [c])
    rec = eng.pack_obs(Yb, d, torch.float32)
    for method in a.methods.split(","):
        ts = []
        for b in range(a.batches):
            n = 10 if method == "matfun" or name == "c2" else 3
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): eng.analysis(X, None, None, nb, 1.1, rec=rec, rbf_gamma=gamma, method=method, defer_retry=True)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
        ts = np.array(ts[1:])
        print(f"{name} {method:6s}: min {ts.min():.3f} ms  median {np.median(ts):.3f} ms  max {ts.max():.3f} ms")
