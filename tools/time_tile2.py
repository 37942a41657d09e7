#!/usr/bin/env python3
"""Timing of the tile route's three kernels (tile lists, split records, analysis) beside the round-2 kernel on per-point
lists, HIP events over batches of back-to-back launches (min / median).  python tools/time_tile2.py [c2|c4] [--grid N]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", default=["c2"])
ap.add_argument("--grid", type=int, default=100000)
ap.add_argument("--batches", type=int, default=9)
ap.add_argument("--m", type=int, default=1)
a = ap.parse_args()
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
cfgs = {"c2": (40, 2, 10.0), "c4": (80, 1, 16.5)}


def timed(fn, n, batches):
    ts = []
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    ts = np.array(ts[1:])
    return "min %.4f  median %.4f ms" % (ts.min(), np.median(ts))


for name in a.configs:
    k, stride, c = cfgs[name]
    X, gx, ox, Yb, d = bench.make_case(a.grid, k, stride, dev)
    if a.m > 1:
        X = X.repeat(a.m, 1, 1).contiguous()
    nb = eng.localize(gx, ox, [c])
    P = Yb.shape[1]
    tiles = eng.localize_tiles(gx, ox, [c], nb.p_max)
    rec = eng.pack_split(Yb, d)
    out = torch.empty((X.shape[0], k, a.grid), dtype=torch.float32, device=dev)
    xa, fl, retry = eng.analysis_tiles(X, rec, P, tiles, 1.1, out=out)
    frec = eng.pack_obs(Yb, d, torch.float32)
    xo = eng.analysis(X, None, None, nb, 1.1, rec=frec, method="matfun", defer_retry=True)
    xo = xo[0] if isinstance(xo, tuple) else xo
    err = float(torch.linalg.norm(xa - xo) / torch.linalg.norm(xo))
    print(f"{name}: G {a.grid} k {k} p_max {nb.p_max} stats {tiles.stats.tolist()} retry {int(retry.item())} "
          f"mean degree {float(((fl >> 8) & 0xff).float().mean()):.2f}  tile2 vs round-2 kernel {err:.2e}")
    print(f"  analysis_tiles   : {timed(lambda: eng.analysis_tiles(X, rec, P, tiles, 1.1, out=out), 20, a.batches)}")
    print(f"  round-2 kernel   : {timed(lambda: eng.analysis(X, None, None, nb, 1.1, rec=frec, method='matfun', defer_retry=True), 20, a.batches)}")
    print(f"  localize_tiles   : {timed(lambda: eng.localize_tiles(gx, ox, [c], nb.p_max), 10, a.batches)}  (index build + tile lists)")
    print(f"  localize (lists) : {timed(lambda: eng.localize(gx, ox, [c], assume_p_max=nb.p_max), 10, a.batches)}  (index build + per-point lists)")
    print(f"  pack_split       : {timed(lambda: eng.pack_split(Yb, d), 10, a.batches)}")
    print(f"  pack_obs (f32)   : {timed(lambda: eng.pack_obs(Yb, d, torch.float32), 10, a.batches)}")
