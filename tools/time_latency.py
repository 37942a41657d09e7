#!/usr/bin/env python3
"""Analysis kernel time against the number of grid points (C2 geometry): the single-wavefront latency (G <= 1 wave per
SIMD) next to the saturated throughput -- how many wavefronts per SIMD does the kernel need to hide its own latency?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
K = 40
for G in (256, 1024, 4096, 16384, 32768, 49152, 65536, 100000):
    X, gx, ox, Yb, d = bench.make_case(G, K, 2, dev)
    nb = eng.localize(gx, ox, [10.0])
    rec = eng.pack_obs(Yb, d, torch.float32)
    out = torch.empty_like(X)
    fn = lambda: eng.analysis(X, None, None, nb, 1.1, rec=rec, method="matfun", defer_retry=True, out=out)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print("G = %6d   %8.1f us per launch   %6.2f ns per point   (%.2f waves, %.2f 16-point tiles per SIMD)" % (G, best * 1e3, best * 1e6 / G, G / 1024.0, G / 16384.0), flush=True)
