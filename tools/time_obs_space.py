#!/usr/bin/env python3
"""Timing of the observation-space preparation kernels (row f1)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
g = torch.Generator(device=dev); g.manual_seed(1)

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

k = 40
for P in (50000, 500000):
    hx = torch.randn((k, P), generator=g, device=dev); y = torch.randn(P, generator=g, device=dev)
    var = torch.rand(P, generator=g, device=dev) + 0.5
    Yb = torch.empty((k, P), device=dev); d = torch.empty(P, device=dev); rec = torch.empty((P, 44), device=dev)
    t = bench(lambda: eng.obs_space(hx, y, var=var, dtype=torch.float32, out=(Yb, d, rec)))
    print("uncorrelated R f32  k=%d P=%d: %.4f ms (Yb + d + records)" % (k, P, t))
for P in (200, 1000, 4000):
    hx = torch.randn((k, P), generator=g, device=dev, dtype=torch.float64); y = torch.randn(P, generator=g, device=dev, dtype=torch.float64)
    x = torch.arange(P, device=dev, dtype=torch.float64)
    cov = torch.exp(-(x[:, None] - x[None, :]).abs() / 3.0) + 0.1 * torch.eye(P, device=dev, dtype=torch.float64)
    t = bench(lambda: eng.obs_space(hx, y, cov=cov, dtype=torch.float64), n=5)
    print("correlated R f64    k=%d P=%d: %.3f ms (blocked Cholesky sweep, incl. the info read-back)" % (k, P, t))
