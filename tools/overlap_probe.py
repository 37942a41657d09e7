#!/usr/bin/env python3
"""Do analysis kernels of different streams overlap usefully?  The lists-in-memory tile kernel (letkf_tile2_kernel, prebuilt tile lists
and records: nothing else on the GPU) launched back to back, round robin over 1 .. 4 streams: period per launch.
python tools/overlap_probe.py [--grid 100000]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=100000)
ap.add_argument("--n", type=int, default=400)
a = ap.parse_args()
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
X, gx, ox, Yb, d = bench.make_case(a.grid, 40, 2, dev)
nb = eng.localize(gx, ox, [10.0])
tiles = eng.localize_tiles(gx, ox, [10.0], nb.p_max)
srec = eng.pack_split(Yb, d)
P = Yb.shape[1]
outs = [torch.empty_like(X) for _ in range(4)]
torch.cuda.synchronize()
for S in (1, 2, 3, 4, 1):
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.n):
            with torch.cuda.stream(streams[i % S]):
                eng.analysis_tiles(X, srec, P, tiles, 1.1)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
    print("%d stream(s): %.1f us per launch (host enqueue alone %.1f us per launch)" % (S, 1e6 * t_all / a.n, 1e6 * t_host / a.n), flush=True)
