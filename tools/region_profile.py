#!/usr/bin/env python3
"""Where a SHORT timed region (the round driver's --steps 20) spends its time: host timestamps of every submit() / result() of
a 20-step region bracketed by synchronize, median over many regions.  python tools/region_profile.py [--steps 20] [--depth 8]"""
import argparse, collections, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
from torch_assimilate_amd.sharded import ShardedLetkf
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--prep-streams", type=int, default=5)
a = ap.parse_args()
mia.build()
dev = torch.device("cuda:0")
X, gx, ox, Yb, d = bench.make_case(100000, 40, 2, dev)
runner = ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, comm_chunks=1, max_in_flight=max(2, a.depth), prep_streams=a.prep_streams,
                      copy_results=False)


def region(n, depth):
    sub, res = [], []
    pend = collections.deque()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pend.append(runner.submit(X, gx, ox, Yb, d))
        sub.append(time.perf_counter() - t0)
        if len(pend) == depth:
            pend.popleft().result()
            res.append(time.perf_counter() - t0)
    while pend:
        pend.popleft().result()
        res.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    return np.array(sub) * 1e6, np.array(res) * 1e6, (time.perf_counter() - t0) * 1e6


for _ in range(30):
    region(a.steps, a.depth)
import gc
gc.collect(); gc.freeze()
S, R, T = [], [], []
for _ in range(200):
    s, r, t = region(a.steps, a.depth)
    S.append(s); R.append(r); T.append(t)
S, R, T = np.median(S, axis=0), np.median(R, axis=0), np.median(T)
print("region of %d steps, depth %d: %.0f us = %.1f us per step" % (a.steps, a.depth, T, T / a.steps))
print("submit returns at (us):", np.array2string(S, precision=0, max_line_width=200))
print("result returns at (us):", np.array2string(R, precision=0, max_line_width=200))
print("result-to-result (us): ", np.array2string(np.diff(R), precision=0, max_line_width=200))
