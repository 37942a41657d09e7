#!/usr/bin/env python3
"""cProfile of the caller's side of the pipelined step loop (submit / result) on a tiny grid, where the GPU is never the bound:
where the ~40 us of host time per step go.  python tools/host_profile.py [--grid 1600]"""
import argparse, collections, cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
from torch_assimilate_amd.sharded import ShardedLetkf
ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=1600)
ap.add_argument("--steps", type=int, default=3000)
a = ap.parse_args()
mia.build()
dev = torch.device("cuda:0")
X, gx, ox, Yb, d = bench.make_case(a.grid, 40, 2, dev)
runner = ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, comm_chunks=1, max_in_flight=8, copy_results=False)


def run(n):
    pend = collections.deque()
    for _ in range(n):
        pend.append(runner.submit(X, gx, ox, Yb, d))
        if len(pend) == 8:
            pend.popleft().result()
    while pend:
        pend.popleft().result()


run(300)
import gc
gc.collect(); gc.freeze()
pr = cProfile.Profile()
pr.enable()
run(a.steps)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
