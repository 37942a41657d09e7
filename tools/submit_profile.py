#!/usr/bin/env python3
"""Where does the caller's host time per step in flight go?  cProfile over the submit / result loop of tools/host_bound.py (the
profiler's own overhead inflates every Python-level call: read the ranking, not the absolute times).
python tools/submit_profile.py [--depth 8] [--time-every 0]"""
import argparse, collections, cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
from torch_assimilate_amd.sharded import ShardedLetkf
ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--time-every", type=int, default=0)
ap.add_argument("--profile", type=int, default=1)
a = ap.parse_args()
mia.build()
dev = torch.device("cuda:0")
X, gx, ox, Yb, d = bench.make_case(100000, 40, 2, dev)
runner = ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=a.depth, copy_results=False)


def run(n):
    pend = collections.deque()
    for it in range(n):
        if a.time_every and it % a.time_every == 0:
            runner.time_next_step()
        pend.append(runner.submit(X, gx, ox, Yb, d))
        if len(pend) == a.depth:
            pend.popleft().result()
    while pend:
        pend.popleft().result()


run(300)
import gc
gc.collect(); gc.freeze()
run(300)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(a.steps)
torch.cuda.synchronize()
print("unprofiled: %.1f us/step (time_every %d)" % (1e6 * (time.perf_counter() - t0) / a.steps, a.time_every))
if a.profile:
    pr = cProfile.Profile()
    pr.enable()
    run(a.steps)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
