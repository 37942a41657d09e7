#!/usr/bin/env python3
"""GPU-side timeline of the analysis queue inside the pipelined loop WITHOUT a profiler: every step's analysis kernel carries
its own start / stop events (the dispatch's timestamps); prints kernel durations, the idle gaps between consecutive
analysis kernels and how the gaps line up with the step index (pipeline slot = i mod depth, preparation stream = i mod n)."""
import argparse, collections, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
from torch_assimilate_amd.sharded import ShardedLetkf
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1200)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--prep-streams", type=int, default=5)
ap.add_argument("--geometry", action="store_true")
a = ap.parse_args()
mia.build()
dev = torch.device("cuda:0")
X, gx, ox, Yb, d = bench.make_case(100000, 40, 2, dev)
runner = ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, comm_chunks=1, max_in_flight=max(2, a.depth), prep_streams=a.prep_streams,
                      copy_results=False)
gid = "g" if a.geometry else None


def run(n, timed):
    pend = collections.deque()
    for _ in range(n):
        if timed:
            runner.time_next_step()
        pend.append(runner.submit(X, gx, ox, Yb, d, geometry_id=gid))
        if len(pend) == a.depth:
            pend.popleft().result()
    while pend:
        pend.popleft().result()


run(300, False)
import gc
gc.collect(); gc.freeze()
run(100, False)
runner.kernel_timings.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
run(a.steps, True)
torch.cuda.synchronize()
per = (time.perf_counter() - t0) / a.steps * 1e6
ev = runner.kernel_timings
dur = np.array([s.elapsed_time(e) for s, e in ev]) * 1e3
gap = np.array([ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(len(ev) - 1)]) * 1e3
print("period %.1f us/step (every step timed); analysis kernel: median %.1f us, mean %.1f; gap to the next analysis kernel: median %.1f, mean %.1f us"
      % (per, np.median(dur), dur.mean(), np.median(gap), gap.mean()))
print("gap histogram (us):", {"<5": int((gap < 5).sum()), "5-10": int(((gap >= 5) & (gap < 10)).sum()), "10-20": int(((gap >= 10) & (gap < 20)).sum()),
                              "20-40": int(((gap >= 20) & (gap < 40)).sum()), "40-80": int(((gap >= 40) & (gap < 80)).sum()), ">=80": int((gap >= 80).sum())})
idx = np.arange(len(gap))
for mod in (a.prep_streams, a.depth):
    print("mean gap by (step mod %d):" % mod, np.array2string(np.array([gap[idx % mod == r].mean() for r in range(mod)]), precision=1))
print("first 48 gaps:", np.array2string(gap[:48], precision=0, max_line_width=220))
print("first 48 durations:", np.array2string(dur[:48], precision=0, max_line_width=220))
