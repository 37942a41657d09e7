#!/usr/bin/env python3
"""Where does the host spend its time in the pipelined bench loop?  Per-step wall time of submit() and result(), the
largest outliers and their position -- to tell a host-side stall (allocator, GC, scheduler) from GPU time."""
import collections, os, sys, time, gc
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
from torch_assimilate_amd.sharded import ShardedLetkf
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
G = 100000
X, gx, ox, Yb, d = bench.make_case(G, bench.K_ENS, bench.OBS_STRIDE, dev)
r = ShardedLetkf(dev, 0, 1, radii=[bench.GC_RADIUS], inf_factor=bench.INF, method="auto", comm_chunks=1, native_step=True, max_in_flight=4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
timing = len(sys.argv) > 2 and sys.argv[2] == "events"
for rep in range(3):
    pend = collections.deque()
    ts, tr = [], []
    torch.cuda.synchronize()
    t00 = time.perf_counter()
    for it in range(n):
        if timing and it % 4 == 0:
            r.time_next_step()
        t0 = time.perf_counter()
        pend.append(r.submit(X, gx, ox, Yb, d))
        t1 = time.perf_counter()
        if len(pend) == 4:
            pend.popleft().result()
        t2 = time.perf_counter()
        ts.append(t1 - t0); tr.append(t2 - t1)
    while pend:
        pend.popleft().result()
    torch.cuda.synchronize()
    el = time.perf_counter() - t00
    tot = [a + b for a, b in zip(ts, tr)]
    big = sorted(range(n), key=lambda i: -tot[i])[:6]
    print("rep %d: %.4f ms/step   submit mean %.1f us  result mean %.1f us   gc counts %s" % (rep, 1e3 * el / n, 1e6 * sum(ts) / n, 1e6 * sum(tr) / n, gc.get_count()))
    print("   largest steps:", ", ".join("#%d submit %.0f us result %.0f us" % (i, 1e6 * ts[i], 1e6 * tr[i]) for i in big), flush=True)
import ctypes as C
a, b, nj = C.c_double(), C.c_double(), C.c_longlong()
r.engine.lib.mia_letkf_step_launch_stats(C.byref(a), C.byref(b), C.byref(nj))
print("launch threads: preparation stage %.1f us per step, analysis + read-back stage %.1f us per step (%d steps)" % (a.value / nj.value, b.value / nj.value, nj.value))
