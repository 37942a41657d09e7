#!/usr/bin/env python3
"""Host-side timeline of a burst of steps in flight (mia_debug_step_trace): per step, when the caller submitted it, when the two
launch threads touched it and how long the caller's own submit / result calls took.  python tools/step_trace.py [--steps 20]"""
import argparse, collections, ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
from torch_assimilate_amd.sharded import ShardedLetkf
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--time-every", type=int, default=4)
ap.add_argument("--options", default="", help="name=value ... for mia_set_option")
a = ap.parse_args()
mia.build()
for o in a.options.split():
    _cabi.set_option(o.split("=")[0], int(o.split("=")[1]))
dev = torch.device("cuda:0")
X, gx, ox, Yb, d = bench.make_case(100000, 40, 2, dev)
r = ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=a.depth, copy_results=False)
clock = lambda: time.clock_gettime_ns(time.CLOCK_MONOTONIC)


def region(n, log=None):
    pend = collections.deque()
    for it in range(n):
        if a.time_every and it % a.time_every == 0:
            r.time_next_step()
        t0 = clock()
        pend.append(r.submit(X, gx, ox, Yb, d))
        t1 = clock()
        if len(pend) == a.depth:
            pend.popleft().result()
        if log is not None:
            log.append((t0, t1, clock()))
    while pend:
        t1 = clock()
        pend.popleft().result()
        if log is not None:
            log.append((None, t1, clock()))


for _ in range(30):
    torch.cuda.synchronize()
    region(a.steps)
import gc
gc.collect(); gc.freeze()
for _ in range(20):
    torch.cuda.synchronize()
    region(a.steps)
torch.cuda.synchronize()
log = []
T0 = clock()
region(a.steps, log)
torch.cuda.synchronize()
T1 = clock()
buf = (C.c_longlong * (8 * a.steps))()
n = _cabi.lib().mia_debug_step_trace(buf, a.steps)
print("region of %d steps: %.1f us (%.1f per step); times in us from the region's start" % (a.steps, (T1 - T0) / 1e3, (T1 - T0) / 1e3 / a.steps))
print("step | caller: submit begins .. ends, then result() until | library: submitted  A begins  A done  B takes  prep seen done  analysis enqueued  read-back enqueued")
us = lambda t: (t - T0) / 1e3
for i in range(n):
    t = [buf[8 * i + q] for q in range(7)]
    entry = buf[8 * i + 7]
    lg = log[i] if i < len(log) else (None, None, None)
    print("%3d  | %7.1f .. %7.1f  result until %7.1f | C entry %7.1f | %7.1f  %7.1f  %7.1f  %7.1f  %7.1f  %7.1f  %7.1f" % (
        i, us(lg[0]) if lg[0] else -1, us(lg[1]), us(lg[2]), us(entry), *[(us(x) if x > 0 else -1.0) for x in t]))
for lg in log[n:]:
    print("drain: result() %7.1f .. %7.1f" % (us(lg[1]), us(lg[2])))
