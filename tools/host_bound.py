#!/usr/bin/env python3
"""Is the pipelined step loop bound by the host?  Host time per step of the caller (submit / result) and of the library's two
launch threads (mia_letkf_step_launch_stats) beside the measured step period.  python tools/host_bound.py [--depth 8]"""
import argparse, collections, ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
from torch_assimilate_amd.sharded import ShardedLetkf
ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--grid", type=int, default=100000)
ap.add_argument("--prep-streams", type=int, default=3)
ap.add_argument("--options", default="", help="name=value ... for mia_set_option")
ap.add_argument("--no-in-event", action="store_true", help="experiment: the preparation stream does not wait for the caller's stream")
a = ap.parse_args()
mia.build()
for o in a.options.split():
    _cabi.set_option(o.split("=")[0], int(o.split("=")[1]))
dev = torch.device("cuda:0")
X, gx, ox, Yb, d = bench.make_case(a.grid, 40, 2, dev)
runner = ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, method="auto", comm_chunks=1, native_step=True,
                          max_in_flight=max(2, a.depth), prep_streams=a.prep_streams, copy_results=False)
lib = _cabi.lib()


def stats():
    ta, tb, n = C.c_double(), C.c_double(), C.c_longlong()
    lib.mia_letkf_step_launch_stats(C.byref(ta), C.byref(tb), C.byref(n))
    return ta.value, tb.value, n.value


def run(n):
    pend, t_sub, t_res = collections.deque(), 0.0, 0.0
    for _ in range(n):
        t0 = time.perf_counter()
        pend.append(runner.submit(X, gx, ox, Yb, d))
        t1 = time.perf_counter()
        if len(pend) == a.depth:
            pend.popleft().result()
        t2 = time.perf_counter()
        t_sub += t1 - t0
        t_res += t2 - t1
    while pend:
        pend.popleft().result()
    return t_sub, t_res


run(200)
if a.no_in_event:
    import ctypes
    for slot in runner._native["slots"]:
        if slot.get("args") is not None:
            slot["args"].in_event = ctypes.POINTER(ctypes.c_void_p)()
import gc
gc.collect(); gc.freeze()
run(200)
torch.cuda.synchronize()
s0 = stats()
co0 = _cabi.step_coalesce_stats()
t0 = time.perf_counter()
t_sub, t_res = run(a.steps)
torch.cuda.synchronize()
el = time.perf_counter() - t0
s1 = stats()
n = s1[2] - s0[2]
print(a.options, "depth", a.depth, "steps/launch %.2f" % ((lambda l0, l1: (l1[1] - l0[1]) / max(1, l1[0] - l0[0]))(co0, _cabi.step_coalesce_stats())))
print("period %.1f us/step;  caller: submit %.1f us, result() %.1f us;  launch thread A (preparation) %.1f us, thread B "
      "(analysis + read-back, excluding its wait for the preparation) %.1f us  [%d jobs]"
      % (1e6 * el / a.steps, 1e6 * t_sub / a.steps, 1e6 * t_res / a.steps, (s1[0] - s0[0]) / n, (s1[1] - s0[1]) / n, n))
