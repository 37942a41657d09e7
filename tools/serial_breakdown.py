"""Where a serial step's wall time goes on the host: the caller's time in submit (argument set-up, the C call that enqueues the
step's launches, the read-back call) and in finish (the wait for the GPU + validation).  The kernels themselves: rocprofv3."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
import bench
mia.build()
dev = torch.device("cuda:0")
case = bench.make_case(100000, 40, 2, dev)
r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
for _ in range(20):
    r.assimilate(*case)
X, gx, ox, Yb, d = case
G = X.shape[2]
lib = r.engine.lib
acc = {"step": 0.0, "rb": 0.0, "sync": 0.0}


def wrap(name, key):
    fn = getattr(lib, name)

    def w(*a):
        t = time.perf_counter()
        rc = fn(*a)
        acc[key] += time.perf_counter() - t
        return rc
    setattr(lib, name, w)


wrap("mia_letkf_sharded_step_streams_f32", "step")
wrap("mia_letkf_step_readback", "rb")
wrap("mia_event_synchronize", "sync")
N = 500
ts = tf = 0.0
torch.cuda.synchronize()
t_all = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter()
    h = r._native_submit(X, gx, ox, Yb, d, G, 0, G, pipelined=False)
    t1 = time.perf_counter()
    out = r._native_finish(h)
    t2 = time.perf_counter()
    ts += t1 - t0
    tf += t2 - t1
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
u = 1e6 / N
print("serial step %.1f us = submit %.1f (of which the step's C call %.1f, the read-back call %.1f, Python %.1f) + finish %.1f "
      "(of which the wait for the event %.1f, Python %.1f)" % (t_all * u, ts * u, acc["step"] * u, acc["rb"] * u,
                                                            (ts - acc["step"] - acc["rb"]) * u, tf * u, acc["sync"] * u, (tf - acc["sync"]) * u))
