"""Where a serial step's wall time goes: the caller's time in submit (argument set-up + the C call that enqueues the step's
launches) and in finish (the wait for the GPU + validation), beside the step's kernels (rocprofv3 gives those)."""
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
ts, tf, tc = 0.0, 0.0, 0.0
import ctypes as C
lib = r.engine.lib
orig = lib.mia_letkf_sharded_step_streams_f32
N = 300
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
print("serial step %.1f us: submit %.1f us (set-up + enqueue), finish %.1f us (wait + validation)" % (1e6 * t_all / N, 1e6 * ts / N, 1e6 * tf / N))
# the C call alone, same arguments, nothing waited for in between (queue fills: launch cost only)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    r.assimilate(*case)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(12)
