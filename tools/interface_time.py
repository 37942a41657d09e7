"""The drop-in classes at config 2: LETKF.analyse_arrays / estimate_weights_arrays + apply (the reference's update_state flow,
filter.py:157-164) end to end, device-resident inputs, wall clock incl. the host -- beside the step driver's serial step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
import bench
mia.build()
dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev)
gxx, oxx = gx[:, None] if gx.dim() == 1 else gx, ox[:, None] if ox.dim() == 1 else ox


def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


a = mia.LETKF(mia.GaspariCohn(10.0, mia.AbsoluteDistance()), inf_factor=1.1, dtype=torch.float32)
ms, xa = timed(lambda: a.analyse_arrays(X, Yb, d, grid_coords=gxx, obs_coords=oxx))
print("LETKF.analyse_arrays (fused path): %.3f ms per call (%d points)" % (ms, G), flush=True)
ms2, W = timed(lambda: a.estimate_weights_arrays(Yb, d, grid_coords=gxx, obs_coords=oxx))
ms3, xa2 = timed(lambda: a.engine.apply_local_weights(X, W))
print("estimate_weights_arrays: %.3f ms, apply_local_weights: %.3f ms; both routes differ by %.1e" %
      (ms2, ms3, float(torch.linalg.norm(xa2 - xa) / torch.linalg.norm(xa))), flush=True)
r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
ms4, xs = timed(lambda: r.assimilate(X, gx, ox, Yb, d), reps=50)
print("ShardedLetkf.assimilate (step driver): %.3f ms; differs from analyse_arrays by %.1e" % (ms4, float(torch.linalg.norm(xs - xa) / torch.linalg.norm(xa))), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    a.analyse_arrays(X, Yb, d, grid_coords=gxx, obs_coords=oxx)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
