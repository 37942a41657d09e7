"""Localised IEnKS at config 2's size: one Gauss-Newton iteration of the per-point weights (transform and bundle variants) and the
final per-point weight transform of the state, against the LETKF weights route."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
import bench
mia.build()
dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev)
loc = mia.GaspariCohn(10.0, mia.AbsoluteDistance())


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


for name, a, scale in (("transform, tau = 1", mia.LocalizedIEnKSTransform(None, loc, tau=1.0, dtype=torch.float32), 1.0),
                       ("transform, tau = 0.8", mia.LocalizedIEnKSTransform(None, loc, tau=0.8, dtype=torch.float32), 1.0),
                       ("bundle, eps = 1e-3", mia.LocalizedIEnKSBundle(None, loc, tau=1.0, epsilon=1e-3, dtype=torch.float32), 1e-3)):
    w = a.generate_prior_weights(40)
    ms1, w1 = timed(lambda: a.inner_loop_arrays(w, Yb * scale, d, grid_coords=gx, obs_coords=ox))
    ms2, w2 = timed(lambda: a.inner_loop_arrays(w1, Yb * scale, d, grid_coords=gx, obs_coords=ox))
    ms3, xa = timed(lambda: a.apply_weights_arrays(X, w2))
    print("%-22s first iteration %.3f ms, second %.3f ms, state transform %.3f ms  (wall clock incl. host, %d points)" % (name, ms1, ms2, ms3, G), flush=True)
