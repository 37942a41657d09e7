"""Many state rows per grid point (m = n_var x n_time of the reference, base.py:257-278): the analysis kernel's row loop against
weights once + the per-point weight transform (what the reference always does: estimate_weights -> _apply_weights)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
import bench
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
X1, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev)
nb = eng.localize(gx, ox, [10.0])
tiles = eng.localize_tiles(gx, ox, [10.0], nb.p_max)
rec = eng.pack_split(Yb, d)
P = Yb.shape[1]


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r


gen = torch.Generator(device=dev); gen.manual_seed(1)
for m in (1, 8, 16, 32, 64, 128):
    X = torch.randn((m, 40, G), generator=gen, device=dev)
    X[0] = X1[0]
    out = torch.empty_like(X)
    t_rows, (xa, fl, rt) = timed(lambda: eng.analysis_tiles(X, rec, P, tiles, 1.1, out=out))
    t_w, (xa1, W, fl2, rt2) = timed(lambda: eng.weights_tiles(X[:1], rec, P, tiles, 1.1))
    t_a, xa2 = timed(lambda: eng.apply_local_weights(X, W))
    err = float(torch.linalg.norm(xa2 - xa) / torch.linalg.norm(xa))
    print("m = %3d: row loop %.3f ms;  weights %.3f + transform %.3f = %.3f ms;  results differ by %.1e" % (m, t_rows, t_w, t_a, t_w + t_a, err), flush=True)
