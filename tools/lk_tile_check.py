"""Config 5 on the tile route (csrc/lketkf_tile.hip) against the one-point-per-wavefront route and the oracle; kernel time."""
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia       # noqa: E402
import bench                             # noqa: E402
from oracle import letkf_oracle as O     # noqa: E402

mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 40
gamma = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
X, gx, ox, Yb, d = bench.make_case(G, k, 2, dev, seed=43)
nb = eng.localize(gx, ox, [10.0])
tiles = eng.localize_tiles(gx, ox, [10.0], nb.p_max)
print("p_max", nb.p_max, "tile stats", tiles.stats.tolist())
res = eng.analysis_tiles_rbf(X, Yb, d, tiles, 1.1, gamma)
assert res is not None
xa, fl, retry = res
torch.cuda.synchronize()
print("retry", int(retry.item()), "flags", int((fl & 0xff).max().item()), "mean degree", float((fl >> 8).float().mean().item()),
      "max degree", int((fl >> 8).max().item()))
rec = eng.pack_obs(Yb, d, torch.float32)
xo = eng.analysis(X, None, None, nb, 1.1, rec=rec, rbf_gamma=gamma, method="matfun")
print("rel diff vs one-point-per-wave route: %.3e" % float(torch.linalg.norm(xa - xo) / torch.linalg.norm(xo)))
pts = np.random.RandomState(2).choice(G, 32, replace=False)
st, yb_h, d_h = X.double().cpu().numpy(), Yb.double().cpu().numpy(), d.double().cpu().numpy()
gxh, oxh = gx.cpu().numpy(), ox.cpu().numpy()
core = lambda a, b, i: O.ketkf_weights(a, b, lambda x, y: O.rbf_kernel(x, y, gamma), i)      # noqa: E731
ref = []
for g in pts:
    lo, hi = max(0, int(g) - 200), min(G, int(g) + 200)
    sel = (oxh >= gxh[lo]) & (oxh <= gxh[hi - 1])
    w = O.localized_weights(O.abs_distance_1d(gxh[g], oxh[sel]), yb_h[:, sel], d_h[sel], [10.0], 1.1, core=core)
    ref.append(O.apply_weights(st[:, :, [g]], w[None])[:, :, 0])
ref = np.stack(ref, axis=-1)
ti = torch.as_tensor(pts, device=dev)
got = xa[:, :, ti].double().cpu().numpy()
goto = xo[:, :, ti].double().cpu().numpy()
mean = st.mean(axis=1, keepdims=True)[:, :, pts]
print("vs oracle: tile %.3e (increments %.3e)   point route %.3e" % (
    np.linalg.norm(got - ref) / np.linalg.norm(ref), np.linalg.norm(got - ref) / np.linalg.norm(ref - mean),
    np.linalg.norm(goto - ref) / np.linalg.norm(ref)))
for name, fn in (("tile", lambda: eng.analysis_tiles_rbf(X, Yb, d, tiles, 1.1, gamma)),
                 ("point", lambda: eng.analysis(X, None, None, nb, 1.1, rec=rec, rbf_gamma=gamma, method="matfun"))):
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%s route: %.4f ms per %d analyses (median of 5)" % (name, float(np.median(ts[1:])), G))
