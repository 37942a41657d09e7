import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
from oracle import letkf_oracle as O
mia.build()
eng = mia.LetkfEngine("cuda:0")
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
for G, inf in ((203, 1.0), (7, 1.1), (16, 1.0)):
    case = O.synthetic_case(G, 40, 2, seed=41, m=1)
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    out = {}
    for sp in (1, 0):
        _cabi.set_option("tile_split", sp)
        xa, fl = eng.analysis(dev(case["state"]), dev(case["yb"]), dev(case["d"]), nb, inf, return_flags=True, method="matfun")
        out[sp] = (xa.cpu().numpy(), fl.cpu().numpy())
    xs, fs = out[1]; xf, ff = out[0]
    ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, inf)
    bad = np.nonzero((fs & 0xff) != 0)[0]
    print("G", G, "flags bad points:", bad[:20], "f32 flags max", (ff & 0xff).max())
    print("  err split vs ref %.3e   f32 vs ref %.3e   split vs f32 %.3e" % (rel(xs, ref), rel(xf, ref), rel(xs, xf)))
    nanpts = np.nonzero(~np.isfinite(xs).all(axis=(0, 1)))[0]
    print("  nonfinite points:", nanpts[:20])
    if len(nanpts):
        g = nanpts[0]; print("  members nonfinite at point", g, np.nonzero(~np.isfinite(xs[0, :, g]))[0])
    g = G - 3
    print("  point", g, "split", xs[0, :4, g], "f32", xf[0, :4, g], "deg", fs[g] >> 8, ff[g] >> 8, "cnt", int(nb.cnt[g]))
