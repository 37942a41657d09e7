"""Accuracy sweep of the kernelised tile route (csrc/lketkf_tile.hip) against the float64 oracle: ensemble sizes (padded and not),
RBF gamma, observation strength.  MIA_BUILD_FLAGS=-DMIA_EXPERIMENTS MIA_LK_MARGIN=0|1|2 selects the degree margin."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia       # noqa: E402
import bench                             # noqa: E402
from oracle import letkf_oracle as O     # noqa: E402

mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
G = 2048
worst = 0.0
for k in (40, 32, 37, 20, 8, 5):
    for stride, c in ((2, 10.0), (1, 4.0), (3, 25.0)):
        X, gx, ox, Yb, d = bench.make_case(G, k, stride, dev, seed=43 + k)
        nb = eng.localize(gx, ox, [c])
        tiles = eng.localize_tiles(gx, ox, [c], nb.p_max)
        if int(tiles.stats[1].item()):
            tiles = eng.localize_tiles(gx, ox, [c], nb.p_max, extra_blocks=1)
        for gamma in (0.01, 0.5, 10.0):
            for strength in (1.0, 10.0, 0.1):
                Ybs, ds = Yb * strength, d * strength
                res = eng.analysis_tiles_rbf(X, Ybs, ds, tiles, 1.1, gamma)
                if res is None:
                    print("k=%d stride=%d: outside the kernel" % (k, stride))
                    break
                xa, fl, retry = res
                pts = np.random.RandomState(5).choice(G, 12, replace=False)
                st, yb_h, d_h = X.double().cpu().numpy(), Ybs.double().cpu().numpy(), ds.double().cpu().numpy()
                gxh, oxh = gx.cpu().numpy(), ox.cpu().numpy()
                core = lambda a, b, i, g_=gamma: O.ketkf_weights(a, b, lambda x, y: O.rbf_kernel(x, y, g_), i)      # noqa: E731
                ref = np.stack([O.apply_weights(st[:, :, [g]], O.localized_weights(O.abs_distance_1d(gxh[g], oxh), yb_h, d_h, [c], 1.1,
                                                                                  core=core)[None])[:, :, 0] for g in pts], axis=-1)
                got = xa[:, :, torch.as_tensor(pts, device=dev)].double().cpu().numpy()
                mean = st.mean(axis=1, keepdims=True)[:, :, pts]
                e = np.linalg.norm(got - ref) / np.linalg.norm(ref)
                ei = np.linalg.norm(got - ref) / max(np.linalg.norm(ref - mean), 1e-300)
                worst = max(worst, e)
                print("k=%2d p_max=%2d gamma=%5.2f strength=%5.1f: %.2e (increments %.2e) retry %d mean deg %.1f max %d" % (
                    k, nb.p_max, gamma, strength, e, ei, int(retry.item()), float((fl >> 8).float().mean().item()), int((fl >> 8).max().item())))
print("worst relative Frobenius error: %.3e" % worst)
