#!/usr/bin/env python3
"""Runs only the hot kernels a few times on a BASELINE config (for rocprofv3 --pmc / --kernel-trace)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch_assimilate_amd as mia  # noqa: E402
mia.build()

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--grid", type=int, default=100000)
ap.add_argument("--route", default="tiles", choices=["tiles", "lists", "step"],
                help="tiles: tile lists + split records + letkf_tile2_kernel (configs 2, 4); lists: per-point lists + round-2 kernels; "
                     "step: whole steps of the native driver, one at a time, the wavefronts localising their own tiles (letkf_tile2f_kernel)")
ap.add_argument("--weights", action="store_true", help="tiles route: also the (G, k, k) weights (letkf_tile2w_kernel)")
a = ap.parse_args()
k, stride, c, gamma = {"c2": (40, 2, 10.0, None), "c4": (80, 1, 16.5, None), "c5": (40, 2, 10.0, 0.5)}[a.config]
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
X, gx, ox, Yb, d = bench.make_case(a.grid, k, stride, dev)
if a.route == "step":
    r = mia.ShardedLetkf(dev, 0, 1, radii=[c], inf_factor=1.1, rbf_gamma=gamma, fuse_tile_lists=True)
    for _ in range(a.reps + 2):
        out = r.assimilate(X, gx, ox, Yb, d)
    torch.cuda.synchronize()
    print("kernel", r.dominant_kernel_name, "flags ok", r.last_flags_ok(), "p_max", r.last_p_max, "mean degree", r.mean_degree())
    r.close()
    sys.exit(0)
nb = eng.localize(gx, ox, [c])
if a.route == "tiles" and gamma is not None:          # RBF-kernelised filter on tiles (csrc/lketkf_tile.hip)
    for _ in range(a.reps):
        tiles = eng.localize_tiles(gx, ox, [c], nb.p_max)
        xa, fl, retry = eng.analysis_tiles_rbf(X, Yb, d, tiles, 1.1, gamma)
    print("tile stats", tiles.stats.tolist(), "declined", int(retry.item()))
elif a.route == "tiles":
    for _ in range(a.reps):
        tiles = eng.localize_tiles(gx, ox, [c], nb.p_max)
        srec = eng.pack_split(Yb, d)
        if a.weights:
            xa, W, fl, retry = eng.weights_tiles(X, srec, Yb.shape[1], tiles, 1.1)
        else:
            xa, fl, retry = eng.analysis_tiles(X, srec, Yb.shape[1], tiles, 1.1)
    print("tile stats", tiles.stats.tolist(), "declined", int(retry.item()))
else:
    rec = eng.pack_obs(Yb, d, torch.float32)
    for _ in range(a.reps):
        xa, fl = eng.analysis(X, None, None, nb, 1.1, rec=rec, return_flags=True, rbf_gamma=gamma)
torch.cuda.synchronize()
f = fl.cpu().numpy()
import numpy as np
print("p_max", nb.p_max, "flags", int((f & 0xff).max()))
print("degree histogram (flags bits 8-15)", np.bincount((f >> 8) & 0xff))
