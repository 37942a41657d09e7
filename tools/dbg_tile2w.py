#!/usr/bin/env python3
"""Debug aid for the weights tile kernel: one synthetic case against the oracle, where and how the weights differ."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
from oracle import letkf_oracle as O
k, stride, c = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
mia.build()
eng = mia.LetkfEngine("cuda:0")
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
case = O.synthetic_case(203, k, stride, seed=3 * k + 1, m=1)
nb = eng.localize(case["grid_x"], case["obs_x"], [c])
tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max)
if tiles.stats.tolist()[1]:
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max, extra_blocks=1)
print("p_max", nb.p_max, "ut", tiles.ut, "stats", tiles.stats.tolist())
rec = eng.pack_split(dev(case["yb"]), dev(case["d"]))
r = eng.weights_tiles(dev(case["state"]), rec, case["yb"].shape[1], tiles, 1.1)
if r is None:
    print("unsupported"); sys.exit(0)
xa, W, fl, retry = r
W = W.cpu().numpy(); f = fl.cpu().numpy()
ref_xa, ref_w = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c, 1.1)
print("flags", np.unique(f & 0xff, return_counts=True), "retry", int(retry.item()))
err = np.abs(W - ref_w).reshape(203, -1).max(axis=1)
print("max abs err per point (first 40):", np.array2string(err[:40], precision=2))
g = int(np.nanargmax(np.where(np.isfinite(err), err, 1e30)))
print("worst point", g, "flag", f[g], "finite", np.isfinite(W[g]).all())
np.set_printoptions(precision=4, linewidth=200, suppress=True)
e = np.abs(W[g] - ref_w[g])
print("err by row block / col block (16):")
kt = (k + 15) // 16
for a in range(kt):
    print([float(e[16 * a:16 * a + 16, 16 * b:16 * b + 16].max()) for b in range(kt)])
print("W[g][:6,:6]\n", W[g][:6, :6], "\nref\n", ref_w[g][:6, :6])
print("w_mean check: row means of (W - ref):", (W[g] - ref_w[g]).mean(axis=1)[:8])
for g in (0, 8, 10):
    print("point", g, "deg", f[g] >> 8, "\nW[:4,:8]\n", W[g][:4, :8], "\nref\n", ref_w[g][:4, :8])
    print(" last rows/cols W[k-3:, k-3:]\n", W[g][k - 3:, k - 3:], "\nref\n", ref_w[g][k - 3:, k - 3:])
