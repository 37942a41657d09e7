#!/usr/bin/env python3
"""Weights tile kernel against the oracle over a list of (k, stride, c) cases: untouched / non-finite / wrong points."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
from oracle import letkf_oracle as O
mia.build()
eng = mia.LetkfEngine("cuda:0")
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")
for spec in sys.argv[1:]:
    k, stride, c = spec.split(",")
    k, stride, c = int(k), int(stride), float(c)
    case = O.synthetic_case(203, k, stride, seed=3 * k + 1, m=1)
    nb = eng.localize(case["grid_x"], case["obs_x"], [c])
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max)
    if tiles.stats.tolist()[1]:
        tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max, extra_blocks=1)
    rec = eng.pack_split(dev(case["yb"]), dev(case["d"]))
    r = eng.weights_tiles(dev(case["state"]), rec, case["yb"].shape[1], tiles, 1.1)
    if r is None:
        print(spec, "unsupported"); continue
    xa, W, fl, retry = r
    W = W.cpu().numpy()
    ref_xa, ref_w = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c, 1.1)
    err = np.abs(W - ref_w).reshape(203, -1).max(axis=1)
    nanp = np.flatnonzero(~np.isfinite(err)); bad = np.flatnonzero(np.isfinite(err) & (err > 1e-4))
    zero = np.flatnonzero((W.reshape(203, -1) == 0).all(axis=1))
    print(spec, "p_max", nb.p_max, "ut", tiles.ut, "| nan points", nanp.tolist()[:20], "| all-zero points", zero.tolist()[:40],
          "| other wrong", [g for g in bad.tolist() if g not in zero][:20], "| max err of the rest %.2e" % np.nanmax(np.where(err < 1e-4, err, 0)))
