#!/usr/bin/env python3
"""Tiny ensembles with large innovations (the analysis mean moves by hundreds of spreads): the round-2 tile kernel with split
half-precision products against its f32 products and the float64 eigensolver, per ensemble size.  python tools/small_k_sweep.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
rs = np.random.RandomState(11)
rel = lambda a, b: float(torch.linalg.norm(a.double() - b.double()) / max(float(torch.linalg.norm(b.double())), 1e-300))
for k in (2, 3, 4, 5, 6, 8, 12, 16, 24, 40):
    worst = {1: 0.0, 0: 0.0, 2: 0.0}
    for case in range(24):
        G = 320
        grid = np.arange(G, dtype=np.float64)[:, None]
        stride = float(rs.choice([1.0, 2.0, 3.0]))
        obs = np.arange(0, G, stride)[:, None] + rs.uniform(-0.2, 0.2)
        p_target = int(rs.randint(1, min(k, 20) + 1))
        c = max(0.6, p_target * stride / 4.0)
        P = obs.shape[0]
        sy, sx = 10.0 ** rs.uniform(0, 3), 10.0 ** rs.uniform(-4, 0)
        inf = float(rs.choice([1.0, 1.1, 1.5]))
        X = torch.as_tensor(rs.normal(size=(1, k, G)) * sx, dtype=torch.float32, device=dev)
        hx = rs.normal(size=(k, P)) * 0.7 * sy
        yb = torch.as_tensor(hx - hx.mean(axis=0), dtype=torch.float32, device=dev)
        d = torch.as_tensor(rs.normal(size=P) * 0.7 * sy, dtype=torch.float32, device=dev)
        nb = eng.localize(grid, obs, [c])
        if nb.p_max > k:
            continue
        ref = eng.analysis(X.double(), yb.double(), d.double(), nb, inf, method="eig")
        for sp in (1, 0):
            _cabi.set_option("tile_split", sp)
            xa, fl, fin = eng.analysis(X, yb, d, nb, inf, return_flags=True, method="matfun", defer_retry=True)
            fin()
            worst[sp] = max(worst[sp], rel(xa, ref))
        _cabi.set_option("tile_split", 1)
        tiles = eng.localize_tiles(grid, obs, [c], nb.p_max)
        if int(tiles.stats[1].item()) == 0:
            xa, fl, retry = eng.analysis_tiles(X, eng.pack_split(yb, d), P, tiles, inf)
            if int(retry.item()):
                eng.retry_points(X, yb, d, nb, inf, xa, fl)
            worst[2] = max(worst[2], rel(xa, ref))
    print("k = %2d: round-2 tile kernel split %.2e / f32 products %.2e; tile route (letkf_tile2_kernel) %.2e" % (k, worst[1], worst[0], worst[2]), flush=True)
