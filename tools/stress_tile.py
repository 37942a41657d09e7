#!/usr/bin/env python3
"""Randomised cross-check of the sixteen-points-per-wavefront kernel (both product variants) against the float64
eigensolver kernel: random ensemble sizes 2 .. 96 (most not multiples of 8 or 16), list lengths up to the route's limit,
1-D / 2-D geometry (2-D grids in random order: tiles get split), ragged grid sizes, 1 .. 5 state rows, magnitudes of the
observation-space inputs and of the state over eight decades, inflation.  Also the tile route of round 3 (tile lists + split
records + letkf_tile2_kernel) where the unions fit.  Exits non-zero above the north star's 1e-5."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_assimilate_amd as mia
from torch_assimilate_amd import _cabi
mia.build()
dev = torch.device("cuda:0")
eng = mia.LetkfEngine(dev)
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rel = lambda a, b: float(torch.linalg.norm(a.double() - b.double()) / max(float(torch.linalg.norm(b.double())), 1e-300))
worst = {1: (0.0, ""), 0: (0.0, ""), 2: (0.0, ""), 3: (0.0, "")}
served2 = servedw = 0
served = 0
for case in range(n_cases):
    k = int(rs.randint(2, 97))
    nc = int(rs.choice([1, 1, 2]))
    G = int(rs.choice([5, 37, 203, 640]))
    m = int(rs.choice([1, 1, 2, 5]))
    p_target = int(rs.randint(1, min(k, 88) + 1))
    if nc == 1:
        grid = np.arange(G, dtype=np.float64)[:, None]
        stride = rs.choice([1.0, 2.0, 3.0])
        obs = np.arange(0, G, stride)[:, None] + rs.uniform(-0.2, 0.2)
        c = max(0.6, p_target * stride / 4.0)
    else:
        grid, P = rs.uniform(0, 1, size=(G, 2)), 400
        obs = rs.uniform(0, 1, size=(P, 2))
        c = float(np.sqrt(p_target / (P * np.pi)) / 2.0) + 0.01
    P = obs.shape[0]
    sy, sx = 10.0 ** rs.uniform(-4, 3), 10.0 ** rs.uniform(-4, 4)
    inf = float(rs.choice([1.0, 1.1, 1.5]))
    X = torch.as_tensor(rs.normal(size=(m, k, G)) * sx, dtype=torch.float32, device=dev)
    hx = rs.normal(size=(k, P)) * 0.7 * sy
    yb = torch.as_tensor(hx - hx.mean(axis=0), dtype=torch.float32, device=dev)
    d = torch.as_tensor(rs.normal(size=P) * 0.7 * sy, dtype=torch.float32, device=dev)
    nb = eng.localize(grid, obs, [c])
    if nb.p_max > k or nb.p_max + 8 > 96 or k < 2:
        continue                                  # outside the tile route
    served += 1
    try:
        ref = eng.analysis(X.double(), yb.double(), d.double(), nb, inf, method="eig")
    except _cabi.MiaError:                         # (float64 block beyond the LDS: the float32 eigensolver kernel instead)
        ref = eng.analysis(X, yb, d, nb, inf, method="eig")
    tag = "k%d G%d nc%d m%d pmax%d inf%.1f sy%.0e sx%.0e" % (k, G, nc, m, nb.p_max, inf, sy, sx)
    for sp in (1, 0):
        _cabi.set_option("tile_split", sp)
        xa, fl, fin = eng.analysis(X, yb, d, nb, inf, return_flags=True, method="matfun", defer_retry=True)
        fin()
        bad = int((fl & 0xff & ~8).max().cpu())
        e = rel(xa, ref)
        if bad or not np.isfinite(e):
            print("FLAGGED / non-finite:", tag, "split", sp, "flags", bad, "err", e)
            sys.exit(2)
        if e > worst[sp][0]:
            worst[sp] = (e, tag)
    # tile route (csrc/letkf_tile2.hip): tile lists sized by the longest list (+ row blocks until the unions fit)
    extra, tiles = 0, None
    ut0 = max(1, (nb.p_max + 8 + 15) // 16)
    while ut0 + extra <= min(6, (k + 15) // 16 + 1):
        tiles = eng.localize_tiles(grid, obs, [c], nb.p_max, extra_blocks=extra)
        if int(tiles.stats[1].item()) == 0:
            break
        tiles, extra = None, extra + 1
    if tiles is not None:
        served2 += 1
        xa, fl, retry = eng.analysis_tiles(X, eng.pack_split(yb, d), P, tiles, inf)
        if int(retry.item()):
            eng.retry_points(X, yb, d, nb, inf, xa, fl)
        bad = int((fl & 0xff & ~8).max().cpu())
        e = rel(xa, ref)
        if bad or not np.isfinite(e):
            print("FLAGGED / non-finite (tile route):", tag, "flags", bad, "err", e)
            sys.exit(2)
        if e > worst[2][0]:
            worst[2] = (e, tag)
        # the weights on the same route (csrc/letkf_tile2w.hip: unions of at most 32 slots) against the float64 eigensolver's
        if tiles.ut <= 2:
            res = eng.weights_tiles(X, eng.pack_split(yb, d), P, tiles, inf)
            if res is not None:
                xw, W, flw, rw = res
                if int(rw.item()):
                    eng.weights_retry(X, yb, d, nb, inf, xw, W, flw)
                try:
                    _, Wref = eng.analysis(X.double(), yb.double(), d.double(), nb, inf, method="eig", return_weights=True)
                except _cabi.MiaError:
                    _, Wref = eng.analysis(X, yb, d, nb, inf, method="eig", return_weights=True)
                bad = int((flw & 0xff & ~8).max().cpu())
                ew, ex = rel(W, Wref), rel(xw, ref)
                servedw += 1
                if bad or not np.isfinite(ew) or not np.isfinite(ex):
                    print("FLAGGED / non-finite (weights on tiles):", tag, "flags", bad, "err", ew, ex)
                    sys.exit(2)
                if max(ew, ex) > worst[3][0]:
                    worst[3] = (max(ew, ex), tag)
_cabi.set_option("tile_split", 1)
print("%d of %d random cases on the round-2 tile kernel, %d of them also on the tile route" % (served, n_cases, served2))
print("%-6s worst %.2e  at %s" % ("tile2", worst[2][0], worst[2][1]))
print("%-6s worst %.2e  at %s  (%d cases: weights and analysis of mia_letkf_weights_tiles_f32)" % ("tile2w", worst[3][0], worst[3][1], servedw))
for sp in (1, 0):
    print("%-6s worst %.2e  at %s" % ("split" if sp else "f32", worst[sp][0], worst[sp][1]))
sys.exit(1 if max(worst[1][0], worst[0][0], worst[2][0], worst[3][0]) > 1.0e-5 else 0)
